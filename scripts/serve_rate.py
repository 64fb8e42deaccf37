#!/usr/bin/env python3
"""Developer tool: requests/s through the two FastAPI routes of frhot.api.create_app, in process
(httpx ASGI transport, no sockets), with `conc` recognise requests in flight at a time.

It measures the Python host stack the reference keeps (base64 + image decode, gamma LUT, the
list-of-floats embedding, the vote) on top of the GPU path; the micro-batchers turn the
requests that wait together into shared device calls.
Usage: python scripts/serve_rate.py [conc] [requests] [enrolled]
"""
import asyncio
import base64
import io
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "face-recognition_amd"))

import httpx  # noqa: E402
import numpy as np  # noqa: E402
from PIL import Image  # noqa: E402

from frhot import synth  # noqa: E402
from frhot.api import create_app  # noqa: E402
from frhot.engine import Context  # noqa: E402
from frhot.weights import pack_blob, synthetic_weights  # noqa: E402


def b64_jpeg(img):
    buf = io.BytesIO()
    Image.fromarray(img[:, :, ::-1]).save(buf, format="JPEG", quality=90)
    return base64.b64encode(buf.getvalue()).decode()


async def run(ctx, fused, conc, total, faces, detector=None, role="1"):
    if detector is None:
        app = create_app(ctx, capacity=1 << 16, api_key="k", max_delay_s=0.004, fused=fused)
    else:       # whole frames in, the face found on the GPU (customers' requests: detection without the check-in gates)
        app = create_app(ctx, capacity=1 << 16, api_key="k", max_delay_s=0.004, fused=fused, quality_gates=True,
                         detector=detector, assume_real=True)
    enrolled = len(faces)
    H = {"X-API-Key": "k"}
    answers = {}
    async with httpx.AsyncClient(transport=httpx.ASGITransport(app=app), base_url="http://frhot", timeout=120) as c:
        for i, img in enumerate(faces):
            r = await c.post("/create_face_img_base64", headers=H, json={
                "img_base64": img, "id": str(i), "name": f"P{i}", "role": role, "store_id": "S", "is_update": True})
            assert r.status_code == 201, r.text
        sem = asyncio.Semaphore(conc)
        ok = 0

        async def one(i):
            nonlocal ok
            async with sem:
                r = await c.post("/face_recog_img_base64", headers=H, json={
                    "img_base64": faces[i % enrolled], "role": role, "store_id": "S"})
                answers[i] = (r.status_code, r.json())
                ok += r.status_code == 200 and r.json().get("id") == str(i % enrolled)
        await asyncio.gather(*[one(i) for i in range(min(total, 2 * conc))])      # warm-up
        ok = 0
        st = app.state
        c0 = (st.batcher.batches, st.search_batcher.batches, st.recogniser.batches if fused else 0)
        t0 = time.perf_counter()
        await asyncio.gather(*[one(i) for i in range(total)])
        dt = time.perf_counter() - t0
        c1 = (st.batcher.batches, st.search_batcher.batches, st.recogniser.batches if fused else 0)
    calls = [b - a for a, b in zip(c0, c1)]
    if detector is not None:
        print(f"  detector: {st.det_batcher.items} frames in {st.det_batcher.batches} device calls")
    print(f"serve ({'fused' if fused else 'two seams'}{', GPU detector' if detector is not None else ''}): {total} recognise requests, {conc} in flight: "
          f"{total / dt:.0f} requests/s, {dt / total * 1e3:.2f} ms each; {ok} answered with the enrolled id; "
          f"device calls: {calls[0]} embeds, {calls[1]} scans, {calls[2]} fused")
    app.state.close()
    return total / dt, answers


async def main():
    conc = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    total = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
    enrolled = int(sys.argv[3]) if len(sys.argv) > 3 else 64
    ctx = Context(device=0, max_batch=256)
    ctx.load_model(pack_blob(synthetic_weights(seed=0xFACE)), input_scale=1.0 / 16.0)
    rng = np.random.default_rng(5)
    faces = [b64_jpeg(synth.crop(rng, int(rng.integers(160, 400)), int(rng.integers(160, 400)))) for _ in range(enrolled)]
    r_seams, a_seams = await run(ctx, False, conc, total, faces)
    r_fused, a_fused = await run(ctx, True, conc, total, faces)
    same = sum(a_seams[i] == a_fused[i] for i in range(total))
    print(f"fused / two seams = {r_fused / r_seams:.2f}x; {same} of {total} answers identical")
    if os.environ.get("SERVE_RATE_DETECTOR", "1") != "0":
        # random weights only find "faces" once the oracle has rescaled them layer by layer (a developer tool may use it)
        sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
        from det_util import calibrated_weights
        from frhot import detector as D
        det = D.Detector(ctx, D.pack_blob(calibrated_weights(seed=3, cls_bias=-6.0)[0]), max_batch=32)
        frames = []
        while len(frames) < enrolled:       # frames in which the random-weight detector sees a confident "face" (after the JPEG round trip)
            f = b64_jpeg(np.clip(synth.crop(rng, 480, 640).astype(np.float32) + rng.normal(0, 20, (480, 640, 3)), 0, 255).astype(np.uint8))
            seen = np.asarray(Image.open(io.BytesIO(base64.b64decode(f))).convert("RGB"))[:, :, ::-1]
            regs = det.regions(np.ascontiguousarray(seen))
            if regs and regs[0]["confidence"] >= 0.8 and regs[0]["w"] > 16 and regs[0]["h"] > 16:
                frames.append(f)
        await run(ctx, True, conc, total, frames, detector=det, role="0")
        det.close()
    ctx.close()


if __name__ == "__main__":
    asyncio.run(main())
