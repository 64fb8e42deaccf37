"""Drop-in for the recognise / enrol paths of the reference's FaceService
(app/src/services/face_service.py:54-616): same coroutine names, same status
codes and JSON bodies, with the embedding and the search served by libfrhot.so.

The check-in quality gates (face_service.py:73-219) are applied when the service is built
with `quality_gates=True`: the blur gate measures on the device; the gates that rest on
third-party models (head direction, eyes, distance, full face, detection, anti-spoofing)
run when the host has registered those models with frhot.quality.install_models and are
skipped otherwise.  Without a registered detector the service takes the reference's
`is_detect_face=False` branch (face_service.py:127-131): the whole decoded image is the
face box.  The MinIO image upload is not reproduced: `image_sink`, if given, receives what
the reference would hand to ImageProcessor.save_face_image.
"""
from __future__ import annotations

import asyncio
import base64
import io
import logging
import time
from concurrent.futures import ThreadPoolExecutor
from typing import Any, Dict, List, Tuple

import numpy as np
from fastapi.responses import JSONResponse

from . import quality
from .legacy import adjust_gamma, check_condition, face_is_real, get_embedding

logger = logging.getLogger(__name__)

HTTP_SEMAPHORE_LIMIT = 10        # face_service.py:42
PROCESSING_SEMAPHORE_LIMIT = 10  # face_service.py:43


def decode_image(img_base64: str) -> np.ndarray:
    """base64 -> pixel array, shaped as cv2.imdecode(flags=-1) shapes it (face_service.py:65-66):
    colour -> uint8 [h,w,3] BGR; colour or grey with alpha -> [h,w,4] BGRA; grey -> [h,w];
    16-bit grey -> uint16 [h,w] (the caller's astype('uint8') wraps it, as in the reference).
    The decoder is PIL (cv2 is not a dependency of this package): JPEG IDCT and chroma
    upsampling may differ from OpenCV's libjpeg build by a few levels per pixel, and 16-bit
    colour files come out as 8-bit.  Hosts that have cv2 pass their own decoder to FaceService."""
    from PIL import Image
    contents = base64.b64decode(img_base64)
    img = Image.open(io.BytesIO(contents))
    img.load()
    if img.mode in ("I;16", "I;16B", "I;16L", "I"):
        return np.asarray(img).astype(np.uint16)
    if img.mode in ("LA", "PA") or (img.mode == "P" and "transparency" in img.info):
        img = img.convert("RGBA")
    elif img.mode not in ("RGB", "RGBA", "L"):
        img = img.convert("L" if img.mode == "1" else "RGB")
    arr = np.asarray(img)
    if arr.ndim == 2:
        return arr
    if arr.shape[2] == 4:
        return np.ascontiguousarray(arr[:, :, [2, 1, 0, 3]])
    return np.ascontiguousarray(arr[:, :, ::-1])


class FaceService:
    def __init__(self, database_client, image_sink=None, decoder=None, recogniser=None, max_in_flight=None,
                 quality_gates=False):
        """`decoder`: base64 string -> pixel array; default decode_image (PIL).  A host with OpenCV
        passes `lambda b: cv2.imdecode(np.frombuffer(base64.b64decode(b), np.uint8), -1)` to decode
        exactly as the reference does."""
        self.database_client = database_client
        self.image_sink = image_sink
        self.decoder = decoder or decode_image
        # `recogniser`: a frhot.batcher.RecogniseBatcher.  With it, embedding and search of a request
        # are one fused device call shared with the requests in flight beside it (the descriptor
        # never becomes a Python list); without it the two seams get_embedding / search_point are
        # used as in the reference.  `max_in_flight` bounds the requests inside the fused step
        # (the reference's limit of 10 bounds CPU threads, face_service.py:43; here it bounds the
        # device batch).
        self.recogniser = recogniser
        self.quality_gates = bool(quality_gates)
        self._fused_limit = max_in_flight or PROCESSING_SEMAPHORE_LIMIT
        self._fused_sem = None
        self._http_sem = None
        self._proc_sem = None
        self._pool = ThreadPoolExecutor(max_workers=PROCESSING_SEMAPHORE_LIMIT, thread_name_prefix="frhot-embed")

    def _sems(self):
        if self._http_sem is None:
            self._http_sem = asyncio.Semaphore(HTTP_SEMAPHORE_LIMIT)
            self._proc_sem = asyncio.Semaphore(PROCESSING_SEMAPHORE_LIMIT)
        return self._http_sem, self._proc_sem

    async def _save(self, data, img, face_id, name, is_checkin=True):
        if self.image_sink is not None:
            await self.image_sink(data, img, face_id, name, is_checkin)

    @staticmethod
    def _refuse(message, status_code=400):
        return False, JSONResponse(status_code=status_code, content={'status': 2, 'message': message})

    def _locate_face(self, data, img_decode, is_detect_face, is_checkin):
        """face_service.py:72-176: the gates on the whole image, detection, the choice of the
        nearest face and the distance gate.  Returns (True, (x1, y1, x2, y2)) or (False, response).
        A detector failure answers 500 at check-in AND outside it: the reference's other branch
        (face_service.py:125-126) returns `img_decode` after `del contents, img_decode` (:117), which
        raises UnboundLocalError into the outer handler (:133-143), i.e. the same 500 body."""
        have = quality.models_installed() if self.quality_gates else {}
        if is_checkin and have.get("face_mesh"):
            ok, message = quality.check_face_left_right(img_decode)
            if not ok:
                return self._refuse(message)
            if not quality.check_eyes_open(img_decode):
                return self._refuse("Eyes are closed! Please open your eyes")
        detected = is_detect_face and have.get("extract_faces")
        if detected:
            try:
                boxes, _scores, distances = quality.detect_face(img_decode)
            except Exception as e:
                logger.warning(f"{data.store_id} - Error when detecting face: {e}")
                return self._refuse("Error when detecting face! Please try again", 500)
        else:
            boxes, distances = [[0, 0, img_decode.shape[1], img_decode.shape[0]]], [50]   # face_service.py:128-131
        try:
            x, y, w, h = boxes[int(np.argmin(distances))]   # no usable face: ValueError (face_service.py:147)
            x1, y1, x2, y2 = int(x), int(y), int(x + w), int(y + h)
            if is_checkin and detected:
                distance = quality.distance_face_to_camera((x1, y1, x2, y2), img_decode.shape[1])
                if distance < 20:
                    return self._refuse("Face is too close! Please move back")
                if distance > 70:
                    return self._refuse("Face is too far! Please move forward")
        except Exception as e:                               # the reference's second try block (face_service.py:261-273)
            logger.warning(f"{data.store_id} - Error in face processing: {e}")
            return self._refuse("Error when processing face! Please try again", 500)
        return True, (x1, y1, x2, y2)

    def _crop_gates(self, face, is_checkin, sharpness=None):
        """face_service.py:188-219 on the crop: full face, then blur.  `sharpness`: the crop's
        Laplacian variance if a fused device call already measured it.  None = passed."""
        if not (is_checkin and self.quality_gates):
            return None
        if sharpness is None and quality.models_installed()["face_detection"]:
            ok, message = quality.is_full_face(face)
            if not ok:
                return self._refuse(message)[1]
        if sharpness is None:
            sharpness = quality.laplacian_variance_of(face)
        if not sharpness >= quality.BLUR_THRESHOLD:
            return self._refuse("Face is blur! Please keep your face in focus")[1]
        return None

    async def detect_and_embed_face(self, data, is_detect_face: bool = True,
                                    is_checkin: bool = True) -> Tuple[bool, Any]:
        """face_service.py:54-273: decode -> gates -> crop -> gamma -> embedding."""
        _, proc = self._sems()
        async with proc:
            loop = asyncio.get_running_loop()
            try:
                img_decode = self.decoder(data.img_base64)
                ok, where = await loop.run_in_executor(
                    self._pool, self._locate_face, data, img_decode, is_detect_face, is_checkin)
            except Exception as e:
                logger.warning(f"{data.store_id} - Error when decoding image: {e}")
                return self._refuse("Error when detecting face! Please try again", 500)
            if not ok:
                return False, where
            try:
                x1, y1, x2, y2 = where
                face = img_decode[y1:y2, x1:x2].astype('uint8')        # face_service.py:180-181
                refused = await loop.run_in_executor(self._pool, self._crop_gates, face, is_checkin)
                if refused is not None:
                    return False, refused
                face = adjust_gamma(face, gamma=1.5)                   # face_service.py:223
                try:
                    # the spoof verdict (legacy.py:60-74) belongs to the gates: one condition for it in
                    # both paths (see _fused), quality_gates and a registered detector
                    whole = img_decode if self.quality_gates else None
                    emb, is_real = await loop.run_in_executor(self._pool, lambda: get_embedding(face, whole))
                    if not is_real and is_checkin:
                        return self._refuse("Face is not real! Please use your real face")
                except Exception as e:
                    logger.warning(f"{data.store_id} - Error when encoding face: {e}")
                    return self._refuse("Error! Please try again", 500)
                return True, (emb, img_decode)
            except Exception as e:
                logger.warning(f"{data.store_id} - Error in face processing: {e}")
                return self._refuse("Error when processing face! Please try again", 500)

    def _decode_and_crop(self, data, is_checkin):
        """face_service.py:65-181 without the gamma LUT (the device applies it): (img_decode, ok,
        crop or response)."""
        img_decode = self.decoder(data.img_base64)
        ok, where = self._locate_face(data, img_decode, True, is_checkin)
        if not ok:
            return img_decode, ok, where
        x1, y1, x2, y2 = where
        return img_decode, True, img_decode[y1:y2, x1:x2].astype('uint8')

    async def _fused(self, data, collection_name: str, want_embedding: bool, is_checkin: bool = False):
        """decode -> gates -> crop -> ONE device call (gamma LUT + descriptor + search + vote input,
        and the blur measure at check-in).  Returns (False, JSONResponse) with the reference's
        failure bodies, or (True, (embedding or None, img_decode, search_result))."""
        if self._fused_sem is None:
            self._fused_sem = asyncio.Semaphore(self._fused_limit)
        loop = asyncio.get_running_loop()
        gated = is_checkin and self.quality_gates
        async with self._fused_sem:
            try:
                img_decode, ok, face = await loop.run_in_executor(self._pool, self._decode_and_crop, data, is_checkin)
            except Exception as e:
                logger.warning(f"{data.store_id} - Error when decoding image: {e}")
                return self._refuse("Error when detecting face! Please try again", 500)
            if not ok:
                return False, face
            try:
                if gated and quality.models_installed()["face_detection"]:
                    ok, message = await loop.run_in_executor(self._pool, quality.is_full_face, face)
                    if not ok:
                        return self._refuse(message)
                fut = self.recogniser.submit(collection_name, face, data.store_id, want_embedding,
                                             want_sharpness=gated)
                status, content, emb, *sharp = await asyncio.wrap_future(fut)
                if gated:
                    refused = self._crop_gates(face, True, sharpness=sharp[0])
                    if refused is not None:
                        return False, refused
                if self.quality_gates:
                    # as get_embedding does in the other path: the detector runs on every request (its
                    # failure answers 500 below), the verdict counts at check-in only (face_service.py:238)
                    is_real = await loop.run_in_executor(self._pool, face_is_real, img_decode)
                    if not is_real and is_checkin:
                        return self._refuse("Face is not real! Please use your real face")
            except Exception as e:
                logger.warning(f"{data.store_id} - Error when encoding face: {e}")
                return self._refuse("Error! Please try again", 500)
        if status == 200 and content.get("message") == "Point found":
            result = {"message": "Point found", "data": [[s, p] for s, p in content.get("data", [])]}
        else:
            result = {"message": "Point not found", "data": []}
        return True, (emb, img_decode, result)

    async def search_face(self, collection_name: str, embedding: List[float], store_id: str) -> Dict[str, Any]:
        """face_service.py:275-308."""
        http, _ = self._sems()
        async with http:
            try:
                results = await self.database_client.search_point(
                    collection_name=collection_name, vector_embedding=embedding, store_id=store_id)
                if results:
                    return {"message": "Point found", "data": [[r['score'], r['payload']] for r in results]}
                return {"message": "Point not found", "data": []}
            except Exception as e:
                logger.error(f"Error in search_face: {e}")
                return {"data": []}

    def extract_face_info(self, search_result: Dict[str, Any]) -> Tuple[str, str, str]:
        """face_service.py:310-330."""
        try:
            if not search_result or 'data' not in search_result or not search_result['data']:
                return "Unknown", "Unknown", "Unknown"
            item = search_result['data'][0]
            if len(item) > 1 and isinstance(item[1], dict):
                return (item[1].get('id', "Unknown"), item[1].get('name', "Unknown"),
                        item[1].get('time_created', "Unknown"))
            return "Unknown", "Unknown", "Unknown"
        except Exception as e:
            logger.error(f"Error extracting face info: {e}")
            return "Unknown", "Unknown", "Unknown"

    async def recognize_face(self, data) -> JSONResponse:
        """face_service.py:332-477."""
        img_decode = None
        t0 = time.time()
        try:
            ok, message = check_condition(data, is_checkin=True)
            if not ok:
                return JSONResponse(status_code=400, content={'status': 2, 'message': message})
            if data.role == '1':
                collection_name, is_checkin = f'{data.store_id}_Employees', True
            elif data.role == '0':
                collection_name, is_checkin = f'{data.store_id}_Customers', False
            else:
                return JSONResponse(status_code=400, content={'status': 2, 'message': "Invalid role"})
            if self.recogniser is not None:
                if not await self.database_client.ensure_collections_exist(data.store_id):
                    return JSONResponse(status_code=500, content={'status': 2, 'message': "Error when create collection"})
                ok, message = await self._fused(data, collection_name, want_embedding=False, is_checkin=is_checkin)
                if not ok:
                    return message
                _emb, img_decode, search_result = message
                if search_result is None:
                    await self._save(data, img_decode, "Unknown", "Unknown")
                    return JSONResponse(status_code=200, content={'status': 1, 'id': "Unknown", 'name': "Unknown"})
            else:
                collection_exists, (check_emb, message) = await asyncio.gather(
                    self.database_client.ensure_collections_exist(data.store_id),
                    self.detect_and_embed_face(data, is_checkin=is_checkin))
                if not collection_exists:
                    return JSONResponse(status_code=500, content={'status': 2, 'message': "Error when create collection"})
                if not check_emb:
                    return message
                emb, img_decode = message
                if emb is None:
                    await self._save(data, img_decode, "Unknown", "Unknown")
                    return JSONResponse(status_code=200, content={'status': 1, 'id': "Unknown", 'name': "Unknown"})
                search_result = await self.search_face(collection_name, emb, data.store_id)
            face_id, name, _time_created = self.extract_face_info(search_result)
            if face_id == "Unknown" and name == "Unknown":
                return JSONResponse(status_code=404, content={
                    'status': 0, 'message': "Face is not existed! Please register your face or checkin again"})
            await self._save(data, img_decode, face_id, name)
            logger.info(f"[TIMING] {data.store_id} - Total successful request time: {time.time() - t0:.3f}s")
            return JSONResponse(status_code=200, content={'status': 1, 'id': face_id, 'name': name})
        except Exception as e:
            logger.error(f"recognize_face - {getattr(data, 'store_id', None)} - Error: {e}")
            try:
                if img_decode is not None:
                    await self._save(data, img_decode, "Unknown", "Unknown")
            except Exception as save_error:
                logger.error(f"Failed to save image: {save_error}")
            return JSONResponse(status_code=500, content={'status': 1, 'id': "Unknown", 'name': "Unknown"})

    async def create_face(self, data, update_face=False) -> JSONResponse:
        """face_service.py:479-616."""
        if update_face or data.is_update:
            data.is_update = True
            verb = "update"
        else:
            verb = "create"
        id_value, name_value, store_id = data.id, data.name, data.store_id
        ok, message = check_condition(data, is_checkin=False)
        if not ok:
            return JSONResponse(status_code=400, content={'status': 2, 'message': message})
        if data.role == '1':
            collection_name = f'{store_id}_Employees'
        elif data.role == '0':
            collection_name = f'{store_id}_Customers'
        else:
            return JSONResponse(status_code=400, content={'status': 2, 'message': "Invalid role"})
        if self.recogniser is not None:
            if not await self.database_client.ensure_collections_exist(store_id):
                return JSONResponse(status_code=500, content={'status': 2, 'message': "Error! Please try again"})
            ok, message = await self._fused(data, collection_name, want_embedding=True)
            if not ok:
                return message
            emb, img_decode, search_result = message
            if search_result is None:
                await self._save(data, img_decode, id_value, name_value, is_checkin=False)
                return JSONResponse(status_code=200, content={
                    'status': 1, 'message': f'{verb} face {name_value} with id {id_value} successfully'})
            if not data.is_update and search_result.get('data'):
                return JSONResponse(status_code=409, content={
                    'status': 0, 'message': "Face is existed! Please use another face"})
        else:
            collection_exists, (check_emb, message) = await asyncio.gather(
                self.database_client.ensure_collections_exist(store_id),
                self.detect_and_embed_face(data, is_checkin=False))
            if not collection_exists:
                return JSONResponse(status_code=500, content={'status': 2, 'message': "Error! Please try again"})
            if not check_emb:
                return message
            emb, img_decode = message
            if emb is None:
                await self._save(data, img_decode, id_value, name_value, is_checkin=False)
                return JSONResponse(status_code=200, content={
                    'status': 1, 'message': f'{verb} face {name_value} with id {id_value} successfully'})
            if not data.is_update:
                search_result = await self.search_face(collection_name, emb, store_id)
                if search_result.get('data') and len(search_result['data']) > 0:
                    return JSONResponse(status_code=409, content={
                        'status': 0, 'message': "Face is existed! Please use another face"})
        success = await self.database_client.insert_point(
            collection_name=collection_name, vector_embedding=emb, id=id_value, name=name_value,
            store_id=store_id, is_update_id=False)
        if not success:
            return JSONResponse(status_code=500, content={'status': 2, 'message': "Error when insert face"})
        await self._save(data, img_decode, id_value, name_value, is_checkin=False)
        return JSONResponse(status_code=201, content={
            'status': 1, 'message': f'{verb} face {name_value} with id {id_value} successfully'})
