"""FastAPI surface of the recognition path: the two handlers of the reference's
face router (app/src/api/routers/face.py:57-72,90-110) behind its X-API-Key check
(app/src/api/routes.py:17-28), wired to the GPU engine.

`create_app(ctx)` builds everything in one process: context -> micro-batcher ->
GalleryService -> search micro-batcher -> DatabaseClient -> FaceService.  The request models repeat the
reference's field set (app/src/core/models.py:11-28,43-54).
"""
from __future__ import annotations

import logging

from fastapi import APIRouter, Depends, FastAPI, HTTPException, Security, status
from fastapi.responses import JSONResponse
from fastapi.security import APIKeyHeader
from pydantic import BaseModel

logger = logging.getLogger(__name__)
DEFAULT_API_KEY = "mac_dinh"   # app/config/settings.py:58


class CreateFace(BaseModel):
    img_base64: str = None
    id: str = None
    name: str = None
    role: str = None
    store_id: str = None
    is_update: bool = False


class FaceRecog(BaseModel):
    img_base64: str = None
    role: str = None
    store_id: str = None


def build_router(face_service, api_key: str = DEFAULT_API_KEY) -> APIRouter:
    api_key_header = APIKeyHeader(name="X-API-Key", auto_error=False)

    async def verify_api_key(key: str = Security(api_key_header)):
        if key == api_key:
            return key
        raise HTTPException(status_code=status.HTTP_403_FORBIDDEN,
                            detail="Could not validate credentials - Access Denied")

    router = APIRouter(tags=["Face"], dependencies=[Depends(verify_api_key)])

    @router.post("/face_recog_img_base64")
    async def face_recog_img_base64(data: FaceRecog):
        try:
            return await face_service.recognize_face(data)
        except Exception as e:
            logger.error(f"Face recognition failed: {e}")
            return JSONResponse(status_code=500, content={'status': 2, 'message': "Internal server error"})

    @router.post("/create_face_img_base64")
    async def create_face_img_base64(data: CreateFace):
        try:
            return await face_service.create_face(data)
        except Exception as e:
            logger.error(f"Face creation failed: {e}", exc_info=True)
            return JSONResponse(status_code=500, content={'status': 2, 'message': f"Internal server error: {e}"})

    return router


def create_app(ctx, capacity: int = 1 << 16, api_key: str = DEFAULT_API_KEY, max_delay_s: float = 0.002,
               fused: bool = True, quality_gates: bool = False, detector=None, anti_spoof=None,
               assume_real: bool = False) -> FastAPI:
    """A FastAPI app serving the two routes from one GPU context with a loaded model.
    `fused`: requests in flight together share ONE frhot_recognise call each round (crops in,
    voted answers out); False serves through the two seams get_embedding / search_point with one
    micro-batcher each, as a maintainer who patches only those seams would run it.
    `quality_gates`: employees' recognition requests pass the reference's check-in gates
    (face_service.py:73-219): blur on this context, the others over the models registered with
    frhot.quality.install_models.
    `detector`: a frhot.detector.Detector on this context.  It takes the place of DeepFace's yolov8
    backend behind detect_face and the spoof pass (legacy.py:60-65,80-86): frames of requests in
    flight share device calls through a DetectBatcher.  DeepFace's anti-spoofing network is not part
    of this package: `anti_spoof(face crop) -> bool` supplies its verdict, or `assume_real=True`
    accepts every face.  Needs `quality_gates` (the service detects only with the gates on)."""
    from . import legacy, quality
    from .batcher import DetectBatcher, EmbedBatcher, RecogniseBatcher, SearchBatcher
    from .database_client import DatabaseClient
    from .face_service import FaceService
    from .gallery_service import GalleryService

    batcher = EmbedBatcher(ctx, max_delay_s=max_delay_s)
    legacy.install_engine(batcher)
    if quality_gates:
        quality.install_engine(ctx)
    det_batcher = None
    if detector is not None:
        if not quality_gates:
            raise ValueError("a detector is used by the quality gates only: pass quality_gates=True")
        if anti_spoof is None and not assume_real:
            raise ValueError("no anti-spoofing model: pass anti_spoof=callable, or assume_real=True to accept every face")
        det_batcher = DetectBatcher(detector, max_delay_s=max_delay_s)
        installed = quality.installed_models()
        quality.install_models(face_mesh=installed["face_mesh"], face_detection=installed["face_detection"],
                               extract_faces=lambda image: det_batcher.extract_faces(
                                   image, expand_percentage=quality.FACE_EXT, anti_spoof=anti_spoof, assume_real=assume_real))
    service = GalleryService(ctx, capacity=capacity)
    search_batcher = SearchBatcher(service, max_delay_s=max_delay_s)
    recogniser = RecogniseBatcher(service, ctx.max_batch, max_delay_s=max_delay_s) if fused else None
    face_service = FaceService(DatabaseClient(service, search_batcher), recogniser=recogniser,
                               max_in_flight=ctx.max_batch, quality_gates=quality_gates)
    app = FastAPI(title="frhot recognition path")
    app.include_router(build_router(face_service, api_key))
    app.state.batcher = batcher
    app.state.search_batcher = search_batcher
    app.state.gallery_service = service
    app.state.face_service = face_service
    app.state.recogniser = recogniser
    app.state.det_batcher = det_batcher

    def close():
        if det_batcher is not None:
            det_batcher.close()
            quality.install_models(face_mesh=installed["face_mesh"], face_detection=installed["face_detection"])
        batcher.close()
        search_batcher.close()
        if recogniser is not None:
            recogniser.close()
        service.close()
        legacy.install_engine(None)
        if quality_gates:
            quality.install_engine(None)

    app.state.close = close
    app.router.on_shutdown.append(close)

    return app
