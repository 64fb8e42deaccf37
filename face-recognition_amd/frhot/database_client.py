"""Drop-in for the reference's DatabaseClient (app/src/utils/database_client.py).

Same coroutine names, arguments and return conventions; the HTTP/JSON hop to the
DB service is replaced by direct calls into an in-process GalleryService, so the
4096-float vector is never serialised to text.  Every failure maps to the same
falsy result the reference returns ([] / False).
"""
from __future__ import annotations

import asyncio
import logging
from typing import Any, Dict, List

logger = logging.getLogger(__name__)


class DatabaseClient:
    def __init__(self, service, search_batcher=None):
        """`search_batcher`: a frhot.batcher.SearchBatcher over the same service.  With it, searches
        that are in flight together (the reference allows 10 per process, face_service.py:42) share
        one gallery scan and the event loop is free while the scan runs."""
        self.service = service
        self.search_batcher = search_batcher

    async def ensure_collection_exists(self, collection_name: str) -> bool:
        """database_client.py:62-96."""
        try:
            if self.service.collection_exists(collection_name):
                return True
            status, _ = self.service.create_collection(collection_name)
            return status in (200, 201)
        except Exception as e:
            logger.error(f"Error ensuring collection '{collection_name}' exists: {e}")
            return False

    async def ensure_collections_exist(self, store_id: str) -> bool:
        """database_client.py:98-120: both f"{store_id}_Customers" and f"{store_id}_Employees"."""
        try:
            ok_c = await self.ensure_collection_exists(f"{store_id}_Customers")
            ok_e = await self.ensure_collection_exists(f"{store_id}_Employees")
            return ok_c and ok_e
        except Exception as e:
            logger.error(f"Error ensuring collections exist for store {store_id}: {e}")
            return False

    async def insert_point(self, collection_name: str, vector_embedding: List[float], id: str, name: str,
                           store_id: str, is_update_id: bool = False) -> bool:
        """database_client.py:122-166: True iff the service answers 201."""
        try:
            status, _ = self.service.insert_point(collection_name, vector_embedding, id, name, store_id, is_update_id)
            return status == 201
        except Exception as e:
            logger.error(f"Error inserting point {id}: {e}")
            return False

    @staticmethod
    def _reshape(status, content) -> List[Dict[str, Any]]:
        """database_client.py:237-250."""
        if status != 200:
            return []
        if content.get("message") == "Point found":
            return [{'id': payload.get('id'), 'score': score, 'payload': payload}
                    for score, payload in content.get("data", [])]
        return []

    async def search_point(self, collection_name: str, vector_embedding: List[float],
                           store_id: str) -> List[Dict[str, Any]]:
        """database_client.py:206-254."""
        try:
            if self.search_batcher is not None:
                fut = self.search_batcher.submit(collection_name, vector_embedding, store_id)
                return self._reshape(*await asyncio.wrap_future(fut))
            return self._reshape(*self.service.search_point(collection_name, vector_embedding, store_id))
        except Exception as e:
            logger.error(f"Error searching points: {e}")
            return []

    async def delete_point(self, collection_name: str, id: str) -> bool:
        """database_client.py:291-331: True iff the service answers 200."""
        try:
            status, _ = self.service.delete_point(collection_name, id)
            return status == 200
        except Exception as e:
            logger.error(f"Error deleting point {id}: {e}")
            return False
