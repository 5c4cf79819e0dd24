"""Glue between providers and the store: restatement of the hot-path parts of VerbatimIndex
(verbatim_rag/index.py): `query` search-type resolution and store hand-off (:552-655) and the
batched embed-then-insert of chunks (:200-223,259-288,340-411; 2000-chunk embed batches).
Chunking / document schemas stay with the reference (out of scope); callers pass ready chunks.
"""
from __future__ import annotations

from typing import Any, Dict, List, Optional, Sequence

from .vector_stores import SearchResult, VectorStore


class HotPathIndex:
    EMBED_BATCH = 2000  # index.py:343

    def __init__(self, vector_store: VectorStore, dense_provider=None, sparse_provider=None):
        if dense_provider is None and sparse_provider is None:  # index.py:58-64
            if not bool(getattr(vector_store, "enable_full_text", False)):
                raise ValueError("At least one embedding provider (dense or sparse) must be provided")
        self.vector_store = vector_store
        self.dense_provider = dense_provider
        self.sparse_provider = sparse_provider

    def _generate_embeddings(self, texts: List[str]):
        """index.py:200-223."""
        dense = self.dense_provider.embed_batch(texts) if self.dense_provider else None
        sparse = self.sparse_provider.embed_batch(texts) if self.sparse_provider else None
        return dense, sparse

    def add_chunks(self, ids: Sequence[str], texts: Sequence[str], enhanced_texts: Optional[Sequence[str]] = None,
                   metadatas: Optional[Sequence[Dict[str, Any]]] = None) -> None:
        """Embeddings are computed on the *enhanced* text, raw text is stored for extraction
        (index.py:152-164,470-486)."""
        enhanced_texts = list(enhanced_texts) if enhanced_texts is not None else list(texts)
        metadatas = list(metadatas) if metadatas is not None else [{} for _ in ids]
        for a in range(0, len(ids), self.EMBED_BATCH):
            b = min(len(ids), a + self.EMBED_BATCH)
            dense, sparse = self._generate_embeddings(enhanced_texts[a:b])
            self.vector_store.add_vectors(list(ids[a:b]), dense, sparse, list(texts[a:b]), enhanced_texts[a:b],
                                          metadatas[a:b])

    def query(self, text: Optional[str] = None, k: int = 5, search_type: str = "auto", filter: Optional[str] = None,
              search_params: Optional[Dict[str, Any]] = None, hybrid_weights: Optional[Dict[str, float]] = None,
              rrf_k: int = 60) -> List[SearchResult]:
        """index.py:552-655, branch for branch."""
        if not text:
            return self.vector_store.query(dense_query=None, sparse_query=None, text_query=None, top_k=k, filter=filter,
                                           search_params=search_params)
        if hybrid_weights is not None:
            qd = qs = None
            if "dense" in hybrid_weights and self.dense_provider:
                qd = self.dense_provider.embed_text(text)
            if "sparse" in hybrid_weights and self.sparse_provider:
                qs = self.sparse_provider.embed_text(text)
            return self.vector_store.query(dense_query=qd, sparse_query=qs, text_query=text, top_k=k, filter=filter,
                                           search_params=search_params, hybrid_weights=hybrid_weights, rrf_k=rrf_k)
        if search_type == "auto":
            if self.dense_provider and self.sparse_provider:
                search_type = "hybrid"
            elif self.dense_provider:
                search_type = "dense"
            elif self.sparse_provider:
                search_type = "sparse"
            elif getattr(self.vector_store, "enable_full_text", False):
                search_type = "full_text"
            else:
                raise ValueError("No search method available")
        if search_type == "full_text":
            return self.vector_store.query(dense_query=None, sparse_query=None, text_query=text, top_k=k,
                                           search_type="full_text", filter=filter, search_params=search_params)
        qd = qs = None
        if search_type in ("dense", "hybrid") and self.dense_provider:
            qd = self.dense_provider.embed_text(text)
        if search_type in ("sparse", "hybrid") and self.sparse_provider:
            qs = self.sparse_provider.embed_text(text)
        return self.vector_store.query(dense_query=qd, sparse_query=qs, text_query=text, top_k=k,
                                       search_type=search_type, filter=filter, search_params=search_params, rrf_k=rrf_k)

    # ------------------------------------------------------------------ cross-query batching
    @staticmethod
    def _embed_queries(provider, texts: List[str]):
        fn = getattr(provider, "embed_queries", None)
        return fn(texts) if fn is not None else [provider.embed_text(t) for t in texts]

    def query_batch(self, texts: Sequence[str], k: int = 5, search_type: str = "auto", filter: Optional[str] = None,
                    search_params: Optional[Dict[str, Any]] = None, hybrid_weights: Optional[Dict[str, float]] = None,
                    rrf_k: int = 60) -> List[List[SearchResult]]:
        """`[query(t, k, ...) for t in texts]` for many concurrent queries: the query embeddings go through the
        providers as shared batches (`embed_queries` when the provider has it) and the store answers them in one
        `query_batch` call when it has one; anything else takes `query`, one text at a time."""
        texts = list(texts)
        store_batch = getattr(self.vector_store, "query_batch", None)
        if store_batch is None or not texts or any(not t for t in texts):
            return [self.query(t, k, search_type, filter, search_params, hybrid_weights, rrf_k) for t in texts]
        if hybrid_weights is not None:
            qd = self._embed_queries(self.dense_provider, texts) if "dense" in hybrid_weights and self.dense_provider else None
            qs = self._embed_queries(self.sparse_provider, texts) if "sparse" in hybrid_weights and self.sparse_provider else None
            return store_batch(dense_queries=qd, sparse_queries=qs, text_queries=texts, top_k=k, filter=filter,
                               search_params=search_params, hybrid_weights=hybrid_weights, rrf_k=rrf_k)
        if search_type == "auto":
            if self.dense_provider and self.sparse_provider:
                search_type = "hybrid"
            elif self.dense_provider:
                search_type = "dense"
            elif self.sparse_provider:
                search_type = "sparse"
        if search_type not in ("dense", "sparse", "hybrid"):
            return [self.query(t, k, search_type, filter, search_params, hybrid_weights, rrf_k) for t in texts]
        qd = self._embed_queries(self.dense_provider, texts) if search_type in ("dense", "hybrid") and self.dense_provider else None
        qs = self._embed_queries(self.sparse_provider, texts) if search_type in ("sparse", "hybrid") and self.sparse_provider else None
        return store_batch(dense_queries=qd, sparse_queries=qs, text_queries=texts, top_k=k, search_type=search_type,
                           filter=filter, search_params=search_params, rrf_k=rrf_k)
