"""Glue between providers and the store for the hot path: the query dispatch of VerbatimIndex
(verbatim_rag/index.py:552-655) and the batched embed-then-insert of chunks (:200-223,259-288,340-411; 2000-chunk embed
batches).  Chunking / document schemas stay with the reference (out of scope); callers pass ready chunks.

The dispatch is one planning step (`_plan`) shared by the single-query and the cross-query entry points: it turns
(search_type, hybrid_weights, which providers exist) into "which providers embed the text" plus the keyword arguments the
store receives -- the contract pinned by the `index_query_trace` fixture, which records what the reference hands to
`VectorStore.query` for every combination.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Any, Dict, List, Optional, Sequence

from .vector_stores import SearchResult, VectorStore

# search_type "auto": resolved from the providers the index was built with (index.py:609-619)
_AUTO = {(True, True): "hybrid", (True, False): "dense", (False, True): "sparse"}


@dataclass
class _Plan:
    embed_dense: bool = False
    embed_sparse: bool = False
    pass_text: bool = True                  # the store sees the query text (full-text capable stores use it)
    batchable: bool = False                 # the store's query_batch can answer it in one pass per method
    store_kwargs: Dict[str, Any] = field(default_factory=dict)


class HotPathIndex:
    EMBED_BATCH = 2000  # index.py:343

    def __init__(self, vector_store: VectorStore, dense_provider=None, sparse_provider=None):
        if dense_provider is None and sparse_provider is None:  # index.py:58-64
            if not bool(getattr(vector_store, "enable_full_text", False)):
                raise ValueError("At least one embedding provider (dense or sparse) must be provided")
        self.vector_store = vector_store
        self.dense_provider = dense_provider
        self.sparse_provider = sparse_provider

    # ------------------------------------------------------------------ ingest
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

    # ------------------------------------------------------------------ query dispatch
    def _plan(self, search_type: str, filter, search_params, hybrid_weights, rrf_k) -> _Plan:
        common = {"filter": filter, "search_params": search_params}
        have_d, have_s = bool(self.dense_provider), bool(self.sparse_provider)
        if hybrid_weights is not None:      # the weights name the methods; search_type is ignored (index.py:590-607)
            return _Plan(embed_dense="dense" in hybrid_weights and have_d, embed_sparse="sparse" in hybrid_weights and have_s,
                         batchable=True, store_kwargs={**common, "hybrid_weights": hybrid_weights, "rrf_k": rrf_k})
        if search_type == "auto":
            search_type = _AUTO.get((have_d, have_s))
            if search_type is None:
                if not getattr(self.vector_store, "enable_full_text", False):
                    raise ValueError("No search method available")
                search_type = "full_text"
        if search_type == "full_text":      # no embeddings, no rrf_k (index.py:622-631)
            return _Plan(store_kwargs={**common, "search_type": "full_text"})
        return _Plan(embed_dense=search_type in ("dense", "hybrid") and have_d,
                     embed_sparse=search_type in ("sparse", "hybrid") and have_s,
                     batchable=search_type in ("dense", "sparse", "hybrid"),
                     store_kwargs={**common, "search_type": search_type, "rrf_k": rrf_k})

    def query(self, text: Optional[str] = None, k: int = 5, search_type: str = "auto", filter: Optional[str] = None,
              search_params: Optional[Dict[str, Any]] = None, hybrid_weights: Optional[Dict[str, float]] = None,
              rrf_k: int = 60) -> List[SearchResult]:
        """Same hand-off to the store as the reference's VerbatimIndex.query (index.py:552-655)."""
        if not text:                        # browse / filter-only (index.py:579-588)
            return self.vector_store.query(dense_query=None, sparse_query=None, text_query=None, top_k=k, filter=filter,
                                           search_params=search_params)
        plan = self._plan(search_type, filter, search_params, hybrid_weights, rrf_k)
        dense = self.dense_provider.embed_text(text) if plan.embed_dense else None
        sparse = self.sparse_provider.embed_text(text) if plan.embed_sparse else None
        return self.vector_store.query(dense_query=dense, sparse_query=sparse, text_query=text, top_k=k, **plan.store_kwargs)

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
        one_by_one = store_batch is None or not texts or any(not t for t in texts)
        plan = None if one_by_one else self._plan(search_type, filter, search_params, hybrid_weights, rrf_k)
        if plan is None or not plan.batchable:
            return [self.query(t, k, search_type, filter, search_params, hybrid_weights, rrf_k) for t in texts]
        dense = self._embed_queries(self.dense_provider, texts) if plan.embed_dense else None
        sparse = self._embed_queries(self.sparse_provider, texts) if plan.embed_sparse else None
        return store_batch(dense_queries=dense, sparse_queries=sparse, text_queries=texts, top_k=k, **plan.store_kwargs)
