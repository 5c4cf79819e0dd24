"""Static-template query pipeline: the caller-side steps of VerbatimRAG.query that BASELINE
config 1 exercises (verbatim_rag/core.py:237-277, template_mode="static"), restated so the
hot path can be driven end to end where the reference is not installed.  With the reference
installed, pass GpuModelSpanExtractor / the Gpu*Provider / GpuVectorStore objects to the
reference's own VerbatimRAG instead -- they are drop-ins for its plug points.
"""
from __future__ import annotations

from typing import Any, Dict, List, Optional

from .response_builder import QueryResponse, ResponseBuilder

# packages/core/verbatim_core/templates/static.py:23-30
DEFAULT_STATIC_TEMPLATE = """## Response

The following is an unordered list of verbatim excerpts from the source documents. No synthesis or ranking is implied:

[DISPLAY_SPANS]

---
*These excerpts are taken verbatim from the source documents to ensure accuracy.*"""

NO_INFO = "No relevant information found in the provided documents."


def fill_static_template(display_spans: List[Dict[str, Any]], template: str = DEFAULT_STATIC_TEMPLATE) -> str:
    """Inline-citation aggregate fill for plain (non-table) spans: `[n] span` blocks joined by a
    blank line (templates/filler.py:114-145,147-174)."""
    blocks = []
    for i, span in enumerate(display_spans, 1):
        cleaned = span.get("text", "").strip()
        if cleaned:
            blocks.append(f"[{i}] {cleaned}")
    content = "\n\n".join(blocks) if blocks else NO_INFO
    return template.replace("[DISPLAY_SPANS]", content).replace("[RELEVANT_SENTENCES]", content)


class StaticVerbatimPipeline:
    def __init__(self, index, extractor, k: int = 5, max_display_spans: int = 5, reranker=None):
        self.index, self.extractor, self.k, self.max_display_spans = index, extractor, k, max_display_spans
        self.reranker = reranker
        self.response_builder = ResponseBuilder()

    def _apply_reranker(self, question: str, results: list):
        """core.py:125-132: optional hook between retrieval and extraction; a failing reranker keeps the order."""
        if not self.reranker:
            return results
        try:
            return self.reranker.rerank(question, results)
        except Exception as exc:
            import logging

            logging.warning(f"Reranker failed, using original order: {exc}")
            return results

    def query(self, question: str, k: Optional[int] = None, filter: Optional[str] = None,
              hybrid_weights: Optional[Dict[str, float]] = None, rrf_k: int = 60) -> QueryResponse:
        results = self.index.query(text=question, k=k or self.k, filter=filter, hybrid_weights=hybrid_weights, rrf_k=rrf_k)
        results = self._apply_reranker(question, results)                            # core.py:246
        spans = self.extractor.extract_spans(question, results)                      # core.py:255
        flat = [{"text": s, "doc_text": t} for t, ss in spans.items() for s in ss]   # core.py:184-193
        display = flat[: self.max_display_spans]
        answer = self.response_builder.clean_answer(fill_static_template(display))  # core.py:261-264
        # core.py:266-272 passes the number of *documents* in the dict as display_span_count (quirk kept)
        return self.response_builder.build_response(question=question, answer=answer, search_results=results,
                                                    relevant_spans=spans, display_span_count=len(spans))

    def query_batch(self, questions, k: Optional[int] = None, filter: Optional[str] = None,
                    hybrid_weights: Optional[Dict[str, float]] = None, rrf_k: int = 60):
        """`[query(q, ...) for q in questions]` with retrieval and extraction batched across the queries
        (HotPathIndex.query_batch, extract_spans_batch) -- the serving shape of BASELINE configs[2]/[4]."""
        questions = list(questions)
        if hasattr(self.index, "query_batch"):
            per_q = self.index.query_batch(questions, k=k or self.k, filter=filter, hybrid_weights=hybrid_weights, rrf_k=rrf_k)
        else:
            per_q = [self.index.query(text=q, k=k or self.k, filter=filter, hybrid_weights=hybrid_weights, rrf_k=rrf_k)
                     for q in questions]
        per_q = [self._apply_reranker(q, r) for q, r in zip(questions, per_q)]
        if hasattr(self.extractor, "extract_spans_batch"):
            spans_per_q = self.extractor.extract_spans_batch(questions, per_q)
        else:
            spans_per_q = [self.extractor.extract_spans(q, r) for q, r in zip(questions, per_q)]
        out = []
        for question, results, spans in zip(questions, per_q, spans_per_q):
            flat = [{"text": s, "doc_text": t} for t, ss in spans.items() for s in ss]
            answer = self.response_builder.clean_answer(fill_static_template(flat[: self.max_display_spans]))
            out.append(self.response_builder.build_response(question=question, answer=answer, search_results=results,
                                                            relevant_spans=spans, display_span_count=len(spans)))
        return out
