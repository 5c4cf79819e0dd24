"""Citation offsets on the host: the role of the reference's ResponseBuilder
(packages/core/verbatim_core/response_builder.py:32-182) and result models (verbatim_core/models.py:13-52, here plain
dataclasses whose `model_dump()` equals the pydantic dump).  Offsets are Python code-point indices from `str.find` with
first-come-first-served overlap suppression; the search keeps its kept ranges sorted instead of scanning them."""
from __future__ import annotations

import bisect
import re
from dataclasses import asdict, dataclass, field
from typing import Any, Dict, List, Optional


@dataclass
class Highlight:
    text: str
    start: int
    end: int

    def __post_init__(self):
        if not self.text or self.start < 0 or self.end <= self.start:
            raise ValueError("end must be greater than start")  # models.py:18-22

    def model_dump(self):
        return asdict(self)


@dataclass
class Citation:
    text: str
    doc_index: int
    highlight_index: int
    number: Optional[int] = None
    type: Optional[str] = None

    def model_dump(self):
        return asdict(self)


@dataclass
class DocumentWithHighlights:
    content: str
    highlights: List[Highlight] = field(default_factory=list)
    title: str = ""
    source: str = ""
    metadata: Dict[str, Any] = field(default_factory=dict)

    def model_dump(self):
        return asdict(self)


@dataclass
class StructuredAnswer:
    text: str
    citations: List[Citation] = field(default_factory=list)

    def model_dump(self):
        return asdict(self)


@dataclass
class QueryResponse:
    question: str
    answer: str
    structured_answer: StructuredAnswer
    documents: List[DocumentWithHighlights] = field(default_factory=list)

    def model_dump(self):
        return asdict(self)


def find_highlights(doc_content: str, spans: List[str]) -> List[Highlight]:
    """Character ranges of the extracted spans inside their chunk: spans in the given order, for each span every
    occurrence from left to right (the search resumes at the end of a match, so occurrences of one span never overlap
    each other), an occurrence being kept unless it overlaps a range kept earlier -- first come, first served.  The kept
    ranges are pairwise disjoint by construction, so they live in a list sorted by start and an overlap test is one
    bisection plus a look at the two neighbours (the reference scans a set of all ranges, response_builder.py:105-151;
    same result, pinned by the `highlights` fixture and the reference's own known answers)."""
    starts: List[int] = []                  # sorted starts of the kept ranges
    ends: List[int] = []                    # ends[i] belongs to starts[i]
    kept: List[Highlight] = []
    for span in spans:
        width = len(span)
        at = doc_content.find(span)
        while at != -1:
            stop = at + width
            i = bisect.bisect_left(starts, at)
            clear_left = i == 0 or ends[i - 1] <= at
            clear_right = i == len(starts) or starts[i] >= stop
            if clear_left and clear_right:
                kept.append(Highlight(text=span, start=at, end=stop))      # an empty span fails Highlight's own check
                starts.insert(i, at)
                ends.insert(i, stop)
            at = doc_content.find(span, stop)
    return kept


_QUOTES = ('"', "'")
_RUNS = ((re.compile(r" {2,}"), " "), (re.compile(r"\n{3,}"), "\n\n"))


class ResponseBuilder:
    """Search results + extracted spans -> `QueryResponse` (packages/core/verbatim_core/response_builder.py:32-182):
    one `DocumentWithHighlights` per result, one numbered `Citation` per highlight in document order; citations numbered
    above `display_span_count` are typed "reference" instead of "display"."""

    def build_response(self, question: str, answer: str, search_results: List[Any],
                       relevant_spans: Dict[str, List[str]], display_span_count: Optional[int] = None) -> QueryResponse:
        per_doc = [self._create_highlights(text, relevant_spans.get(text, []))
                   for text in (getattr(r, "text", "") for r in search_results)]
        citations: List[Citation] = []
        for doc_index, highlights in enumerate(per_doc):
            for highlight_index, h in enumerate(highlights):
                number = len(citations) + 1
                shown = display_span_count is None or number <= display_span_count
                citations.append(Citation(text=h.text, doc_index=doc_index, highlight_index=highlight_index, number=number,
                                          type="display" if shown else "reference"))
        documents = [DocumentWithHighlights(content=getattr(r, "text", ""), highlights=highlights,
                                            title=getattr(r, "title", "") or r.metadata.get("title", ""),
                                            source=getattr(r, "source", "") or r.metadata.get("source", ""),
                                            metadata=getattr(r, "metadata", {}))
                     for r, highlights in zip(search_results, per_doc)]
        return QueryResponse(question=question, answer=answer,
                             structured_answer=StructuredAnswer(text=answer, citations=citations), documents=documents)

    def _create_highlights(self, doc_content: str, spans: List[str]) -> List[Highlight]:
        return find_highlights(doc_content, spans) if spans else []

    @staticmethod
    def _has_overlap(start: int, end: int, regions) -> bool:
        """Half-open interval test against a collection of (start, end) pairs (kept for callers of the reference's helper)."""
        return any(lo < end and start < hi for lo, hi in regions)

    @staticmethod
    def clean_answer(answer: str) -> str:
        """Tidies generated answer text (response_builder.py:153-182): one pair of surrounding quotes dropped, literal
        backslash-n turned into newlines, runs of spaces collapsed to one and runs of 3+ newlines to a paragraph break."""
        if not answer:
            return ""
        if answer[0] in _QUOTES and answer[-1] == answer[0]:
            answer = answer[1:-1]
        answer = answer.replace("\\n", "\n")
        for pattern, repl in _RUNS:
            answer = pattern.sub(repl, answer)
        return answer.strip()
