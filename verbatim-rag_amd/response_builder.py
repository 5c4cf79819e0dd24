"""Citation offsets: host-side restatement of ResponseBuilder
(packages/core/verbatim_core/response_builder.py:32-151) and of the result models
(verbatim_core/models.py:13-52) as plain dataclasses whose `model_dump()` equals the pydantic
dump.  Offsets are Python code-point indices from `str.find`, first-come-wins overlap
suppression -- kept in Python exactly like the reference."""
from __future__ import annotations

import re
from dataclasses import asdict, dataclass, field
from typing import Any, Dict, List, Optional, Set, Tuple


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


class ResponseBuilder:
    def build_response(self, question: str, answer: str, search_results: List[Any],
                       relevant_spans: Dict[str, List[str]], display_span_count: Optional[int] = None) -> QueryResponse:
        """response_builder.py:32-103."""
        docs, citations = [], []
        number = 1
        for result_index, result in enumerate(search_results):
            content = getattr(result, "text", "")
            highlights: List[Highlight] = []
            spans = relevant_spans.get(content, [])
            if spans:
                highlights = self._create_highlights(content, spans)
                for hi, h in enumerate(highlights):
                    is_display = display_span_count is None or number <= display_span_count
                    citations.append(Citation(text=h.text, doc_index=result_index, highlight_index=hi, number=number,
                                              type="display" if is_display else "reference"))
                    number += 1
            docs.append(DocumentWithHighlights(
                content=content, highlights=highlights,
                title=getattr(result, "title", "") or result.metadata.get("title", ""),
                source=getattr(result, "source", "") or result.metadata.get("source", ""),
                metadata=getattr(result, "metadata", {})))
        return QueryResponse(question=question, answer=answer,
                             structured_answer=StructuredAnswer(text=answer, citations=citations), documents=docs)

    def _create_highlights(self, doc_content: str, spans: List[str]) -> List[Highlight]:
        """response_builder.py:105-136: every non-overlapping occurrence, left to right."""
        highlights: List[Highlight] = []
        regions: Set[Tuple[int, int]] = set()
        for span in spans:
            start = 0
            while True:
                start = doc_content.find(span, start)
                if start == -1:
                    break
                end = start + len(span)
                if not self._has_overlap(start, end, regions):
                    highlights.append(Highlight(text=span, start=start, end=end))
                    regions.add((start, end))
                start = end
        return highlights

    @staticmethod
    def _has_overlap(start: int, end: int, regions: Set[Tuple[int, int]]) -> bool:
        return any(start < r_end and end > r_start for r_start, r_end in regions)

    @staticmethod
    def clean_answer(answer: str) -> str:
        """response_builder.py:153-182."""
        if not answer:
            return ""
        if answer.startswith('"') and answer.endswith('"'):
            answer = answer[1:-1]
        elif answer.startswith("'") and answer.endswith("'"):
            answer = answer[1:-1]
        answer = answer.replace("\\n", "\n")
        answer = re.sub(r" {2,}", " ", answer)
        answer = re.sub(r"\n{3,}", "\n\n", answer)
        return answer.strip()
