"""Known-answer vectors held by the reference's own tests for the host half of the path, as data: inputs and expected
outputs of /root/reference tests/test_response_builder.py (:12-34 highlights, :41-51 overlap, :58-75 clean_answer,
:90-140 build_response / citation numbering) and tests/test_models.py (:15-35 Highlight validation, :39-47 Citation,
:51-60 documents, :64-80 QueryResponse round trip), run against this package's restatements."""
import types

import pytest

import verbatim_rag_amd  # noqa: F401
from verbatim_rag_amd.response_builder import (Citation, DocumentWithHighlights, Highlight, QueryResponse, ResponseBuilder,
                                               StructuredAnswer)

B = ResponseBuilder()

HIGHLIGHT_KATS = [   # (document, spans, expected [(text, start, end)])
    ("The cat sat on the mat.", ["cat"], [("cat", 4, 7)]),
    ("The cat sat on the mat.", ["cat", "mat"], [("cat", 4, 7), ("mat", 19, 22)]),
    ("The cat sat.", ["dog"], []),
    ("The big cat sat.", ["big cat", "cat"], [("big cat", 4, 11)]),       # "cat" overlaps "big cat": first come wins
]
OVERLAP_KATS = [((0, 5, {(10, 15)}), False), ((3, 8, {(5, 10)}), True), ((6, 8, {(5, 10)}), True), ((0, 5, set()), False)]
CLEAN_KATS = [('"Hello world"', "Hello world"), ("'Hello world'", "Hello world"), ("line1\\nline2", "line1\nline2"),
              ("a   b", "a b"), ("a\n\n\n\nb", "a\n\nb"), ("", "")]
BAD_HIGHLIGHTS = [dict(text="hello", start=5, end=5), dict(text="hello", start=5, end=3), dict(text="hello", start=-1, end=5),
                  dict(text="", start=0, end=5)]


@pytest.mark.parametrize("doc,spans,want", HIGHLIGHT_KATS)
def test_create_highlights(doc, spans, want):
    assert [(h.text, h.start, h.end) for h in B._create_highlights(doc, spans)] == want


@pytest.mark.parametrize("args,want", OVERLAP_KATS)
def test_has_overlap(args, want):
    assert B._has_overlap(*args) is want


@pytest.mark.parametrize("raw,want", CLEAN_KATS)
def test_clean_answer(raw, want):
    assert B.clean_answer(raw) == want


def _result(text, title="", source=""):
    return types.SimpleNamespace(text=text, metadata={"title": title, "source": source}, title=title, source=source)


def test_build_response_basic_numbering_and_empty():
    r = B.build_response(question="What animal?", answer="A cat.", search_results=[_result("The cat sat on the mat.")],
                         relevant_spans={"The cat sat on the mat.": ["cat"]})
    assert (r.question, r.answer, len(r.documents), len(r.documents[0].highlights)) == ("What animal?", "A cat.", 1, 1)
    assert r.structured_answer.citations[0].text == "cat"
    r = B.build_response(question="Q", answer="A",
                         search_results=[_result("Doc one has alpha and beta."), _result("Doc two has gamma.")],
                         relevant_spans={"Doc one has alpha and beta.": ["alpha", "beta"], "Doc two has gamma.": ["gamma"]},
                         display_span_count=2)
    assert [(c.number, c.type) for c in r.structured_answer.citations] == [(1, "display"), (2, "display"), (3, "reference")]
    r = B.build_response(question="Q", answer="A", search_results=[_result("Some text.")], relevant_spans={"Some text.": []})
    assert r.documents[0].highlights == [] and r.structured_answer.citations == []


@pytest.mark.parametrize("kw", BAD_HIGHLIGHTS)
def test_highlight_validation(kw):
    with pytest.raises(ValueError):          # pydantic's ValidationError is a ValueError
        Highlight(**kw)


def test_models_defaults_and_round_trip():
    h = Highlight(text="hello", start=0, end=5)
    assert (h.text, h.start, h.end) == ("hello", 0, 5)
    c = Citation(text="span", doc_index=0, highlight_index=0, number=1, type="display")
    assert (c.number, c.type) == (1, "display")
    c = Citation(text="span", doc_index=0, highlight_index=0)
    assert c.number is None and c.type is None
    d = DocumentWithHighlights(content="Some text")
    assert d.highlights == [] and d.title == "" and d.metadata == {}
    assert len(DocumentWithHighlights(content="Some text", highlights=[Highlight(text="Some", start=0, end=4)]).highlights) == 1
    qr = QueryResponse(question="What?", answer="Answer text", structured_answer=StructuredAnswer(text="Answer text", citations=[]),
                       documents=[DocumentWithHighlights(content="Doc content")])
    assert qr.question == "What?" and len(qr.documents) == 1
    data = qr.model_dump()
    assert data["question"] == "What?" and data["documents"][0]["content"] == "Doc content" and data["structured_answer"]["citations"] == []


def test_format_detection_failure_warns_before_legacy_fallback(tmp_path, caplog):
    """tests/test_extractors.py:11-27: an unreadable config logs `Highlighter detection failed for <path>: <error>` and
    falls back to the legacy QA format; `auto_map` naming a *Highlighter* class selects the v2 path (extractors.py:135-149)."""
    import json
    import logging

    from verbatim_rag_amd.extractors import GpuModelSpanExtractor as E

    missing = str(tmp_path / "org" / "highlighter")
    with caplog.at_level(logging.WARNING):
        assert E._detect_format(missing) == E._FORMAT_QA_MODEL
    assert f"Highlighter detection failed for {missing}: " in caplog.text
    d = tmp_path / "v2"
    d.mkdir()
    (d / "config.json").write_text(json.dumps({"auto_map": {"AutoModel": "modeling_highlighter.ModernBertHighlighter"}}))
    assert E._detect_format(str(d)) == E._FORMAT_HIGHLIGHTER
    (d / "config.json").write_text(json.dumps({"model_type": "modernbert"}))
    assert E._detect_format(str(d)) == E._FORMAT_QA_MODEL
