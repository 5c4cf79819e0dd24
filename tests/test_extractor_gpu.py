"""GPU drop-in extractor vs golden outputs captured from the reference's ModelSpanExtractor
(legacy qa_model path) -- through the C ABI, on the committed fixtures."""
import json
import os
import threading
import types

import numpy as np
import pytest

from oracle import modernbert_np as O

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")
TINY = dict(vocab_size=512, hidden_size=128, num_hidden_layers=4, num_attention_heads=2, intermediate_size=192,
            pad_token_id=0, cls_token_id=1, sep_token_id=2)


def _shape(cfg):
    from verbatim_rag_amd.engine import ModernBertShape

    return ModernBertShape(vocab_size=cfg.vocab_size, hidden_size=cfg.hidden_size, num_hidden_layers=cfg.num_hidden_layers,
                           num_attention_heads=cfg.num_attention_heads, intermediate_size=cfg.intermediate_size,
                           pad_token_id=cfg.pad_token_id, cls_token_id=cfg.cls_token_id, sep_token_id=cfg.sep_token_id)


@pytest.fixture(scope="module")
def setup():
    from tokenizers import Tokenizer

    from verbatim_rag_amd.engine import EncoderEngine
    from verbatim_rag_amd.extractors import GpuModelSpanExtractor

    cfg = O.EncoderConfig(**TINY)
    w = O.random_weights(cfg, seed=7)
    z = np.load(os.path.join(G, "encoder_tiny.npz"))
    eng = EncoderEngine(_shape(cfg), w, max_tokens=8192, max_seqs=64, max_seq_len=512, max_ranges=1024)
    eng.set_qa_head(z["qa_Wc"], z["qa_bc"])
    tok = Tokenizer.from_file(os.path.join(G, "tokenizer.json"))
    ext = GpuModelSpanExtractor(engine=eng, tokenizer=tok, threshold=0.5)
    with open(os.path.join(G, "host_fixtures.json")) as f:
        fx = json.load(f)
    yield cfg, w, z, eng, ext, fx
    eng.close()


def test_golden_hidden_and_logits(setup):
    cfg, w, z, eng, ext, fx = setup
    for S in (7, 64, 130, 200):
        eng.load_batch([z[f"ids_{S}"]])
        eng.run()
        got = eng.read_hidden(True)
        assert np.abs(got - z[f"hidden_{S}"]).max() < 3e-2
    bounds = [(3, 20), (22, 60), (62, 129), (100, 129)]   # reference clamps (100,400) -> (100,129) and skips (50,10)
    lg = eng.qa_logits([z["ids_130"]], [bounds])[0]
    assert np.abs(lg - z["qa_logits_130"]).max() < 1e-3


def test_extract_spans_equals_reference_extractor(setup):
    cfg, w, z, eng, ext, fx = setup
    e = fx["extract_e2e"]
    for run in e["runs"]:
        ext.threshold = run["threshold"]
        results = [types.SimpleNamespace(text=t) for t in run["texts"]]
        got = ext.extract_spans(run["question"], results)
        assert got == run["spans"]
        assert list(got.keys()) == list(run["spans"].keys())      # dict order = result order, duplicates collapse
    # logits themselves within 1e-3 of the reference's
    ext.threshold = 0.5
    texts = e["runs"][0]["texts"]
    all_sents, samples = ext.pack_qa(e["runs"][0]["question"], texts)
    for t, smp, ref in zip(texts, samples, e["logits"]):
        if smp is None:
            assert ref == []
            continue
        lg = eng.qa_logits([smp.input_ids], [smp.sentence_boundaries])[0]
        assert np.abs(lg - np.asarray(ref, np.float32)).max() < 1e-3


def test_spans_are_exact_substrings_and_offsets(setup):
    from verbatim_rag_amd.response_builder import ResponseBuilder

    cfg, w, z, eng, ext, fx = setup
    c = fx["config1"]
    ext.threshold = c["threshold"]
    results = [types.SimpleNamespace(text=t, metadata={"title": f"Doc {i}", "source": f"src{i}.md"})
               for i, t in enumerate(c["docs"])]
    spans = ext.extract_spans(c["question"], results)
    resp = ResponseBuilder().build_response(c["question"], c["response"]["answer"], results, spans,
                                            display_span_count=len(spans))
    # citation offsets bit-exact vs the reference pipeline's QueryResponse
    assert [d["highlights"] for d in resp.model_dump()["documents"]] == [d["highlights"] for d in c["response"]["documents"]]
    assert resp.model_dump()["structured_answer"]["citations"] == c["response"]["structured_answer"]["citations"]


def test_empty_and_blank_inputs(setup):
    cfg, w, z, eng, ext, fx = setup
    assert ext.extract_spans("q?", []) == {}
    got = ext.extract_spans("q?", [types.SimpleNamespace(text=""), types.SimpleNamespace(text="   "),
                                   types.SimpleNamespace()])
    assert got == {"": [], "   ": []}


def test_threaded_callers_like_asyncio_to_thread(setup):
    cfg, w, z, eng, ext, fx = setup
    run = fx["extract_e2e"]["runs"][0]
    ext.threshold = run["threshold"]
    results = [types.SimpleNamespace(text=t) for t in run["texts"]]
    out, errs = [None] * 8, []

    def work(i):
        try:
            out[i] = ext.extract_spans(run["question"], results)
        except Exception as e:  # pragma: no cover
            errs.append(e)

    ts = [threading.Thread(target=work, args=(i,)) for i in range(8)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errs and all(o == run["spans"] for o in out)


def test_micro_batching_is_bit_identical(setup):
    from verbatim_rag_amd.engine import EncoderEngine

    cfg, w, z, eng, ext, fx = setup
    rng = np.random.default_rng(9)
    seqs = [rng.integers(3, 400, size=n).astype(np.int32) for n in (300, 17, 256, 129, 511, 64, 200)]
    eng.load_batch(seqs)
    eng.run()
    ref = eng.read_hidden(True)
    e2 = EncoderEngine(_shape(cfg), w, max_tokens=8192, max_seqs=64, max_seq_len=512, max_ranges=64, micro_batch_tokens=600)
    e2.load_batch(seqs)
    e2.run()
    got = e2.read_hidden(True)
    e2.close()
    assert np.array_equal(ref, got)


def test_capacity_and_argument_errors(setup):
    from verbatim_rag_amd._lib import VragError

    cfg, w, z, eng, ext, fx = setup
    with pytest.raises(VragError, match="capacity"):
        eng.load_batch([np.ones(500, np.int32)] * 20)           # 10000 tokens > 8192
    with pytest.raises(VragError, match="outside the vocabulary"):
        eng.load_batch([np.asarray([1, 9999, 2], np.int32)])
    with pytest.raises(VragError, match="length"):
        eng.load_batch([np.ones(513, np.int32)])
    eng.load_batch([np.ones(10, np.int32)])
    with pytest.raises(VragError, match="not inside sequence"):
        eng.load_ranges([0], [3], [10])


def test_base_config_logits_vs_transformers_golden():
    """ModernBERT-base shapes: GPU sentence logits within 1e-3 of fp32 transformers (golden)."""
    from verbatim_rag_amd.engine import EncoderEngine, ModernBertShape
    from verbatim_rag_amd.weights import random_init, random_qa_head

    z = np.load(os.path.join(G, "encoder_base.npz"))
    shape = ModernBertShape.base()
    eng = EncoderEngine(shape, random_init(shape, seed=1234), max_tokens=2048, max_seqs=4, max_seq_len=512, max_ranges=64)
    eng.set_qa_head(*random_qa_head(shape))
    seqs = [z["ids_0"], z["ids_1"]]
    bounds = [[tuple(b) for b in z["bounds_0"].tolist()], [tuple(b) for b in z["bounds_1"].tolist()]]
    got = eng.qa_logits(seqs, bounds)
    hid = eng.read_hidden(True)
    eng.close()
    for n in range(2):
        err = np.abs(got[n] - z[f"logits_{n}"]).max()
        assert err < 1e-3, f"seq {n}: sentence-logit max-abs error {err}"
    assert np.abs(hid[:4] - z["hidden_rows_0"]).max() < 5e-2


def test_cross_query_batching_equals_per_query_calls(setup):
    cfg, w, z, eng, ext, fx = setup
    run = fx["extract_e2e"]["runs"][0]
    ext.threshold = run["threshold"]
    results = [types.SimpleNamespace(text=t) for t in run["texts"]]
    qs = [run["question"], "Who built the iron bridge?", run["question"]]
    rs = [results, results[:3], results[2:]]
    batched = ext.extract_spans_batch(qs, rs)
    assert batched == [ext.extract_spans(q, r) for q, r in zip(qs, rs)]
    assert batched[0] == run["spans"]


def test_two_handles_alternate_sub_batches_bit_identically(setup):
    """Small workspaces force many sub-batches; two handles of the same model run them from two worker threads."""
    from verbatim_rag_amd.engine import EncoderEngine
    from verbatim_rag_amd.extractors import GpuModelSpanExtractor

    cfg, w, z, eng, ext, fx = setup
    run = fx["extract_e2e"]["runs"][0]
    ext.threshold = run["threshold"]
    engs = [EncoderEngine(_shape(cfg), w, max_tokens=1024, max_seqs=8, max_seq_len=512, max_ranges=256) for _ in range(2)]
    try:
        for e in engs:
            e.set_qa_head(z["qa_Wc"], z["qa_bc"])
        two = GpuModelSpanExtractor(engine=engs[0], extra_engines=engs[1:], tokenizer=ext.tokenizer, threshold=run["threshold"])
        results = [types.SimpleNamespace(text=t) for t in run["texts"]]
        qs = [run["question"], "Who built the iron bridge?", "When was it opened?"] * 6
        rs = [results, results[:3], results[2:]] * 6
        got = two.extract_spans_batch(qs, rs)
        assert got == ext.extract_spans_batch(qs, rs) and got[0] == run["spans"]
    finally:
        for e in engs:
            e.close()


def test_coalescing_scheduler_on_the_gpu_extractor(setup):
    """Concurrent per-query calls (the reference's to_thread pattern) coalesced into shared GPU batches."""
    import threading

    from verbatim_rag_amd.extractors import CoalescingSpanExtractor

    cfg, w, z, eng, ext, fx = setup
    run = fx["extract_e2e"]["runs"][0]
    ext.threshold = run["threshold"]
    results = [types.SimpleNamespace(text=t) for t in run["texts"]]
    qs = [run["question"], "Who built the iron bridge?", "When was it opened?", run["question"]] * 4
    rs = [results, results[:3], results[2:], results[1:4]] * 4
    want = [ext.extract_spans(q, r) for q, r in zip(qs, rs)]
    co = CoalescingSpanExtractor(ext, max_wait_ms=20.0)
    try:
        got = [None] * len(qs)

        def work(i):
            got[i] = co.extract_spans(qs[i], rs[i])

        ts = [threading.Thread(target=work, args=(i,)) for i in range(len(qs))]
        [t.start() for t in ts]
        [t.join() for t in ts]
        assert got == want
        assert co.batches_run < len(qs)
    finally:
        co.close()


def test_concurrent_threads_on_separate_and_shared_handles(setup):
    """SURVEY 8b threading contract: callers arrive from asyncio.to_thread workers.  Four threads hammer their own
    engines while four more share ONE engine; every result must equal the single-threaded one."""
    import threading

    from verbatim_rag_amd.engine import EncoderEngine

    cfg, w, z, eng, ext, fx = setup
    shape = eng.shape
    rng = np.random.default_rng(77)
    work = []
    for _ in range(6):
        seqs = [rng.integers(3, cfg.vocab_size, size=int(n)).astype(np.int32) for n in rng.integers(5, 300, size=int(rng.integers(1, 9)))]
        work.append((seqs, [[(1, len(s) - 1)] for s in seqs]))
    qa_w = (rng.standard_normal((2, cfg.hidden_size)) * 0.1).astype(np.float32)
    qa_b = np.zeros(2, np.float32)

    def fresh():
        e = EncoderEngine(shape, w, max_tokens=4096, max_seqs=16, max_seq_len=512, max_ranges=64)
        e.set_qa_head(qa_w, qa_b)
        return e

    ref_eng = fresh()
    want = [np.concatenate(ref_eng.qa_logits(s, b)) for s, b in work]
    ref_eng.close()
    shared = fresh()
    engines = [fresh() for _ in range(4)] + [shared] * 4
    errors = []

    def run(e):
        try:
            for _rep in range(5):
                for (s, b), ref in zip(work, want):
                    got = np.concatenate(e.qa_logits(s, b))
                    if not np.array_equal(got, ref):
                        errors.append(float(np.abs(got - ref).max()))
        except Exception as exc:   # pragma: no cover
            errors.append(repr(exc))

    ts = [threading.Thread(target=run, args=(e,)) for e in engines]
    [t.start() for t in ts]
    [t.join() for t in ts]
    for e in engines[:4]:
        e.close()
    shared.close()
    assert not errors, errors[:3]


def test_constructor_from_a_model_directory(tmp_path):
    """`GpuModelSpanExtractor(model_path)`: config.json + model.safetensors (`bert.*` + `classifier.*`) + tokenizer.json
    on disk -> the same spans as the engine= route on the same tensors, and logits within 1e-3 of the fp32 oracle."""
    from test_checkpoint_loading import make_qa_checkpoint_dir
    from tokenizers import Tokenizer

    from verbatim_rag_amd.engine import EncoderEngine
    from verbatim_rag_amd.extractors import GpuModelSpanExtractor

    m, Wc, bc = make_qa_checkpoint_dir(str(tmp_path))
    ext = GpuModelSpanExtractor(model_path=str(tmp_path), threshold=0.5)
    W = {k: v.numpy() for k, v in m.state_dict().items()}
    eng = EncoderEngine(ext.engine.shape, W, max_tokens=8192, max_seqs=64, max_seq_len=512, max_ranges=1024)
    try:
        eng.set_qa_head(Wc, bc)
        ext2 = GpuModelSpanExtractor(engine=eng, tokenizer=Tokenizer.from_file(os.path.join(G, "tokenizer.json")), threshold=0.5)
        texts = ["The tall iron tower is in paris. It was built for the world fair. Millions of visitors climb it every year.",
                 "A stone bridge crosses the river. The engineer opened it at night.", " "]
        results = [types.SimpleNamespace(text=t) for t in texts]
        q = "Where is the tall iron tower?"
        got = ext.extract_spans(q, results)
        assert got == ext2.extract_spans(q, results) and list(got) == texts and got[" "] == []
        assert all(s in t for t, spans in got.items() for s in spans)
        sents, samples = ext.pack_qa(q, texts[:1])
        vb = [tuple(b) for b in samples[0].sentence_boundaries]
        lg = ext.engine.qa_logits([samples[0].input_ids], [vb])[0]
        cfg = O.EncoderConfig(**TINY)
        ref = O.qa_sentence_logits(O.encoder_forward(cfg, W, samples[0].input_ids), vb, Wc, bc)
        assert np.abs(lg - ref).max() < 1e-3
    finally:
        eng.close()
        ext.engine.close()
