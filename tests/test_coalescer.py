"""CPU: the coalescing scheduler in front of a batch-capable extractor.  The inner extractor
is a deterministic fake (no GPU): results must equal per-call extraction, concurrent callers must share
batches, and a failing batch must not take its neighbours down."""
import asyncio
import threading
import time
import types

import pytest

import verbatim_rag_amd  # noqa: F401
from verbatim_rag_amd.extractors import CoalescingSpanExtractor, SpanExtractor


class FakeBatchExtractor(SpanExtractor):
    def __init__(self, fail_on=None, delay=0.0):
        self.calls, self.batch_sizes, self.fail_on, self.delay = 0, [], fail_on, delay
        self._lock = threading.Lock()

    def _one(self, q, results):
        if self.fail_on is not None and q == self.fail_on:
            raise ValueError("boom")
        return {getattr(r, "text", ""): [w for w in getattr(r, "text", "").split() if w.startswith(q[:1])] for r in results}

    def extract_spans(self, question, search_results):
        return self._one(question, search_results)

    def extract_spans_batch(self, questions, results_per_question):
        with self._lock:
            self.calls += 1
            self.batch_sizes.append(len(questions))
        if self.delay:
            time.sleep(self.delay)
        return [self._one(q, r) for q, r in zip(questions, results_per_question)]


def _results(i):
    return [types.SimpleNamespace(text=f"alpha beta {i} apple"), types.SimpleNamespace(text=f"banana avocado {i}")]


def test_results_equal_per_call_and_batches_are_shared():
    inner = FakeBatchExtractor(delay=0.01)
    co = CoalescingSpanExtractor(inner, max_wait_ms=30.0, max_pairs=1000)
    try:
        n = 24
        outs = [None] * n
        qs = ["a question" if i % 2 else "b question" for i in range(n)]

        def work(i):
            outs[i] = co.extract_spans(qs[i], _results(i))

        ts = [threading.Thread(target=work, args=(i,)) for i in range(n)]
        [t.start() for t in ts]
        [t.join() for t in ts]
        for i in range(n):
            assert outs[i] == inner.extract_spans(qs[i], _results(i))
        assert co.queries_served == n
        assert inner.calls < n and max(inner.batch_sizes) > 1          # callers were coalesced
        assert list(outs[3].keys()) == [r.text for r in _results(3)]   # dict order = result order
    finally:
        co.close()


def test_max_pairs_splits_batches_and_single_call_is_not_delayed_forever():
    inner = FakeBatchExtractor()
    co = CoalescingSpanExtractor(inner, max_wait_ms=1.0, max_pairs=4)   # 2 chunks per query -> <= 2 queries per batch
    try:
        t0 = time.monotonic()
        assert co.extract_spans("a", _results(0)) == inner.extract_spans("a", _results(0))
        assert time.monotonic() - t0 < 1.0
        outs = []
        ts = [threading.Thread(target=lambda i=i: outs.append(co.extract_spans("a", _results(i)))) for i in range(10)]
        [t.start() for t in ts]
        [t.join() for t in ts]
        assert len(outs) == 10 and max(inner.batch_sizes) <= 2
    finally:
        co.close()


def test_failing_query_does_not_fail_its_batch_neighbours():
    inner = FakeBatchExtractor(fail_on="x bad")
    co = CoalescingSpanExtractor(inner, max_wait_ms=40.0)
    try:
        res = {}

        def work(q, i):
            try:
                res[i] = co.extract_spans(q, _results(i))
            except Exception as exc:
                res[i] = exc

        ts = [threading.Thread(target=work, args=(q, i)) for i, q in enumerate(["a ok", "x bad", "b ok"])]
        [t.start() for t in ts]
        [t.join() for t in ts]
        assert isinstance(res[1], ValueError)
        assert res[0] == inner.extract_spans("a ok", _results(0)) and res[2] == inner.extract_spans("b ok", _results(2))
    finally:
        co.close()


def test_async_variant_and_close():
    inner = FakeBatchExtractor()
    co = CoalescingSpanExtractor(inner, max_wait_ms=5.0)

    async def go():
        return await asyncio.gather(*[co.extract_spans_async("a", _results(i)) for i in range(6)])

    outs = asyncio.run(go())
    assert [o == inner.extract_spans("a", _results(i)) for i, o in enumerate(outs)] == [True] * 6
    co.close()
    with pytest.raises(RuntimeError):
        co.extract_spans("a", _results(0))
    with pytest.raises(TypeError):
        CoalescingSpanExtractor(object())
