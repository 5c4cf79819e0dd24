"""Host-side input packing for the extractor path (stays in Python like the reference).

Mirrors, behaviour for behaviour:
  * ModelSpanExtractor._split_into_sentences          packages/core/verbatim_core/extractors.py:190-195
  * QADataset.encode_question_and_sentences_with_offsets
                                                       packages/core/verbatim_core/extractor_models/dataset.py:109-243
    ([CLS] question [SEP] s1 [SEP] s2 ... [SEP]; inclusive token ranges; budget max_length-2;
     sentences that do not fit are dropped with the reference's warning text)
The packed ids of several (question, chunk) pairs are then handed to the GPU as ONE
padding-free batch instead of the reference's batch-size-1 loop (extractors.py:233-268).
"""
from __future__ import annotations

import logging
import re
from dataclasses import dataclass
from typing import Any, List, Optional, Sequence, Tuple

logger = logging.getLogger(__name__)

_SENT_SPLIT = re.compile(r"(?<=[.!?])\s+")


def split_into_sentences(text: str) -> List[str]:
    """extractors.py:190-195 -- regex split, strip, drop empties."""
    return [s.strip() for s in _SENT_SPLIT.split(text) if s.strip()]


def split_into_sentences_batch(texts: Sequence[str], device: int = 0, cap: int = 64) -> List[List[str]]:
    """`[split_into_sentences(t) for t in texts]` for a batch of chunk texts with the boundaries found on the GPU
    (`vrag_split_sentences`, csrc/text.hip: one lane scans one document's UTF-8 bytes) -- the ingest-time form of the
    reference's per-chunk, per-query regex split.  A document with more than `cap` sentences is re-split by
    the regex here; the result is identical either way (tests/test_text_gpu.py)."""
    import ctypes as C

    import numpy as np

    from . import _lib

    texts = list(texts)
    if not texts:
        return []
    lib = _lib.load()
    _lib.require_gpu()
    raw, unencodable = [], set()
    for d, t in enumerate(texts):
        try:
            raw.append(t.encode("utf-8"))
        except UnicodeEncodeError:      # lone surrogates: the regex path accepts such text, so this document takes it
            raw.append(b"")
            unencodable.add(d)
    off = np.zeros(len(raw) + 1, np.int64)
    np.cumsum([len(b) for b in raw], out=off[1:])
    blob = b"".join(raw) or b"\x00"
    counts = np.empty(len(raw), np.int32)
    starts = np.empty((len(raw), cap), np.int32)
    ends = np.empty((len(raw), cap), np.int32)
    buf = (C.c_char * len(blob)).from_buffer_copy(blob)
    _lib.check("vrag_split_sentences", lib.vrag_split_sentences(
        C.cast(buf, C.c_void_p), off.ctypes.data_as(C.POINTER(C.c_int64)), len(raw), cap, counts.ctypes.data_as(C.POINTER(C.c_int32)),
        starts.ctypes.data_as(C.POINTER(C.c_int32)), ends.ctypes.data_as(C.POINTER(C.c_int32)), device))
    out: List[List[str]] = []
    for d, (t, b) in enumerate(zip(texts, raw)):
        n = int(counts[d])
        if n > cap or d in unencodable:
            out.append(split_into_sentences(t))
        elif len(b) == len(t):           # pure ASCII: byte offsets are character offsets
            out.append([t[a:z] for a, z in zip(starts[d, :n].tolist(), ends[d, :n].tolist())])
        else:
            out.append([b[a:z].decode("utf-8") for a, z in zip(starts[d, :n].tolist(), ends[d, :n].tolist())])
    return out


class TokenizerAdapter:
    """Uniform `ids(text, add_special_tokens, max_length)` over a HF fast tokenizer
    (transformers) or a raw `tokenizers.Tokenizer`; truncation like the reference's
    `encode_plus(..., max_length=, truncation=True)` (dataset.py:131-137,161-167)."""

    def __init__(self, tokenizer: Any, sep_token_id: Optional[int] = None, cls_token_id: Optional[int] = None):
        self.tok = tokenizer
        self._raw = hasattr(tokenizer, "encode_batch") and not hasattr(tokenizer, "batch_encode_plus") \
            and tokenizer.__class__.__module__.startswith("tokenizers")
        sid = getattr(tokenizer, "sep_token_id", None) if not self._raw else None
        if sep_token_id is not None:
            sid = sep_token_id
        if sid is None and self._raw:
            sid = tokenizer.token_to_id("[SEP]")
        if sid is None:
            raise ValueError("tokenizer has no sep_token_id; pass sep_token_id=")
        self.sep_token_id = int(sid)
        cid = cls_token_id
        if cid is None:
            cid = tokenizer.token_to_id("[CLS]") if self._raw else getattr(tokenizer, "cls_token_id", None)
        self.cls_token_id = None if cid is None else int(cid)

    @classmethod
    def for_model(cls, tokenizer: Any, shape: Any) -> "TokenizerAdapter":
        """Special-token ids from the tokenizer when it defines them, otherwise from the model's config.json (`shape`): a
        raw `tokenizers.Tokenizer`, or an HF fast tokenizer loaded from a directory that holds only tokenizer.json,
        knows the template but not which ids are [CLS] / [SEP]."""
        def override(name):
            return None if isinstance(getattr(tokenizer, name, None), int) else getattr(shape, name, None)

        return cls(tokenizer, sep_token_id=override("sep_token_id"), cls_token_id=override("cls_token_id"))

    def ids(self, text: str, add_special_tokens: bool, max_length: int) -> List[int]:
        if self._raw:
            # `[CLS] $A [SEP]` template applied here so that truncation keeps the special tokens
            # and cuts content from the right, like HF `truncation=True` on a single sequence.
            body = list(self.tok.encode(text, add_special_tokens=False).ids)
            if not add_special_tokens:
                return body[:max_length]
            if self.cls_token_id is None:
                raise ValueError("raw tokenizer without a [CLS] id; pass cls_token_id=")
            return [self.cls_token_id] + body[: max(0, max_length - 2)] + [self.sep_token_id]
        enc = self.tok(text, add_special_tokens=add_special_tokens, max_length=max_length, truncation=True)
        return list(enc["input_ids"])

    def ids_batch(self, texts: Sequence[str], max_length: int) -> List[List[int]]:
        """Sentences are tokenised independently with add_special_tokens=False (dataset.py:161-167),
        so one batched call is bit-identical to the reference's per-sentence calls."""
        if not texts:
            return []
        if self._raw:
            return [list(e.ids)[:max_length] for e in self.tok.encode_batch(list(texts), add_special_tokens=False)]
        enc = self.tok(list(texts), add_special_tokens=False, max_length=max_length, truncation=True)
        return [list(x) for x in enc["input_ids"]]


@dataclass
class PackedSample:
    input_ids: List[int]
    sentence_boundaries: List[Tuple[int, int]]  # inclusive token ranges
    n_sentences_in: int = 0


def encode_question_and_sentences(
    question_ids_with_special: Sequence[int],
    sentence_ids: Sequence[Sequence[int]],
    sep_token_id: int,
    max_length: int = 512,
) -> PackedSample:
    """dataset.py:127-243 on pre-tokenised pieces.

    `question_ids_with_special` = tokenizer(question, add_special_tokens=True, truncation to
    max_length-2); `sentence_ids[i]` = tokenizer(sentence_i, add_special_tokens=False, same truncation).
    """
    n_in = len(sentence_ids)
    budget = max_length - 2                                   # dataset.py:127
    input_ids = list(question_ids_with_special)
    if len(input_ids) > 1 and input_ids[-1] == sep_token_id:  # dataset.py:142-145
        input_ids.pop()
    boundaries: List[Tuple[int, int]] = []
    for sent in sentence_ids:
        if len(input_ids) + len(sent) + 1 > budget:           # dataset.py:172-179
            logger.warning(
                "Legacy QA input exceeded the %d-token budget; dropping %d sentence(s)",
                budget + 2,
                n_in - len(boundaries),
            )
            break
        input_ids.append(sep_token_id)                        # dataset.py:183
        start = len(input_ids)
        input_ids.extend(sent)
        boundaries.append((start, len(input_ids) - 1))        # inclusive end, dataset.py:195-198
    if len(input_ids) < budget:                               # dataset.py:202-205
        input_ids.append(sep_token_id)
    if len(input_ids) > budget:                               # dataset.py:210-227 (only reachable when
        last_valid = 0                                        # the question alone exceeds the budget)
        for i, (_s, e) in enumerate(boundaries):
            if e < budget:
                last_valid = i
            else:
                break
        if boundaries:
            last_tok = boundaries[last_valid][1]
            input_ids = input_ids[: last_tok + 1]
            boundaries = boundaries[: last_valid + 1]
    return PackedSample(input_ids=input_ids, sentence_boundaries=boundaries, n_sentences_in=n_in)


def valid_boundaries(boundaries: Sequence[Tuple[int, int]], seq_len: int) -> List[Tuple[int, int]]:
    """QAModel.forward's range handling (extractor_models/model.py:88-96): clamp end to S-1,
    skip `end < start` or `start < 0` (skipped rows shift later logits up, reproduced as-is)."""
    out = []
    for s, e in boundaries:
        if e >= seq_len:
            e = seq_len - 1
        if e < s or s < 0:
            continue
        out.append((s, e))
    return out

