"""ctypes binding of libvrag_amd.so (include/vrag_amd.h). Fails loudly: there is no
Python/CPU fallback for any entry point."""
from __future__ import annotations

import ctypes as C
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
_LOCK = threading.Lock()
_LIB = None

VRAG_OK = 0
ABI_VERSION = 6
PROF_CLASSES = (
    "embed", "layernorm", "gemm_qkv", "attn_global", "attn_local",
    "gemm_wo", "gemm_wi", "gemm_wo_mlp", "head", "qkv_attn_global", "qkv_attn_local",
)


class VragError(RuntimeError):
    """A libvrag_amd call returned a non-zero status."""

    def __init__(self, fn: str, status: int, message: str):
        super().__init__(f"{fn} failed (status {status}): {message}")
        self.status = status


class EncoderConfig(C.Structure):
    _fields_ = [
        ("vocab_size", C.c_int32), ("hidden_size", C.c_int32), ("num_layers", C.c_int32),
        ("num_heads", C.c_int32), ("intermediate_size", C.c_int32), ("global_every", C.c_int32),
        ("sliding_window", C.c_int32), ("rope_theta_global", C.c_float), ("rope_theta_local", C.c_float),
        ("norm_eps", C.c_float), ("pad_token_id", C.c_int32), ("max_seq_len", C.c_int32),
        ("max_tokens", C.c_int32), ("max_seqs", C.c_int32), ("max_ranges", C.c_int32),
        ("micro_batch_tokens", C.c_int32), ("device", C.c_int32), ("operand_dtype", C.c_int32),
    ]


OPERAND_DTYPES = {"bf16": 0, "f16": 1}
_FP = C.POINTER(C.c_float)
_FPP = C.POINTER(_FP)
_IP = C.POINTER(C.c_int32)
_LP = C.POINTER(C.c_int64)


class EncoderWeights(C.Structure):
    _fields_ = [
        ("tok_embeddings", _FP), ("emb_norm", _FP), ("attn_norm", _FPP), ("wqkv", _FPP), ("wo", _FPP),
        ("mlp_norm", _FPP), ("wi", _FPP), ("wo_mlp", _FPP), ("final_norm", _FP),
    ]


class BertConfig(C.Structure):
    _fields_ = [
        ("vocab_size", C.c_int32), ("hidden_size", C.c_int32), ("num_layers", C.c_int32), ("num_heads", C.c_int32),
        ("intermediate_size", C.c_int32), ("max_position_embeddings", C.c_int32), ("norm_eps", C.c_float),
        ("pad_token_id", C.c_int32), ("max_seq_len", C.c_int32), ("max_tokens", C.c_int32), ("max_seqs", C.c_int32),
        ("max_ranges", C.c_int32), ("micro_batch_tokens", C.c_int32), ("device", C.c_int32), ("operand_dtype", C.c_int32),
    ]


class BertWeights(C.Structure):
    _fields_ = [
        ("word_embeddings", _FP), ("position_embeddings", _FP), ("token_type_row", _FP), ("emb_norm_w", _FP),
        ("emb_norm_b", _FP), ("wqkv", _FPP), ("bqkv", _FPP), ("wo", _FPP), ("bo", _FPP), ("attn_norm_w", _FPP),
        ("attn_norm_b", _FPP), ("w1", _FPP), ("b1", _FPP), ("w2", _FPP), ("b2", _FPP), ("out_norm_w", _FPP),
        ("out_norm_b", _FPP),
    ]


# name -> (restype, argtypes); every symbol include/vrag_amd.h declares.
_H = C.c_void_p
SIGNATURES = {
    "vrag_last_error": (C.c_char_p, []),
    "vrag_abi_version": (C.c_int, []),
    "vrag_device_count": (C.c_int, []),
    "vrag_encoder_create": (C.c_int, [C.POINTER(EncoderConfig), C.POINTER(EncoderWeights), C.POINTER(_H)]),
    "vrag_bert_encoder_create": (C.c_int, [C.POINTER(BertConfig), C.POINTER(BertWeights), C.POINTER(_H)]),
    "vrag_encoder_destroy": (None, [_H]),
    "vrag_encoder_set_qa_head": (C.c_int, [_H, _FP, _FP, C.c_int32]),
    "vrag_encoder_set_token_head": (C.c_int, [_H, _FP, _FP, _FP, _FP, C.c_int32]),
    "vrag_encoder_set_mlm_head": (C.c_int, [_H, _FP, _FP, _FP, _FP]),
    "vrag_encoder_set_mlm_head_ex": (C.c_int, [_H, _FP, _FP, _FP, _FP, _FP, _FP]),
    "vrag_encoder_set_head_precision": (C.c_int, [_H, C.c_int32]),
    "vrag_encoder_set_token_types": (C.c_int, [_H, _FP, C.c_int32]),
    "vrag_encoder_load_token_types": (C.c_int, [_H, _IP, C.c_void_p]),
    "vrag_encoder_set_pair_head": (C.c_int, [_H, _FP, _FP, _FP, _FP, C.c_int32]),
    "vrag_encoder_run_pair_head": (C.c_int, [_H, C.c_void_p]),
    "vrag_encoder_read_pair_logits": (C.c_int, [_H, _FP, C.c_void_p]),
    "vrag_encoder_load_batch": (C.c_int, [_H, _IP, _IP, C.c_int32, C.c_void_p]),
    "vrag_encoder_run": (C.c_int, [_H, C.c_void_p]),
    "vrag_encoder_run_layers": (C.c_int, [_H, C.c_int32, C.c_void_p]),
    "vrag_encoder_load_ranges": (C.c_int, [_H, _IP, _IP, _IP, C.c_int32, C.c_void_p]),
    "vrag_encoder_run_qa_head": (C.c_int, [_H, C.c_void_p]),
    "vrag_encoder_read_qa_logits": (C.c_int, [_H, _FP, C.c_void_p]),
    "vrag_encoder_run_pool": (C.c_int, [_H, C.c_int32, C.c_void_p]),
    "vrag_encoder_read_pool": (C.c_int, [_H, _FP, C.c_void_p]),
    "vrag_encoder_run_token_head": (C.c_int, [_H, C.c_void_p]),
    "vrag_encoder_read_token_logits": (C.c_int, [_H, _FP, C.c_void_p]),
    "vrag_encoder_run_splade": (C.c_int, [_H, C.c_void_p]),
    "vrag_encoder_read_splade": (C.c_int, [_H, _FP, C.c_void_p]),
    "vrag_encoder_read_splade_sparse": (C.c_int, [_H, C.c_float, C.c_int32, _IP, _IP, _FP, C.c_void_p]),
    "vrag_encoder_read_hidden": (C.c_int, [_H, C.c_int32, _FP, C.c_void_p]),
    "vrag_encoder_extract_qa": (C.c_int, [_H, _IP, _IP, C.c_int32, _IP, _IP, _IP, C.c_int32, _FP]),
    "vrag_encoder_graph_stats": (C.c_int, [_H, C.c_int32, _LP, _IP]),
    "vrag_encoder_f16_saturated": (C.c_int, [_H, C.c_int32, _IP]),
    "vrag_encoder_set_concurrency": (C.c_int, [_H, C.c_int32]),
    "vrag_encoder_set_profiling": (C.c_int, [_H, C.c_int32]),
    "vrag_encoder_read_profile": (C.c_int, [_H, _FP, _LP, C.c_int32]),
    "vrag_set_small_batch_rows": (C.c_int, [C.c_int32]),
    "vrag_dense_index_create": (C.c_int, [C.c_int32, C.c_int64, C.c_int32, C.c_int32, C.POINTER(_H)]),
    "vrag_dense_index_destroy": (None, [_H]),
    "vrag_dense_index_size": (C.c_int64, [_H]),
    "vrag_dense_index_add": (C.c_int, [_H, _FP, C.c_int64]),
    "vrag_dense_index_add_device": (C.c_int, [_H, C.c_void_p, C.c_int64, C.c_void_p]),
    "vrag_dense_index_search": (C.c_int, [_H, _FP, C.c_int32, C.c_int32, _FP, _LP, C.c_void_p]),
    "vrag_dense_index_run_resident": (C.c_int, [_H, C.c_int32, C.c_int32, C.c_void_p]),
    "vrag_dense_index_search_device": (C.c_int, [_H, _FP, C.c_int32, C.c_int32, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p,
                                                 C.c_void_p, C.c_void_p]),
    "vrag_sparse_index_search_device": (C.c_int, [_H, _LP, _IP, _FP, C.c_int32, C.c_int32, C.c_void_p, C.c_int64, C.c_int64,
                                                  C.c_void_p, C.c_void_p, C.c_void_p]),
    "vrag_pack_qa_pairs": (C.c_int, [_IP, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, _IP, C.c_int32, C.c_int32, _IP, C.c_int64,
                                      _LP, _LP, C.c_int64, _IP, _IP, _LP]),
    "vrag_split_sentences": (C.c_int, [C.c_void_p, _LP, C.c_int32, C.c_int32, _IP, _IP, _IP, C.c_int32]),
    "vrag_topk_fill_empty": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p]),
    "vrag_sparse_index_create": (C.c_int, [C.c_int32, C.c_int64, _LP, _IP, _FP, C.c_int32, C.POINTER(_H)]),
    "vrag_sparse_index_destroy": (None, [_H]),
    "vrag_sparse_index_stats": (C.c_int, [_H, _LP, _LP, _LP]),
    "vrag_sparse_index_search": (C.c_int, [_H, _LP, _IP, _FP, C.c_int32, C.c_int32, _FP, _LP, C.c_void_p]),
    "vrag_sparse_index_run_resident": (C.c_int, [_H, C.c_int32, C.c_int32, C.c_void_p]),
    "vrag_comm_get_unique_id": (C.c_int, [C.c_void_p]),
    "vrag_comm_create": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.POINTER(_H)]),
    "vrag_comm_destroy": (None, [_H]),
    "vrag_comm_info": (C.c_int, [_H, _IP, _IP, _IP]),
    "vrag_comm_allgather": (C.c_int, [_H, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "vrag_topk_allgather_merge": (C.c_int, [_H, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
                                            C.c_void_p]),
    "vrag_topk_merge": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int64,
                                  C.c_int64, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]),
}


# include/vrag_amd_debug.h: the tuning / unit-test harness, exported by libvrag_amd_dbg.so only (load_debug()).
DEBUG_SIGNATURES = {
    "vrag_debug_gemm_ms": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _FP]),
    "vrag_debug_attn_run": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
                                      C.c_void_p, C.c_void_p, C.c_int32]),
    "vrag_debug_attn_ms": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _FP]),
    "vrag_debug_qkv_attn_ms": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _FP]),
}
_DBG = None


def library_path() -> str:
    return os.environ.get("VRAG_AMD_LIB", os.path.join(_HERE, "libvrag_amd.so"))


def _preload_torch_hip_runtime() -> None:
    """One HIP runtime per process.  PyTorch-ROCm wheels bundle their own `libamdhip64.so` and link it by that
    un-versioned name, so when libvrag_amd.so (linked against the system `libamdhip64.so.7`) is loaded BEFORE torch, the
    loader does not recognise the two as the same library and maps both: torch then finds no usable GPU, and stream
    handles could not cross between the two runtimes.  Mapping torch's copy first (without importing torch) makes
    libvrag_amd.so bind to it by SONAME -- the arrangement that `import torch` before this package produces anyway."""
    import importlib.util
    import sys

    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def load() -> C.CDLL:
    """Loads the shared library and binds every declared symbol (raises if anything is missing)."""
    global _LIB
    with _LOCK:
        if _LIB is not None:
            return _LIB
        path = library_path()
        if not os.path.exists(path):
            raise ImportError(
                f"{path} not found: build it with `python __graft_entry__.py build` "
                "(hipcc --offload-arch=gfx950). verbatim_rag_amd has no CPU fallback."
            )
        _preload_torch_hip_runtime()
        lib = C.CDLL(path)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError if the symbol is not exported
            fn.restype = res
            fn.argtypes = args
        if lib.vrag_abi_version() != ABI_VERSION:
            raise ImportError(f"{path}: ABI version {lib.vrag_abi_version()} != {ABI_VERSION}")
        _LIB = lib
        return lib


def debug_library_path() -> str:
    return os.environ.get("VRAG_AMD_DEBUG_LIB", os.path.join(_HERE, "libvrag_amd_dbg.so"))


def load_debug() -> C.CDLL:
    """The harness library (include/vrag_amd_debug.h; tools/ and the attention unit test): the product sources plus the
    synthetic-operand timing loops.  The product path never loads it."""
    global _DBG
    with _LOCK:
        if _DBG is not None:
            return _DBG
        path = debug_library_path()
        if not os.path.exists(path):
            raise ImportError(f"{path} not found: build it with `python __graft_entry__.py build`")
        _preload_torch_hip_runtime()
        lib = C.CDLL(path)
        for name, (res, args) in DEBUG_SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        lib.vrag_last_error.restype = C.c_char_p
        lib.vrag_set_small_batch_rows.restype = C.c_int     # the harness library is a full build: its own copy of the threshold
        lib.vrag_set_small_batch_rows.argtypes = [C.c_int32]
        _DBG = lib
        return lib


def check_debug(fn: str, status: int) -> None:
    if status != VRAG_OK:
        msg = load_debug().vrag_last_error()
        raise VragError(fn, status, msg.decode("utf-8", "replace") if msg else "")


def last_error() -> str:
    msg = load().vrag_last_error()
    return msg.decode("utf-8", "replace") if msg else ""


def check(fn: str, status: int) -> None:
    if status != VRAG_OK:
        raise VragError(fn, status, last_error())


def require_gpu() -> None:
    if load().vrag_device_count() <= 0:
        raise RuntimeError(
            "verbatim_rag_amd: no HIP device visible. The MI355X hot path has no CPU fallback; "
            "use the reference's own providers on machines without a GPU."
        )
