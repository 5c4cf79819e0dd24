"""The C-ABI library loads on a CPU-only box and exports every symbol include/vrag_amd.h declares."""
import ctypes
import os
import re

import pytest

import verbatim_rag_amd  # noqa: F401
from verbatim_rag_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "vrag_amd.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(vrag_[a-z0-9_]+)\s*\(", src)))


def test_every_declared_symbol_is_exported_and_bound():
    names = _declared()
    assert len(names) >= 30
    lib = ctypes.CDLL(_lib.library_path())
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/vrag_amd.h but not exported"
        assert n in _lib.SIGNATURES, f"{n} has no ctypes signature in _lib.py"
    assert sorted(_lib.SIGNATURES) == names


def test_debug_harness_is_a_separate_library():
    """include/vrag_amd_debug.h is exported by libvrag_amd_dbg.so only: a maintainer binding the product header gets no tuning
    probes as API, and the product library carries none of their symbols."""
    src = open(os.path.join(ROOT, "include", "vrag_amd_debug.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = sorted(set(re.findall(r"\b(vrag_[a-z0-9_]+)\s*\(", src)))
    assert names == sorted(_lib.DEBUG_SIGNATURES) and all(n.startswith("vrag_debug_") for n in names)
    prod = ctypes.CDLL(_lib.library_path())
    dbg = ctypes.CDLL(_lib.debug_library_path())
    for n in names:
        assert not hasattr(prod, n), f"{n} leaked into the product library"
        assert hasattr(dbg, n)
    assert not [n for n in _declared() if n.startswith("vrag_debug_")]
    _lib.load_debug()


def test_abi_version_and_error_string():
    lib = _lib.load()
    assert lib.vrag_abi_version() == _lib.ABI_VERSION == 6
    assert lib.vrag_device_count() >= 0
    assert isinstance(_lib.last_error(), str)


def test_no_cpu_fallback_without_gpu():
    """Product path fails loudly when there is no device (never routes to oracle/)."""
    lib = _lib.load()
    if lib.vrag_device_count() > 0:
        pytest.skip("GPU present")
    from verbatim_rag_amd.engine import EncoderEngine, ModernBertShape
    from verbatim_rag_amd.vector_stores import DenseShard, GpuVectorStore

    with pytest.raises(RuntimeError, match="no HIP device"):
        EncoderEngine(ModernBertShape(vocab_size=512, hidden_size=128, num_hidden_layers=1, num_attention_heads=2,
                                      intermediate_size=192, pad_token_id=0), {})
    with pytest.raises(RuntimeError, match="no HIP device"):
        DenseShard(64, 10)
    with pytest.raises(RuntimeError, match="no HIP device"):
        GpuVectorStore()
    # and the raw C entry points report VRAG_ERR_NO_DEVICE instead of computing on the host
    h = ctypes.c_void_p()
    assert lib.vrag_dense_index_create(64, 10, 0, 0, ctypes.byref(h)) == -4
    assert "no HIP device" in _lib.last_error()
    # the exchange entry points too: no communicator without a device, and no crash on bad arguments
    ident = (ctypes.c_uint8 * 128)()
    assert lib.vrag_comm_get_unique_id(ident) == -4
    assert lib.vrag_comm_create(bytes(128), 0, 1, 0, ctypes.byref(h)) == -4
    assert lib.vrag_comm_create(bytes(128), 3, 2, 0, ctypes.byref(h)) == -1
    assert lib.vrag_comm_allgather(None, None, None, 0, None) == -1


def test_product_package_never_imports_oracle():
    pkg = os.path.join(ROOT, "verbatim-rag_amd")
    for dirpath, _d, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                txt = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M), f


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    """No silent fallback when the shared object is absent: importing the binding raises."""
    monkeypatch.setenv("VRAG_AMD_LIB", str(tmp_path / "nope" / "libvrag_amd.so"))
    monkeypatch.setattr(_lib, "_LIB", None)
    with pytest.raises(ImportError, match="no CPU fallback"):
        _lib.load()
    monkeypatch.undo()
    assert _lib.load().vrag_abi_version() == _lib.ABI_VERSION


def test_graft_entry_build_is_consistent_with_the_binding():
    """__graft_entry__.build() (the driver's build check) must accept the library the binding accepts."""
    import importlib.util
    import os

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("_graft_entry", os.path.join(root, "__graft_entry__.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    mod.build()   # incremental: objects are up to date after the session's build


def test_one_hip_runtime_whatever_the_import_order():
    """libvrag_amd.so first, torch second must still leave ONE libamdhip64 mapped (torch's wheel bundles its own copy
    under an un-versioned NEEDED name; two runtimes in one process = torch sees no GPU, r2c GPU session)."""
    import subprocess
    import sys

    code = ("import sys; sys.path.insert(0, %r); import verbatim_rag_amd; from verbatim_rag_amd import _lib; _lib.load(); "
            "import torch; "
            "print(len({l.split()[-1] for l in open('/proc/self/maps') if 'libamdhip64' in l}))") % ROOT
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    assert out.stdout.strip().splitlines()[-1] == "1", out.stdout
