"""The fused Wqkv + RoPE + attention kernel (csrc/qkv_attn.hip) against the oracle at every sequence geometry the encoder
suite holds -- ragged lengths 1 .. 512 in one batch, lengths off the 64-key tile grid, banded and global layers, both
operand types.  By default the kernel only serves throughput-sized micro-batches (> 8192 token rows; test_full_shapes_gpu
and the bench batch exercise it there); VRAG_FUSED_QKV_ATTN=2 makes it serve the small batches of the encoder suite too.
The knob is read when an encoder handle is created, so the suite runs in a child process."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _run(files, fused):
    env = {**os.environ, "VRAG_FUSED_QKV_ATTN": str(fused)}
    out = subprocess.run([sys.executable, "-m", "pytest", "-m", "gpu", "-x", "-q", "-p", "no:cacheprovider", *files], cwd=ROOT, env=env,
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
    return out.stdout


def test_encoder_suite_through_the_fused_kernel():
    _run(["tests/test_encoder_gpu.py", "tests/test_heads_gpu.py"], 2)


def test_fuzzed_batches_through_the_fused_kernel():
    """Round 4: wave-slot packing -- a workgroup holds a group of consecutive sequences, each on ceil(S / 64) of its eight waves.
    The randomised batch compositions of test_fuzz_gpu.py (1 .. 512 tokens, up to 23 sequences, random micro-batch cuts) give
    groups of every shape: many one-wave sequences in one workgroup, a sequence ending a group, a 512-token one alone."""
    _run(["tests/test_fuzz_gpu.py"], 2)


def test_headline_batch_through_the_two_kernel_path():
    """The other direction: the 256 x 512 batch with the fused kernel switched off (what sequences above 512 tokens get)."""
    _run(["tests/test_full_shapes_gpu.py::test_configs1_batch_256x512_sample_vs_oracle"], 0)
