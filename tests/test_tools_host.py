"""Host logic of the measurement tools (no GPU): parsers and summaries on captured outputs under profiles/."""


def test_energy_probe_parses_rocm_smi_and_summarises():
    """tools/energy_probe.py host logic on the rocm-smi text captured in profiles/r01_power_probe.txt."""
    import importlib.util
    import os

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("energy_probe", os.path.join(root, "tools", "energy_probe.py"))
    ep = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ep)
    text = open(os.path.join(root, "profiles", "r01_power_probe.txt")).read()
    blocks = text.split("== sample")
    idle = ep.parse_power(blocks[0])
    samples = [ep.parse_power(b) for b in blocks[1:]]
    assert idle is not None and 200 < idle < 300 and len(samples) == 14 and all(s is not None for s in samples)
    line = {"value": 5676.0, "ms_per_step": 45.1, "n_gpus": 1, "config": {"chunks_per_gpu_per_step": 256}}
    out = ep.summarize([idle] * 3 + samples, idle, line, 1400.0)
    assert out["busy_samples"] == 14 and 1280 < out["avg_busy_w"] < 1400
    assert abs(out["joules_per_chunk"] - out["avg_busy_w"] * 45.1e-3 / 256) < 1e-12 and 0.2 < out["joules_per_chunk"] < 0.3
    assert ep.summarize([], idle, line)["samples"] == 0


def test_gelu_constants_in_the_kernel_header_are_the_fitted_ones():
    """csrc/common.h:gelu_fast evaluates erfc(z) = 2^P(z); its literal coefficients, re-evaluated here in float32 exactly as
    the kernel does (fmaf chain, exp2), must reproduce the erf-form GELU to < 1e-6 absolute -- the bound tools/fit_gelu.py
    prints for them and the reason the 16-bit GeGLU outputs do not move."""
    import os
    import re

    import numpy as np
    from scipy.special import erf

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = open(os.path.join(root, "verbatim-rag_amd", "csrc", "common.h")).read()
    body = src[src.index("float gelu_fast(float x)"):]
    body = body[:body.index("}")]
    lits = [float(v) for v in re.findall(r"(-?\d\.\d+e[+-]\d+)f", body)]
    assert len(lits) == 6, lits            # c6 .. c1 in Horner order
    zmax = float(re.search(r"fminf\(a \* [0-9.]+f, ([0-9.]+)f\)", body).group(1))
    f = np.float32
    x = np.linspace(-9, 9, 900001).astype(f)
    a = np.abs(x)
    z = np.minimum(a * f(0.70710678118654752440), f(zmax))
    t = np.full_like(z, f(lits[0]))
    for c in lits[1:]:
        t = (t * z + f(c)).astype(f)
    got = (f(-0.5) * a) * np.exp2((t * z).astype(f)).astype(f) + np.maximum(x, f(0))
    ref = 0.5 * x.astype(np.float64) * (1 + erf(x.astype(np.float64) / np.sqrt(2)))
    assert np.abs(got - ref).max() < 1e-6


def test_committed_bench_line_follows_the_output_contract():
    """profiles/r02_bench_line.json is what `python bench.py` printed on the GPU box: the keys the driver and the judge read must
    be there with the right kinds (a guard against silently dropping one from bench.py's line)."""
    import json
    import os

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    line = json.loads(open(os.path.join(root, "profiles", "r02_bench_line.json")).read().strip().splitlines()[-1])
    for key, kind in (("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int),
                      ("ms_per_step", float), ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str),
                      ("config", dict), ("roofline", dict), ("cpu_baseline", dict)):
        assert isinstance(line[key], kind), key
    assert "vs_baseline" in line and line["vs_baseline"] is None          # BASELINE.md holds no published number
    assert line["scaling"] == "weak" and line["higher_is_better"] is True and "workload" in line["config"]
    assert abs(line["value"] - 256 * line["n_gpus"] / (line["ms_per_step"] * 1e-3)) < 1e-6 * line["value"]
    roof = line["roofline"]
    assert roof["bound"] in ("hbm", "mfma") and roof["unit"] in ("GB/s", "TFLOP/s")
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-9 and 0 < roof["frac"] < 1
    assert roof["traffic"] is None or roof["traffic"] > 0
    cpu = line["cpu_baseline"]
    assert cpu["kind"] in ("reference", "port") and cpu["cores"] >= 1 and cpu["value"] > 0 and cpu["sample"]
