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
