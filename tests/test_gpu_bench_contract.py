"""bench.py prints one JSON line with the fields the driver reads."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_line_has_the_contract_fields():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "4", "--warmup", "2", "--in-flight", "2"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 4 and d["warmup"] == 2 and d["unit"] == "Mray/s" and d["value"] > 10
    assert abs(d["value"] - d["config"]["rays_per_frame"] / (d["ms_per_step"] * 1e-3) / 1e6) < 1e-6 * d["value"]
    rf = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in rf, k
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12 and 0 < rf["frac"] <= 1
    # the default workload is the configuration of Control.h as committed (512^2, 1000 photons, BOUNCES 0)
    assert d["config"]["workload"].startswith("cfg1:") and d["config"]["rays_per_frame"] == 512 * 512
    # the count that prices the frame is work the kernels execute: never more than the hardware count, and the
    # equivalent over the reference's gate-box pairs is kept apart from `frac`
    assert 0 < rf["algorithmic_flop_per_frame"] <= rf["flop_per_frame"] * (1 + 1e-9)
    assert "pairs_equivalent" in rf and "flop_source" in rf
    others = d["other_workloads"]
    assert set(others) == {"cfg2", "cfg3", "cfg4", "cfg5"}
    for name, o in others.items():
        assert o["frame_ok"] is True and o["value"] > 1
        if name == "cfg5":            # 4096^2, 1 000 000 photons, in bands: timed, its photon pass and LBVH against HBM
            assert o["config"]["segments"] > 20_000_000 and o["phases_gb_per_s"]["lbvh_build"] > 100
            continue
        assert 0 < o["roofline_frac"] <= 1, (name, o["roofline_frac"])
        assert 0 < o["roofline"]["frac"] <= 1 and o["roofline"]["kernel_ms"] > 0
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and cb["unit"] == "Mray/s" and "sample" in cb
    assert d["frame_ok"] is True and d["pipelined"]["frames_identical"] is True
    # the same frames with nothing kept from one frame to the next are reported beside the headline, and are slower
    unkept = d["schedule_rebuilt_every_frame"]
    assert unkept["ms_per_step"] > 0 and unkept["value"] > 10 and unkept["unit"] == "Mray/s"
