"""The exchange code of a job of several GPUs, run by one rank on one GPU: RCCL process group, segment broadcast,
the strip gather pipelined on the collective's stream (tiles.StripPipeline, two frames in flight), bench.py's loop."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env(port):
    env = dict(os.environ)
    env.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    return env


def test_pipelined_strip_gather_over_rccl_with_one_rank():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "exchange_rehearsal.py")], stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=300, env=_env(29519))
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-3000:])
    assert "exchange rehearsal ok" in r.stdout


def test_bench_runs_its_exchange_path_with_one_rank():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--rehearse-exchange", "--steps", "30", "--warmup", "5",
                        "--workload", "cfg2", "--only", "--no-cpu-baseline", "--in-flight", "0"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300, env=_env(29521))
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.strip()][-1])
    assert d["frame_ok"] is True and d["n_gpus"] == 1 and "RCCL" in d["config"]["parallelism"]
    assert d["value"] > 10 and 0 < d["roofline"]["frac"] <= 1
