"""bench.py on a GPU box (-m gpu): the JSON contract of a short real run, and the N > 1 launch paths with real device work (the ranks share
the one GPU of the test box: more ranks than devices -> the barrier and the MAX-reduce go over gloo; on a multi-GPU node the same code takes
one rank per device over RCCL)."""
import json
import os
import socket
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu
KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config")


def _line(cmd, timeout=600):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-500:]          # stdout carries exactly one JSON line
    return json.loads(lines[0]), r.stderr


def test_single_rank_line_carries_roofline_and_the_contract_keys():
    j, _ = _line([sys.executable, "bench.py", "--model", "fast", "--steps", "24", "--warmup", "8", "--no-cpu-baseline", "--no-side-legs"])
    for k in KEYS:
        assert k in j
    assert j["n_gpus"] == 1 and j["steps"] == 24 and j["scaling"] == "weak" and j["vs_baseline"] is None and j["data"] == "synthetic"
    assert j["value"] > 1e8 and abs(j["value"] - 512 * 10000 * 24 / (j["ms_per_step"] * 24e-3)) / j["value"] < 1e-6
    rf = j["roofline"]
    assert rf["bound"] == "mfma" and 0 < rf["frac"] < 1 and rf["unit"] == "TFLOP/s" and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    assert "workload" in j["config"] and "model" not in j["config"]


@pytest.mark.parametrize("launcher", ["self", "torchrun"])
def test_two_ranks_on_this_box_aggregate_over_the_slowest_rank(launcher):
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    tail = ["bench.py", "--gpus", "2", "--model", "fast", "--steps", "24", "--warmup", "8"]
    cmd = [sys.executable] + tail if launcher == "self" else [
        sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
        "--master-port", str(port)] + tail
    j, err = _line(cmd)
    assert j["n_gpus"] == 2 and j["steps"] == 24 and j["scaling"] == "weak"
    # whole-job aggregate: both ranks' samples over the slowest rank's time
    assert abs(j["value"] - 2 * 512 * 10000 * 24 / (j["ms_per_step"] * 24e-3)) / j["value"] < 1e-6
    assert abs(j["per_gpu"] * 2 - j["value"]) / j["value"] < 1e-9
    assert "rank 0/2" in err and "rank 1/2" in err                      # every rank says which device it drives
    assert j["cpu_baseline"] is None and j["other_configs"] is None      # rank 0, N = 1 only
