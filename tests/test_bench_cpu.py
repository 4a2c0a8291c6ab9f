"""Host-only checks of bench.py's argument handling (the measurement itself needs an MI355X)."""
import importlib
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def bench():
    sys.path.insert(0, ROOT)
    saved = sys.argv
    sys.argv = ["bench.py"]
    try:
        return importlib.import_module("bench")
    finally:
        sys.argv = saved


def test_multi_lane_detection_decides_the_hardware_queue_request(bench):
    assert not bench._multi_lane([])
    assert not bench._multi_lane(["--model", "hac", "--lanes", "1"])
    assert bench._multi_lane(["--lanes", "3"])
    assert bench._multi_lane(["--lanes=2"])
    assert bench._multi_lane(["--model", "fast"])                 # the fast model defaults to three lanes
    assert bench._multi_lane(["--quantize"]) and not bench._multi_lane(["--quantize", "--lanes", "1"])
    assert not bench._multi_lane(["--model", "fast", "--lanes", "1"])


def test_defaults_follow_the_contract(bench, monkeypatch):
    monkeypatch.setattr(sys, "argv", ["bench.py"])
    a = bench.parse()
    assert (a.gpus, a.model, a.decoder, a.batch, a.chunk, a.lanes) == (1, "hac", "beam", 512, 10000, 1)
    assert a.steps >= 1 and a.warmup >= 0 and a.set == []
    monkeypatch.setattr(sys, "argv", ["bench.py", "--model", "sup_lstm", "--set", "beam_fork=1", "--set", "conv_ws=0"])
    a = bench.parse()
    assert (a.batch, a.chunk, a.lanes, a.set) == (256, 20000, 1, ["beam_fork=1", "conv_ws=0"])


def test_batches_per_engine_call(bench, monkeypatch):
    """A step is always one batch of --batch chunks; hac in fp16 hands the engine two of them per call (paired rings) when the
    step count allows it, and never changes the number of timed steps."""
    def parsed(*argv):
        monkeypatch.setattr(sys, "argv", ["bench.py"] + list(argv))
        return bench.parse()
    a = parsed()
    assert (a.per_call, a.batch, a.call_batch, a.steps % a.per_call) == (4, 512, 2048, 0)
    assert parsed("--steps", "7").per_call == 1                      # exactly K steps: an odd K runs one batch per call
    assert parsed("--steps", "50").per_call == 2 and parsed("--steps", "20").per_call == 4
    assert parsed("--per-call", "1").call_batch == 512
    q = parsed("--quantize")                                         # 8-bit path: two lanes x four batches per call
    assert (q.lanes, q.per_call) == (2, 4) and parsed("--quantize", "--lanes", "1").per_call == 1
    f = parsed("--model", "fast")                                    # three lanes x eight batches per call
    assert (f.lanes, f.per_call, f.call_batch) == (3, 8, 4096) and parsed("--model", "fast", "--lanes", "1").per_call == 1
    assert parsed("--lanes", "2").per_call == 1
    # 1024-state models: two 256-chunk batches per call (their decode is one wave per chunk; two batches decode in the time of one)
    assert parsed("--model", "sup").per_call == 2 and parsed("--model", "sup_lstm").call_batch == 512
    assert parsed("--model", "sup", "--steps", "7").per_call == 1


def test_pmc_traffic_table_names_the_roofline_kernel(bench):
    table = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    entry = table["lstm_layer_wg_kernel"]
    assert entry["bytes_per_launch"] > 1e9 and os.path.exists(os.path.join(ROOT, entry["source"]))


def test_pmc_traffic_lookup_by_workload_and_kernel_name(bench):
    """`roofline.traffic` comes from the committed PMC passes and only for the workload they were taken on: the kernel name as the engine
    describes it (template arguments, a "gemm + " prefix for layers with a separate input projection), the chunks of ONE launch."""
    import types
    hac = types.SimpleNamespace(model="hac", call_batch=2048, chunk=10000)
    got, src = bench.pmc_traffic("lstm_layer_wgx2_kernel<12,3>", hac, 1024)
    assert got and 2.5e9 < got < 2.8e9 and src.startswith("profiles/")
    assert bench.pmc_traffic("lstm_layer_wgx2_kernel<12,3>", hac, 2048) == (None, None)          # another launch size: no measurement
    fast = types.SimpleNamespace(model="fast", call_batch=4096, chunk=10000)
    got, _ = bench.pmc_traffic("lstm_layer_cta_kernel<3,3>", fast)
    assert got and 2.5e9 < got < 2.8e9
    wide = types.SimpleNamespace(model="sup_lstm", call_batch=512, chunk=20000)
    got, _ = bench.pmc_traffic("gemm + lstm_layer_wide_kernel<32,true>", wide, 256)
    assert got and 8.5e9 < got < 9.5e9
    assert bench.pmc_traffic("attention_ring_kernel<12>", wide, 256) == (None, None)


def test_rocprof_summary_reports_median_and_roofline_leg_average(tmp_path):
    """tools/rocprof_summary.py on a synthetic rocpd database: AverageNs over every launch, MedianNs, and Last30AverageNs = the launches of
    bench.py's roofline leg (run one kernel at a time behind the timed regions) - the figure `roofline.avg_launch_ms` must agree with."""
    import csv
    import sqlite3
    import subprocess
    db = str(tmp_path / "t.db")
    con = sqlite3.connect(db)
    con.execute("create table kernels (name, start, end, vgpr_count, accum_vgpr_count, sgpr_count, lds_size, grid_x, workgroup_x)")
    for i in range(100):                 # 70 launches of the pipelined region (some stretched by overlap), then 30 clean ones
        dur = 3600 if i >= 70 else (7900 if i % 7 == 0 else 3500)
        con.execute("insert into kernels values ('k', ?, ?, 252, 0, 112, 99840, 65536, 256)", (i * 10000, i * 10000 + dur))
    con.commit()
    con.close()
    out = str(tmp_path / "t.csv")
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "rocprof_summary.py"), db, out], check=True, capture_output=True)
    row = list(csv.DictReader(open(out)))[0]
    assert int(row["Calls"]) == 100 and int(row["MedianNs"]) == 3500 and float(row["Last30AverageNs"]) == 3600.0
    assert float(row["AverageNs"]) > 3900 and int(row["MaxNs"]) == 7900


def test_experimental_lstm_flags_never_write_the_product_library(tmp_path):
    """`BH_EXTRA_LSTM_FLAGS` builds (timing experiments, wrong results on purpose) go to libbonito_hip_expt.so / build/obj_expt; a
    stray environment variable must not be able to replace bonito_amd/libbonito_hip.so (review, round 3)."""
    import subprocess
    import sys
    from conftest import ROOT
    code = ("import os, build; print(build.LIB); print(build.OBJ); print(build.EXTRA_FLAGS.get('lstm.hip'))")
    env = dict(os.environ, BH_EXTRA_LSTM_FLAGS="-DBH_EXPT_SHARE")
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    lib, obj, flags = r.stdout.strip().split("\n")
    assert lib.endswith("libbonito_hip_expt.so") and obj.endswith("obj_expt") and "BH_EXPT_SHARE" in flags
    env.pop("BH_EXTRA_LSTM_FLAGS")
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env, capture_output=True, text=True)
    lib, obj, flags = r.stdout.strip().split("\n")
    assert lib.endswith("bonito_amd/libbonito_hip.so") and obj.endswith("build/obj") and flags == "None"


@pytest.mark.parametrize("launcher", ["self", "torchrun"])
def test_eight_rank_launch_plumbing_dry_run(launcher):
    """The driver's 8-GPU run must not fail on plumbing: `bench.py --gpus 8` (its own spawner) and the driver's
    `python -m torch.distributed.run --nproc-per-node 8 ... bench.py --gpus 8` both rendezvous, barrier, MAX-reduce and print ONE JSON
    line with n_gpus 8 and the contract's keys. `--dry-run` puts a sleep in place of the hot path (no device here); everything else is
    the code the real run executes up to the device assertion."""
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    tail = [os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "6", "--warmup", "2", "--dry-run"]
    cmd = [sys.executable] + tail if launcher == "self" else [
        sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
        "--master-port", str(port)] + tail
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=dict(env, OMP_NUM_THREADS="1"))
    assert r.returncode == 0, r.stderr[-1500:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-500:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 8 and j["steps"] == 6 and j["warmup"] == 2 and j["scaling"] == "weak" and j["dry_run"] is True
    for key in ("metric", "value", "unit", "ms_per_step", "higher_is_better", "vs_baseline", "dtype", "data", "config"):
        assert key in j
    assert j["data"] == "dry-run" and j["value"] > 0 and j["ms_per_step"] >= 2.0
    # whole-job aggregate: eight ranks' samples over the slowest rank's time
    assert abs(j["value"] - 8 * 512 * 10000 * 6 / (j["ms_per_step"] * 6e-3)) / j["value"] < 1e-6
