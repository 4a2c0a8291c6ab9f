#!/usr/bin/env python3
"""
Headline benchmark: signal samples/sec/GPU at chunk=10000, batch=512 (BASELINE.json `metric`).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--model hac|fast|sup|sup_lstm] [--decoder viterbi|beam] [--quantize]

One "step" = one pass of the hot path over one synthetic batch: fp16 signal [512, 10000] -> HIP encoder (conv x3,
LSTM x5, LinearCRFEncoder) -> HIP CRF decode -> int8 moves / sequence / qstring on the host. THREE distinct batches rotate
through the steps. Weights are seeded random-init tensors of the named architecture (no checkpoints offline).

Two kinds of timed region, each of EXACTLY K steps bracketed by barrier + synchronize, MAX over ranks:
  * `value` / `ms_per_step`: input batches already resident in HBM when the region starts (the bench contract: a PCIe-inclusive
    rate is never `value`);
  * `with_h2d` / `value_with_h2d`: the same steps with each batch copied pinned-host -> device on a copy stream inside the step
    ("H2D of fp16 signal included"). The copy of call k+1 overlaps the kernels of call k; the FIRST call's batch is
    staged before the clock starts, as the product's reader thread has it (crf/basecall.py). This leg runs FIRST.
With the driver's flags (--steps 20) a region is five engine calls = 0.24 s, and one hiccup moved its mean by 46 % (round 5: 17.30 ms
mean against a 10.88 median). Each kind of region is therefore run `--repeats` times (default: enough for >= 16 engine calls per kind,
at most 5) and the MEDIAN region is reported; every region's figure is in `regions_ms_per_step`, every call's time goes to stderr.
`ms_per_step` is elapsed / K of that region; `ms_per_step_median` is the median distance between consecutive steps' decode-done events.
The warm-up is W steps AND at least --warmup-seconds of engine calls (clocks, instruction caches and the allocator settle in
about a second; a 20-step run used to be timed cold: mean 20.3 ms against a median of 16.3).

More legs in the same JSON line (rank 0, N = 1 only; each bounded to a few seconds):
  per_call_1    -- the reference's call shape: ONE batch of 512 chunks per engine call (what `--per-call 1` or a short input runs; the
                   product path groups batches into the default leg's calls of 2048 chunks), with the roofline of the kernel that serves it
  other_configs -- the other BASELINE.json configurations (fast 512 x 10000, sup transformer 256 x 12000, sup LSTM-1024
                   256 x 20000, hac --quantize), each a child process of this script with a short timed region

N > 1: `python bench.py --gpus N` spawns N ranks itself (one process per GPU, RCCL for the barrier and the MAX-reduce only);
under `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N` the ranks come from the environment. Read
chunks shard embarrassingly, every rank runs the same per-GPU workload, no data-path collective -> "scaling": "weak".

The JSON line also carries
  roofline     -- dominant kernel's algorithmic FLOP/s from HIP-event timings on the engine stream (kernels running alone)
  cpu_baseline -- the CPU oracle (PyTorch-CPU fp32 restatement of bonito/nn.py + C Viterbi) on a bounded sample (rank 0, N=1)
"""
import argparse
import json
import os
import re
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
# HIP maps streams onto 4 hardware queues by default; with several batch lanes (2 streams each, plus the copy / helper
# streams) streams would share queues and serialise (fast model, 3 lanes: 9.1 ms/step with 4 queues, 4.5 with 8). With one
# lane the default is marginally better (hac: 25.5 vs 25.8 ms/step), so only multi-lane runs ask for more. Has to be in
# the environment before the HIP runtime is loaded, hence the look at argv ahead of `import torch`.
def _multi_lane(argv):
    lanes = None
    for i, tok in enumerate(argv):
        if tok == "--lanes" and i + 1 < len(argv):
            lanes = argv[i + 1]
        elif tok.startswith("--lanes="):
            lanes = tok.split("=", 1)[1]
    if lanes is not None:
        return lanes.isdigit() and int(lanes) > 1
    # the fast model defaults to three lanes, the 8-bit path to two
    return any(tok in ("fast", "--model=fast", "--quantize") for tok in argv)


if _multi_lane(sys.argv[1:]):
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, ROOT)

MFMA_F16_PEAK_TFLOPS = 2500.0     # MI355X dense fp16/bf16 (MI355X_MICROARCH.md)
MFMA_I8_PEAK_TOPS = 5000.0        # dense int8 (2x the fp16 rate: v_mfma_i32_16x16x64_i8)
HBM_PEAK_GBS = 8000.0
N_BATCHES = 3                     # distinct input batches rotating through the steps


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--model", default="hac", choices=["hac", "fast", "sup", "sup_lstm"])
    ap.add_argument("--batch", type=int, default=0, help="default: 512 (hac/fast), 256 (sup)")
    ap.add_argument("--chunk", type=int, default=0, help="default: 10000 (hac/fast), 12000 (sup)")
    ap.add_argument("--decoder", default="beam", choices=["viterbi", "beam"])
    ap.add_argument("--quantize", action="store_true",
                    help="int8 recurrent path (the reference's --quantize, cli/basecaller.py:186-189): Q8-1 kernels; NOT the default")
    ap.add_argument("--lanes", type=int, default=0,
                    help="independent batches in flight (each lane: own engine replica, encoder stream, decoder stream); "
                         "default 1; 3 for the narrow `fast` model whose kernels leave most CUs idle (1 lane 8.9 ms/step, "
                         "3 lanes 4.5 with GPU_MAX_HW_QUEUES=8); hac / sup kernels fill the chip and gain nothing")
    ap.add_argument("--per-call", type=int, default=0,
                    help="batches per engine call (a step stays ONE batch of --batch chunks). Default 4 for hac in fp16: with more "
                         "than 32 rings in a call the recurrent kernel carries two rings per workgroup on one copy of the weights "
                         "(lstm_layer_wgx2_kernel) and the hand-off of one hides behind the step of the other: 19.8 -> 18.3 ms per "
                         "batch with two batches per call on the same box, 17.2 with four (two paired launches per layer; the "
                         "decode kernels of 2048 chunks pack the CUs better); 1 = one batch per call (lstm_layer_wgx_kernel). Needs "
                         "--steps divisible by it, otherwise the largest divisor among 4, 2, 1 is used.")
    ap.add_argument("--warmup-seconds", type=float, default=1.5, help="the warm-up also lasts at least this long")
    ap.add_argument("--repeats", type=int, default=0,
                    help="timed regions per leg, each of exactly --steps steps; the median region is reported (default: as many as "
                         "give >= 16 engine calls per leg, between 1 and 5)")
    ap.add_argument("--parity-chunks", type=int, default=-1,
                    help="chunks of the CPU-oracle sample (cpu_baseline + parity); default 128 for hac / fast, 2 for the sup models; 0 = none")
    ap.add_argument("--dry-run", action="store_true",
                    help="plumbing check without a device (tests/test_bench_cpu.py): ranks, rendezvous (gloo), barrier, MAX-reduce and the JSON "
                         "contract of an N-rank launch, with a sleep in place of the hot path; the line says \"data\": \"dry-run\" and is not a measurement")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-side-legs", action="store_true", help="skip per_call_1 and other_configs (child runs use this)")
    ap.add_argument("--no-h2d-leg", action="store_true", help="skip the second timed region (H2D inside the step)")
    ap.add_argument("--set", action="append", default=[], metavar="NAME=VALUE",
                    help="library tuning option (bh_set_option), e.g. beam_fork=1; for A/B runs, not part of the contract")
    a = ap.parse_args(argv)
    a.batch = a.batch or (256 if a.model in ("sup", "sup_lstm") else 512)
    a.chunk = a.chunk or (12000 if a.model == "sup" else 20000 if a.model == "sup_lstm" else 10000)
    lanes_given = a.lanes > 0
    a.lanes = a.lanes or (3 if a.model == "fast" else 2 if a.quantize and a.model == "hac" else 1)
    if not a.per_call:
        # measured on MI355X (profiles/r02_bench_lines.jsonl): hac fp16 - four batches per call, one lane (paired recurrent kernel);
        # hac --quantize - two lanes (8-bit kernels compiled for two workgroups per CU) x four batches per call (two until round 5);
        # fast - three lanes x eight batches per call (the ring-in-a-workgroup kernel of one 512-chunk batch fills an eighth of the
        # chip): 3.57 -> 2.51 ms per batch with four, 2.28 with eight
        if a.model == "hac" and not a.quantize and a.lanes == 1:
            a.per_call = 4
        elif a.model == "hac" and a.quantize and a.lanes == 2 and not lanes_given:
            a.per_call = 4          # (round 5: 2048-chunk calls give the decode stage eight chunks per CU: 12.28 -> 11.63 ms; 2 before)
        elif a.model == "fast" and not a.quantize and a.lanes == 3 and not lanes_given:
            a.per_call = 8          # (4096 chunks = 256 rings = one ring-in-a-workgroup per CU; round 5: 2.37 -> 2.28 ms, four before)
        elif a.model in ("sup", "sup_lstm") and a.lanes == 1:
            # 1024 states: the decode is one wave per chunk (a latency chain), two 256-chunk batches decode in the time of one
            # (round 5: sup 64.1 -> 62.8 ms per batch, sup_lstm 101.3 -> 97.5); what crf/basecall.py batches_per_call picks
            a.per_call = 2
        else:
            a.per_call = 1
    while a.steps % a.per_call:                       # exactly --steps batches are timed: fall back to a divisor
        a.per_call //= 2
    a.call_batch = a.batch * a.per_call
    if a.repeats <= 0:
        a.repeats = max(1, min(5, -(-16 // max(1, a.steps // a.per_call))))
    if a.parity_chunks < 0:
        a.parity_chunks = parity_chunks(a.model)
    return a


def build_model(name, batch, chunk):
    from bonito_amd import synthetic
    if name == "sup":
        return synthetic.make_transformer_model(head_gain=4.0, batchsize=batch, chunksize=chunk)
    return synthetic.make_model(name, batchsize=batch, chunksize=chunk)


def flops(name, chunk):
    from bonito_amd import synthetic
    return synthetic.transformer_flops_per_chunk(chunksize=chunk) if name == "sup" else synthetic.flops_per_chunk(name, chunk)


def pmc_traffic(kernel, a, launch_chunks=None):
    """(HBM bytes per launch, source file) of the dominant kernel from the committed PMC passes (profiles/pmc_traffic.json:
    rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, gfx950 FETCH_SIZE x2 correction); (None, None) when no
    measurement exists for this kernel / workload. PMC counters cannot be read from inside this process. Keys of the table:
    "<kernel base name>|<workload>" (round 6: one kernel serves several workloads) or, older entries, the bare base name with the
    workload inside the entry."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as fh:
            table = json.load(fh)
    except (OSError, ValueError):
        return None, None
    workload = "%s%s %dx%d" % (a.model, " quantize" if getattr(a, "quantize", False) else "", launch_chunks or a.call_batch, a.chunk)
    base = (kernel or "").split("<")[0].split(" ")[0]
    named = re.search(r"(lstm_layer_\w+_kernel)", kernel or "")          # "gemm + lstm_layer_wide_kernel<32,true>" -> the recurrent kernel
    for key in ([named.group(1)] if named else []) + [base]:
        ent = table.get("%s|%s" % (key, workload))
        if ent:
            return ent["bytes_per_launch"], ent.get("source")
    ent = table.get(kernel or "") or table.get(base) or (table.get(named.group(1)) if named else None)
    if not ent or ent.get("workload") != workload:
        return None, None
    return ent["bytes_per_launch"], ent.get("source")


def log(msg):
    sys.stderr.write("[bench %7.1fs] %s\n" % (time.perf_counter() - T_START, msg))
    sys.stderr.flush()


T_START = time.perf_counter()


def cpu_baseline_worker(name, chunk, decoder="beam", n=None, keep=None, extras=True):
    """CPU oracle timed on this host: oracle/nn_ref.py forward (fp32) + oracle/crf_oracle.c decode, ONE pass over `n` chunks (the bounded
    sample: 64 x 10000 samples of hac take 10-20 s of the GPU box's 16-core quota). `value` is the same pipeline as the GPU leg
    (forward + the decoder the GPU leg ran); the other decoder's rate is reported beside it. `keep`: path of an .npz that receives the
    oracle's outputs for its chunks - the parent compares the HIP path with them (`parity` in the JSON line). With `extras` the file also
    holds (untimed) the outputs of the fp16-STORAGE oracle on the same chunks (prefix `h_`) and the BS-2-against-BS-1 quality guard."""
    import numpy as np
    import torch
    from oracle import nn_ref, parity
    model = build_model(name, 8, chunk)
    nn_ref.round_params_to_half_(model)
    from bonito_amd.util import effective_cpu_count
    ncores = max(1, min(effective_cpu_count(), 32))      # affinity capped by the cgroup quota; small matmuls stop scaling early
    torch.set_num_threads(ncores)
    n = n or parity_chunks(name)
    x = parity_input(n, chunk)
    tm = {}
    out = parity.oracle_outputs(model, x.float(), timers=tm, threads=ncores)
    t_fwd, t_vit, t_beam = tm["forward"], tm["viterbi"], tm["beam"]
    work = n * chunk
    rate = {"viterbi": work / (t_fwd + t_vit), "beam": work / (t_fwd + t_beam)}
    if keep:
        if extras:
            t0 = time.perf_counter()
            h = parity.oracle_outputs(model, x.float(), threads=ncores, fp16=True)
            out.update({"h_" + k: v for k, v in h.items()})
            guard = parity.bs2_vs_bs1(out["scores"], int(out["state_len"]), out["beam_seq"], threads=ncores)
            out["bs2_vs_bs1_json"] = np.frombuffer(json.dumps(guard).encode(), np.uint8)
            sys.stderr.write("cpu leg: fp16-storage oracle + BS-1 guard %.1f s (untimed)\n" % (time.perf_counter() - t0))
        np.savez(keep, **out)
    return {"value": rate[decoder], "unit": "samples/s", "cores": ncores, "kind": "port",
            "decoder": decoder, "value_viterbi": rate["viterbi"], "value_beam": rate["beam"], "forward_only": work / t_fwd,
            "seconds": {"forward": round(t_fwd, 2), "viterbi": round(t_vit, 2), "beam": round(t_beam, 2)},
            "sample": "one pass over %d chunks x %d samples (%.1f s of CPU work): oracle/nn_ref.py fp32 forward (torch, %d threads; kind "
                      "'port': a restatement of the reference's bonito/nn.py, equal to it to 1.5e-7 on the committed fixtures "
                      "tests/golden/nn_*.npz - the reference's own code is never executed by the bench) + oracle/crf_oracle.c %s decode "
                      "(%d chunk slices in parallel); both decoders were timed on the same scores"
                      % (n, chunk, t_fwd + t_vit + t_beam, ncores, decoder, min(ncores, n))}


def parity_chunks(name):
    """Chunks of the CPU-oracle sample (cpu_baseline + parity) of a headline run: 128 x 10000 samples of hac are ~10 s of the GPU box's 16-core
    quota for the timed fp32 pass (+ as much again, untimed, for the fp16-storage oracle and the decoder guard)."""
    return 2 if name in ("sup", "sup_lstm") else 128


def parity_input(n, chunk):
    import torch
    return torch.randn(n, 1, chunk, generator=torch.Generator().manual_seed(25)).half()


def cpu_baseline(name, chunk, decoder="beam", hard_timeout=240.0, keep=None, n=None):
    """Run the CPU leg in a child process with a hard wall-clock bound so it can never stall the bench."""
    import subprocess
    code = ("import json,sys; sys.path.insert(0, %r); import bench; "
            "print('CPU_BASELINE ' + json.dumps(bench.cpu_baseline_worker(%r, %d, %r, n=%r, keep=%r)))" % (ROOT, name, chunk, decoder, n, keep))
    try:
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=hard_timeout,
                           env=dict(os.environ, HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES=""))
        for line in r.stdout.splitlines():
            if line.startswith("CPU_BASELINE "):
                return json.loads(line[len("CPU_BASELINE "):])
        log("cpu baseline produced no result: " + r.stderr[-400:])
    except subprocess.TimeoutExpired:
        log("cpu baseline exceeded %.0fs and was abandoned" % hard_timeout)
    return None


def parity_leg(a, model, signals, dec, keep):
    """`parity`: the HIP path (this process: the timed engine + decoders, the oracle's chunks placed at the head of a full engine call)
    against the oracle outputs the cpu_baseline child left in `keep` - against the fp32 CPU path (the top-level figures) AND against the
    fp16-storage oracle (`vs_fp16_storage_oracle`), plus the two oracles against each other and the BS-2-vs-BS-1 guard. See
    oracle/parity.py for the definitions."""
    import numpy as np
    import torch
    from oracle import parity
    ora = dict(np.load(keep))
    n = int(ora["scores"].shape[0])
    x = signals[0].clone()
    x[:n] = parity_input(n, a.chunk).to(x.device)
    hip = parity.hip_outputs(model, x, n, decoder=dec if a.decoder == "beam" else None)
    res = parity.compare(hip, ora)
    res["ctx_equal"] = hip.get("ctx_equal")
    if "h_scores" in ora:
        ora16 = {k[2:]: v for k, v in ora.items() if k.startswith("h_")}
        res["vs_fp16_storage_oracle"] = parity.compare(hip, ora16)
        res["fp16_storage_oracle_vs_fp32_oracle"] = parity.compare(ora16, ora)
    if "bs2_vs_bs1_json" in ora:
        res["bs2_vs_bs1"] = json.loads(bytes(ora["bs2_vs_bs1_json"]).decode())
    res["note"] = ("top level: HIP encoder + HIP decoders vs the fp32 CPU path (oracle/nn_ref.py on the fp16-rounded weights, scores rounded to "
                   "fp16, oracle/crf_oracle.c decoders) on the same %d chunks x %d samples, run as the first chunks of a full %d-chunk "
                   "engine call; identity = matches / alignment columns (oracle/parity.py). vs_fp16_storage_oracle: the same HIP outputs "
                   "against nn_ref's fp16-STORAGE mode (values rounded where the engine stores fp16; fp32 accumulation in torch's order, libm) "
                   "- what is left there is summation order + the hardware exponential; fp16_storage_oracle_vs_fp32_oracle: the two CPU "
                   "paths against each other = what fp16 storage alone costs. bs2_vs_bs1: exact fp64 sequence log-probability of the "
                   "product decoder's answer minus that of the rounds-1-4 decoder on the oracle's scores. Seeded random weights: a "
                   "statement about arithmetic, not about read accuracy on trained checkpoints" % (n, a.chunk, x.shape[0]))
    return res


def e2e_worker(name="hac", reads=20000, mean_len=100000, batchsize=512):
    """The PRODUCT path end to end on synthetic reads, with the reference CLI's own definition of samples/s (bonito/cli/basecaller.py:
    156-164: t0 right before writer.start(), stop after writer.join(), samples from writer.log): reads -> chunk -> batch -> H2D ->
    HIP encoder -> HIP beam decode -> D2H -> stitch -> FASTQ records with move tables -> the real Writer into /dev/null. Model load /
    engine build is outside the clock, as in the reference."""
    import importlib
    import numpy as np
    from bonito_amd import io as bio
    from bonito_amd import synthetic, util
    basecall_records = importlib.import_module("bonito_amd.crf.basecall").basecall_records
    util.limit_host_threads(8)
    model = synthetic.make_model(name, batchsize=batchsize, chunksize=10000)
    model.use_koi(batchsize=batchsize, chunksize=9996, quantize=False)
    model = model.half().cuda()

    class Read:
        run_id, filename, channel, mux, start, duration, template_start, template_duration, trimmed_samples = "run", "f", 0, 0, 0.0, 0.0, 0.0, 0.0, 0

        def __init__(self, i, sig):
            self.read_id, self.signal, self.num_samples = "read_%d" % i, sig, len(sig)

    rng = np.random.default_rng(1)
    lens = np.clip(rng.normal(mean_len, mean_len / 3, reads), 5000, None).astype(int)
    # a pool of distinct signals reused cyclically: every read is a window of one of them, with its own id and length
    pool = [np.random.default_rng(100 + k).standard_normal(int(lens.max()) + 1).astype(np.float32) for k in range(16)]

    def gen(ls, base=0):
        for i, n in enumerate(ls):
            yield Read(base + i, pool[i % len(pool)][:int(n)])

    def once(ls):
        records = basecall_records(model, gen(ls), "fastq", chunksize=9996, overlap=498, batchsize=batchsize)
        with open(os.devnull, "w") as sink:
            w = bio.Writer("fastq", records, fd=sink, preformatted=True)
            t0 = time.perf_counter()
            w.start()
            w.join()
            dt = time.perf_counter() - t0
        if w.error is not None:
            raise w.error
        return sum(n for _, n in w.log), len(w.log), dt

    once(lens[:600])                      # clocks, allocator, pinned pools
    done, nreads, dt = once(lens)
    assert done == int(lens.sum()), (done, int(lens.sum()))
    aff = len(os.sched_getaffinity(0))
    # what the DEVICE computed for those reads: every read is cut into chunks of 9996 samples that overlap by 498, plus one stub chunk
    # where the length does not fit (util.chunk) - 9-10 % more chunk-samples than read samples at these lengths. `value` counts read
    # samples (the reference CLI's definition); the figure to hold against the bench's `value` (chunk-samples/s) is this one
    step = 9996 - 498
    n_chunks = int(sum((int(n) - 498 + step - 1) // step if n >= 9996 else 1 for n in lens))
    return {"value": done / dt, "unit": "samples/s", "reads": nreads, "samples": done, "seconds": dt, "host_cpus": aff,
            "chunks": n_chunks, "chunk_samples_per_s": n_chunks * 9996 / dt,
            "chunk_samples_note": "chunks x 9996 / seconds: the device-side rate of this run (overlap 498 of 9996 and one stub chunk per read "
                                  "make it 9-10 % more than the read samples); compare THIS with `value` of the bench line",
            "host_cpus_effective": util.effective_cpu_count(),
            "definition": "bonito/cli/basecaller.py:156-164 (clock around writer.start() .. writer.join(), samples from writer.log)",
            "workload": "%s, %d synthetic reads (normal lengths, mean %d), chunksize 9996 overlap 498 batchsize %d, FASTQ + move tables "
                        "to /dev/null through bonito_amd.io.Writer; one process, one GPU" % (name, nreads, mean_len, batchsize)}


def e2e_leg(name, hard_timeout=150.0):
    import subprocess
    code = ("import json,sys; sys.path.insert(0, %r); import bench; print('E2E ' + json.dumps(bench.e2e_worker(%r)))" % (ROOT, name))
    try:
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=hard_timeout,
                           env=dict(os.environ, GPU_MAX_HW_QUEUES="8"))
        for line in r.stdout.splitlines():
            if line.startswith("E2E "):
                return json.loads(line[4:])
        return {"error": (r.stderr or "no output")[-300:]}
    except subprocess.TimeoutExpired:
        return {"error": "exceeded %.0f s" % hard_timeout}


OTHER_PARITY_CHUNKS = {"fast": 32, "sup": 4, "sup_lstm": 4, "sup_20000": 4, "hac_quantize": 32}     # chunks of each child's CPU-oracle sample
OTHER_CONFIGS = {            # BASELINE.json configs 2, 4, 5 (+ the 8-bit path of config 3); the headline itself is config 3
    "fast": ["--model", "fast", "--steps", "384", "--warmup", "48"],        # (three lanes x eight batches per call = 16 calls per lane; 96 steps were
                                                                            #  half of them ramp-up - 2.53 ms against 2.37 over 384 steps)
    "sup": ["--model", "sup", "--steps", "12", "--warmup", "3"],
    "sup_lstm": ["--model", "sup_lstm", "--steps", "8", "--warmup", "2"],
    # config 5 on the OTHER candidate graph ("run both the @v5.0.toml graph and the @v4.3.toml graph at this shape")
    "sup_20000": ["--model", "sup", "--chunk", "20000", "--steps", "8", "--warmup", "2"],
    "hac_quantize": ["--model", "hac", "--quantize", "--steps", "48", "--warmup", "8"],
}


def other_configs(a, hard_timeout=240.0):
    """The other BASELINE configurations, each as a child process of this script (its own HIP context and queue settings) with a short
    timed region; the parent's kernels are idle meanwhile. Returns {name: {value, ms_per_step, ms_per_step_median, config, roofline}}."""
    import subprocess
    out = {}
    for name, flags in OTHER_CONFIGS.items():
        if name == ("hac_quantize" if a.quantize else a.model) and a.model != "hac":
            continue
        if a.model == "hac" and a.quantize and name == "hac_quantize":
            continue
        # every child carries its own small CPU-oracle sample (`parity`: config 5 = 256 x 20000 on BOTH candidate graphs included) and
        # runs ONE timed region: the steps of these legs are long enough (sup: 12 x 60 ms) or many enough (fast: 384)
        cmd = [sys.executable, os.path.abspath(__file__)] + flags + ["--no-h2d-leg", "--no-side-legs", "--repeats", "1",
                                                                     "--parity-chunks", str(OTHER_PARITY_CHUNKS.get(name, 2)),
                                                                     "--warmup-seconds", "1.0", "--decoder", a.decoder]
        log("other config: " + " ".join(flags))
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=hard_timeout)
            line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
            if r.returncode or not line:
                out[name] = {"error": (r.stderr or "no output")[-300:]}
                continue
            j = json.loads(line[-1])
            out[name] = {k: j.get(k) for k in ("value", "ms_per_step", "ms_per_step_median", "steps", "dtype", "roofline", "kernel_ms_per_step",
                                               "parity", "cpu_baseline")}
            out[name]["workload"] = j["config"]["workload"]
        except subprocess.TimeoutExpired:
            out[name] = {"error": "exceeded %.0f s" % hard_timeout}
    return out


def config1_worker(config="dna_r9.4.1@v2.toml", n_reads=16, read_len=4000):
    """BASELINE config 1: the legacy bonito.ctc path (QuartzNet `dna_r9.4.1` + CTC greedy / prefix-beam-5 decode) on 16 synthetic chunks
    of 4000 samples - the reference's own CPU-runnable case (bonito/ctc/basecall.py:14-61). The graph comes from the reference's config
    file (kept as test data under tests/golden/configs), the weights from its constructors under the CLI's seed. The HIP path runs the
    product pipeline `bonito_amd.ctc.basecall` (chunk -> batchify -> engine -> unbatchify -> stitch -> HIP CTC decoders); the CPU path is
    oracle/nn_ref.ctc_forward (fp32) + oracle/ctc_ref decoders on the same reads. Identity = matches / alignment columns."""
    import numpy as np
    import torch
    from bonito_amd import synthetic, util
    from bonito_amd.ctc import Model, basecall
    from oracle import ctc_ref, nn_ref, parity
    path = os.path.join(ROOT, "tests", "golden", "configs", config)
    cfg = util.load_toml(path)
    torch.manual_seed(25)
    cpu_model = Model(cfg)
    synthetic.randomise_batchnorm_(cpu_model, 26)
    cpu_model.eval()
    # a freshly initialised 50-layer QuartzNet forgets its input (every read decodes to the same three bases): the convolutions get the
    # gain that keeps the signal's variance through the stack and the head a gain that makes it emit (as synthetic.make_model does for
    # the CRF heads) - 800 greedy / 580 beam bases per read that depend on the read
    with torch.no_grad():
        for mod in cpu_model.encoder.modules():
            if isinstance(mod, torch.nn.Conv1d):
                mod.weight.mul_(2.0)
        cpu_model.decoder.layers[0].weight.mul_(4.0)
    nn_ref.round_params_to_half_(cpu_model)
    gpu_model = Model(cfg)
    gpu_model.load_state_dict(cpu_model.state_dict())
    gpu_model.eval()
    gpu_model.use_koi(batchsize=n_reads, chunksize=read_len, quantize=False)
    gpu_model = gpu_model.half().to("cuda")

    class Read:
        def __init__(self, i, sig):
            self.read_id, self.signal = "read_%d" % i, sig

    rng = np.random.default_rng(25)
    reads = [Read(i, rng.standard_normal(read_len).astype(np.float32)) for i in range(n_reads)]
    out = {"workload": "%s (QuartzNet CTC, seeded weights), %d reads = %d chunks x %d samples, batchsize %d, greedy + prefix beam 5"
                       % (config, n_reads, n_reads, read_len, n_reads)}
    hip = {}
    for beamsize in (1, 5):
        list(basecall(gpu_model, iter(reads), beamsize=beamsize, chunksize=read_len, overlap=0, batchsize=n_reads))      # warm-up
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        hip[beamsize] = [res for _, res in basecall(gpu_model, iter(reads), beamsize=beamsize, chunksize=read_len, overlap=0, batchsize=n_reads)]
        torch.cuda.synchronize()
        out["hip_seconds_beam%d" % beamsize] = time.perf_counter() - t0
    torch.set_num_threads(max(1, min(util.effective_cpu_count(), 16)))
    t0 = time.perf_counter()
    x = torch.stack([torch.from_numpy(r.signal) for r in reads])[:, None, :].half().float()
    with torch.no_grad():
        lp = nn_ref.ctc_forward(cpu_model, x).permute(1, 0, 2).numpy()
    t1 = time.perf_counter()
    ora = {1: [ctc_ref.viterbi_search(lp[i], gpu_model.alphabet, gpu_model.qscale, gpu_model.qbias) for i in range(n_reads)]}
    t2 = time.perf_counter()
    ora[5] = [ctc_ref.beam_search(lp[i], gpu_model.alphabet, 5, 1e-3) for i in range(n_reads)]
    t3 = time.perf_counter()
    work = n_reads * read_len
    out["value"] = work / out["hip_seconds_beam5"]
    out["value_greedy"] = work / out["hip_seconds_beam1"]
    out["unit"] = "samples/s"
    out["cpu_oracle"] = {"value": work / ((t1 - t0) + (t3 - t2)), "value_greedy": work / (t2 - t0), "unit": "samples/s",
                         "cores": torch.get_num_threads(), "kind": "port"}
    for beamsize, key in ((1, "greedy"), (5, "beam5")):
        m = c = 0
        for h, o in zip(hip[beamsize], ora[beamsize]):
            mm, cc = parity.alignment_identity(h["sequence"], o[0])
            m, c = m + mm, c + cc
        out["%s_seq_identity" % key] = (m / c) if c else 1.0
        out["%s_alignment_columns" % key] = c
        out["%s_reads_identical" % key] = int(sum(h["sequence"] == o[0] for h, o in zip(hip[beamsize], ora[beamsize])))
    g = [(h, o) for h, o in zip(hip[1], ora[1]) if h["sequence"] == o[0]]
    if g:
        out["greedy_qstring_identity_on_identical_reads"] = float(np.mean([np.mean([a == b for a, b in zip(h["qstring"], o[1])] or [1.0]) for h, o in g]))
        out["greedy_paths_identical_on_identical_reads"] = int(sum(list(h["moves"]) == list(o[2]) for h, o in g))
    # the engine's log-probabilities against the fp32 CPU forward, same chunks
    with torch.no_grad():
        glp = gpu_model(x.half().cuda()).permute(1, 0, 2).float().cpu().numpy()
    out["logp_max_abs"] = float(np.abs(glp - lp).max())
    out["logp_mean_abs"] = float(np.abs(glp - lp).mean())
    out["note"] = ("plumbing configuration (the reference runs it on PyTorch-CPU): far too small to fill a GPU - a correctness row, not a "
                   "throughput claim; fast_ctc_decode (Rust) is absent, the decoders' conventions are unpinned ([EXT])")
    return out


def config1_leg(hard_timeout=120.0):
    import subprocess
    if not os.path.exists(os.path.join(ROOT, "tests", "golden", "configs", "dna_r9.4.1@v2.toml")):
        return {"error": "tests/golden/configs/dna_r9.4.1@v2.toml (the reference's config, kept as test data) is not in this checkout"}
    code = ("import json,sys; sys.path.insert(0, %r); import bench; print('CONFIG1 ' + json.dumps(bench.config1_worker()))" % ROOT)
    try:
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=hard_timeout)
        for line in r.stdout.splitlines():
            if line.startswith("CONFIG1 "):
                return json.loads(line[8:])
        return {"error": (r.stderr or "no output")[-300:]}
    except subprocess.TimeoutExpired:
        return {"error": "exceeded %.0f s" % hard_timeout}


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: start N copies of this command, one per GPU, wired up like
    torch.distributed.run does (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT); rank 0 prints the JSON line."""
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    procs = []
    for rank in range(n):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if rank == 0 else subprocess.DEVNULL))
    rcs = [p.wait() for p in procs]
    if any(rcs):
        sys.stderr.write("bench.py: rank exit codes %s\n" % rcs)
    return 1 if any(rcs) else 0


def dry_run(a, json_out):
    """`--dry-run`: everything of an N-rank launch that is not device work - the environment contract of the launcher (RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_*), the rendezvous, the barriers that bracket the timed region, the MAX-reduce of the elapsed time, rank 0's one
    JSON line. The hot path is a sleep; the line is marked and is not a measurement."""
    import torch.distributed as dist
    from bonito_amd import parallel
    rank, world, local = parallel.init("gloo", timeout=600.0)
    if world != a.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started %d rank(s)" % (a.gpus, world))
    sys.stderr.write("bench.py: dry run, rank %d/%d (local %d)\n" % (rank, world, local))
    step_s = 2e-3

    def barrier():
        if world > 1:
            dist.barrier()

    for _ in range(a.warmup):
        time.sleep(step_s)
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        time.sleep(step_s)
    barrier()
    elapsed = parallel.max_over_ranks(time.perf_counter() - t0, device="cpu")
    if rank == 0:
        samples = a.batch * a.chunk * a.steps * world
        json_out.write(json.dumps({
            "metric": "signal samples/sec/GPU (chunk=10000, batch=512) + read accuracy vs ref", "value": samples / elapsed,
            "unit": "samples/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1e3 * elapsed / a.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f16", "data": "dry-run",
            "config": {"workload": "DRY RUN: a %.0f ms sleep per step in place of the hot path - plumbing only" % (1e3 * step_s),
                       "parallelism": "replicas x%d (shard-by-read, no collective)" % world},
            "per_gpu": samples / elapsed / world, "roofline": None, "cpu_baseline": None, "dry_run": True}) + "\n")
        json_out.flush()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(a.gpus))
    # stdout carries exactly one JSON line: keep a private handle for it and point descriptor 1 at stderr meanwhile, so
    # that nothing a library prints on stdout (gloo announces its connections there) can precede or split the line
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    import torch
    import torch.distributed as dist
    from bonito_amd import parallel
    if a.dry_run:
        return dry_run(a, json_out)
    assert torch.cuda.is_available(), "bench.py needs a HIP device; there is no CPU fallback"
    ndev = torch.cuda.device_count()
    # one process per GPU; RCCL only for barrier + MAX-reduce. More ranks than GPUs (a 1-GPU test box running `--gpus 2`):
    # ranks share devices and the two collectives go over gloo, since RCCL refuses two ranks on one device.
    oversubscribed = int(os.environ.get("WORLD_SIZE", "1")) > ndev
    # (the side legs of rank 0 - roofline spans, CPU oracle - run between two barriers: a generous bound on the default group)
    rank, world, local = parallel.init("gloo" if oversubscribed else "nccl", timeout=3600.0)
    if world != a.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started %d rank(s)" % (a.gpus, world))
    local %= ndev
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    # every rank says which device it drives (stderr: visible in the driver's tail), and with one rank per GPU the ordinals must be
    # distinct - a mis-launch (two ranks on one device, a stale visibility mask) would otherwise only show up as a bad scaling number
    props = torch.cuda.get_device_properties(local)
    ident = (rank, local, "%s" % getattr(props, "name", "?"), "%s" % (getattr(props, "pci_bus_id", None) or getattr(props, "uuid", "")))
    sys.stderr.write("bench.py: rank %d/%d -> device %d of %d (%s %s) HIP_VISIBLE_DEVICES=%s ROCR_VISIBLE_DEVICES=%s\n" % (
        rank, world, local, ndev, ident[2], ident[3], os.environ.get("HIP_VISIBLE_DEVICES"), os.environ.get("ROCR_VISIBLE_DEVICES")))
    if world > 1 and not oversubscribed:
        idents = [None] * world
        dist.all_gather_object(idents, ident)
        ordinals = sorted(i[1] for i in idents)
        if len(set(ordinals)) != world:
            raise SystemExit("bench.py: %d ranks on %d distinct device ordinal(s) %s - expected one rank per GPU" % (world, len(set(ordinals)), ordinals))

    from bonito_amd import decode
    from bonito_amd.util import limit_host_threads
    log("host threads: %d" % limit_host_threads(4))
    enc_opts = {}
    if a.quantize and a.lanes > 1 and not any(kv.startswith("lstm_q8_variant=") for kv in a.set):
        decode.set_option("lstm_q8_variant", 2)        # 8-bit recurrent kernels compiled for two workgroups per CU (as the CLI does)
    for kv in a.set:
        name, _, value = kv.partition("=")
        if name.startswith("enc:"):          # per-engine option (bh_encoder_set_option), e.g. enc:lstm_tune=16
            enc_opts[name[4:]] = int(value)
        else:
            decode.set_option(name, int(value))
    log("building model %s%s" % (a.model, " (quantize)" if a.quantize else ""))
    model = build_model(a.model, a.call_batch, a.chunk)
    model.use_koi(batchsize=a.call_batch, chunksize=a.chunk, quantize=a.quantize)
    model = model.half().to(dev)
    gen = torch.Generator(device=dev).manual_seed(25 + rank)
    signals = [torch.randn(a.call_batch, 1, a.chunk, generator=gen, device=dev).half() for _ in range(N_BATCHES)]
    host_signals = [s.cpu().pin_memory() for s in signals]
    copy_stream = torch.cuda.Stream(dev)

    # A lane = one engine replica (same seeded weights) + its encoder stream + its decoder stream + two decode contexts.
    # Within a lane the decode of batch i overlaps the encoder of batch i+1; lanes overlap whole batches with each other.
    class Lane:
        pass

    lanes = []
    for li in range(a.lanes):
        ln = Lane()
        if li == 0:
            ln.model = model
        else:
            ln.model = build_model(a.model, a.call_batch, a.chunk)
            ln.model.use_koi(batchsize=a.call_batch, chunksize=a.chunk, quantize=a.quantize)
            ln.model = ln.model.half().to(dev)
        # (measured and dropped: a higher hardware-queue priority for the encoder stream, 20.0 -> 21.1 ms per step, and
        # s_setprio 3 inside the recurrent kernel, 20.0 -> 20.0)
        ln.enc_stream, ln.dec_stream = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
        ln.stage = [torch.empty_like(signals[0]) for _ in range(2)]      # H2D landing buffers (double buffered)
        ln.stage_free = [None, None]
        ln.staged = [None, None]
        ln.tickets = [None, None]
        ln.count = 0
        lanes.append(ln)

    def stage_in(ln, k, b):
        """Enqueue the pinned-host -> device copy of batch b into landing buffer k of this lane (copy stream)."""
        with torch.cuda.stream(copy_stream):
            if ln.stage_free[k] is not None:
                copy_stream.wait_event(ln.stage_free[k])          # the forward that read this buffer has finished
            ln.stage[k].copy_(host_signals[b], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(copy_stream)
        ln.staged[k] = ev

    def encode(ln, b, h2d, b_next=None):
        k = ln.count & 1
        if h2d and ln.staged[k] is None:
            stage_in(ln, k, b)                                    # first step of the lane: nothing was prefetched
        x = ln.stage[k] if h2d else signals[b]
        with torch.cuda.stream(ln.enc_stream):
            if h2d:
                ln.enc_stream.wait_event(ln.staged[k])
                ln.staged[k] = None
            sc = ln.model(x)
            ev = torch.cuda.Event()
            ev.record(ln.enc_stream)
            if h2d:
                ln.stage_free[k] = ev
        if h2d and b_next is not None:
            # The NEXT batch of this lane goes out now, a whole step ahead, like the product pipeline's reader thread does
            # (crf/basecall.py). Issued right before its own forward the copy costs ~5 ms per step: the runtime's copy
            # kernel cannot become resident next to a persistent recurrent kernel that owns every CU's register file and
            # has to wait for the gap between two layer launches.
            stage_in(ln, k ^ 1, b_next)
        return sc, ev

    # probe output geometry, build two decode contexts per lane (double buffered pinned outputs)
    for ln in lanes:
        sc0, ev0 = encode(ln, 0, False)
        ev0.synchronize()
        for k, v in enc_opts.items():
            ln.model._hip.set_option(k, v)
        T_out, C_out = sc0.shape[1], sc0.shape[2]
        ln.decs = [decode.CRFDecoder(a.call_batch, T_out, C_out, dev, mode=a.decoder) for _ in range(2)]
        del sc0
    decs = lanes[0].decs

    def reset_lanes():
        for ln in lanes:
            ln.tickets = [None, None]
            ln.staged = [None, None]
            ln.count = 0

    def prestage(steps):
        """The first call of every lane has its batch on the way to the device BEFORE the clock starts - where the product pipeline's
        reader thread has it (crf/basecall.py stages batch k+1 while batch k computes; round 5 left it inside the region, where five
        engine calls could not hide an un-prefetched 40 MB copy)."""
        for i, ln in enumerate(lanes[:steps]):
            stage_in(ln, 0, i % N_BATCHES)

    def run(steps, h2d=False, marks=None, reset=True):
        """`steps` engine calls (a.per_call batches each) of the hot path, software-pipelined: inside a lane encoder(i+1) overlaps
        decode(i) on two HIP streams, and the lanes run round-robin. Every step's int8 outputs are on the host when this
        returns. `marks`: list that receives one timing event per step, recorded behind the step's decode + D2H."""
        if reset:
            reset_lanes()
        for i in range(steps):
            ln = lanes[i % len(lanes)]
            nxt = i + len(lanes)                   # this lane's next step
            sc, ev = encode(ln, i % N_BATCHES, h2d, nxt % N_BATCHES if nxt < steps else None)
            k = ln.count & 1
            if ln.tickets[k] is not None:
                ln.tickets[k].result()             # its pinned buffers are about to be reused
            with torch.cuda.stream(ln.dec_stream):
                ln.dec_stream.wait_event(ev)
                sc.record_stream(ln.dec_stream)
                ln.tickets[k] = ln.decs[k].submit(sc)
                if marks is not None:
                    m = torch.cuda.Event(enable_timing=True)
                    m.record(ln.dec_stream)
                    marks.append(m)
            ln.count += 1
        out = None
        for ln in lanes:
            for tk in ln.tickets:
                if tk is not None:
                    out = tk.result()
        return out

    def barrier():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier() if oversubscribed else dist.barrier(device_ids=[local])
        torch.cuda.synchronize(dev)

    def check_engines():
        for ln in lanes:
            ln.model._hip.check()

    def timed(h2d):
        """ONE timed region of exactly a.steps steps -> (elapsed seconds MAX over ranks, median ms per step, per-call ms list)."""
        calls = a.steps // a.per_call
        reset_lanes()
        if h2d:
            prestage(calls)
        barrier()
        marks = []
        t0 = time.perf_counter()
        run(calls, h2d, marks, reset=False)           # exactly a.steps batches
        barrier()
        el = time.perf_counter() - t0
        el = parallel.max_over_ranks(el, device="cpu" if oversubscribed else dev)
        check_engines()
        # marks arrive in submission order, lane after lane: the distance between a lane's consecutive calls covers one call of every lane
        nl = len(lanes)
        gaps = [marks[i].elapsed_time(marks[i + nl]) / (a.per_call * nl) for i in range(len(marks) - nl)]
        return el, (statistics.median(gaps) if gaps else 1e3 * el / a.steps), gaps

    def timed_leg(h2d, name):
        """a.repeats regions of exactly a.steps steps each; the MEDIAN region is the leg's figure (all of them are reported)."""
        regs = []
        for r in range(a.repeats):
            el, med, gaps = timed(h2d)
            regs.append((el, med))
            log("%s region %d/%d: %.3f ms/step (median step %.3f); per batch, call by call: %s"
                % (name, r + 1, a.repeats, 1e3 * el / a.steps, med, " ".join("%.2f" % g for g in gaps)))
        order = sorted(range(len(regs)), key=lambda i: regs[i][0])
        el, med = regs[order[(len(order) - 1) // 2]]          # the median region (the lower one of an even count)
        return el, med, [round(1e3 * e / a.steps, 4) for e, _ in regs]

    log("warmup")
    t_w = time.perf_counter()
    run(max(-(-a.warmup // a.per_call), 2 * len(lanes)))      # at least a.warmup batches, and two calls of every lane
    while time.perf_counter() - t_w < a.warmup_seconds:       # ... and at least --warmup-seconds: the first second runs 15-25 % slow
        run(2 * len(lanes), not a.no_h2d_leg)
    torch.cuda.synchronize(dev)
    check_engines()
    h2d = None
    if not a.no_h2d_leg:
        log("timed leg 1 (every batch copied host -> device inside its step), %d region(s) of %d steps" % (a.repeats, a.steps))
        el2, med2, regs2 = timed_leg(True, "with H2D")
        samples = a.batch * a.chunk * a.steps * world
        h2d = {"value": samples / el2, "ms_per_step": 1e3 * el2 / a.steps, "ms_per_step_median": med2, "regions_ms_per_step": regs2,
               "note": "same K steps with the fp16 batch copied pinned host -> device on a copy stream inside every step (SURVEY 8d); the "
                       "first call's batch is staged before the clock starts, as the product's reader thread has it; median of the regions"}
        log("with H2D: %.2f ms/step (median %.2f)" % (1e3 * el2 / a.steps, med2))
    log("timed leg 2 (inputs resident in HBM), %d region(s) of %d steps" % (a.repeats, a.steps))
    elapsed, med, regs = timed_leg(False, "resident")
    log("timed legs done: %.2f ms/step (median step %.2f ms)" % (1e3 * elapsed / a.steps, med))

    # ---- roofline leg: per-kernel-class HIP-event timings on the engine's stream (after the timed regions)
    def roofline_of(mdl, dec, sigs, call_batch, per_call, pipelined=None):
        """Per-kernel-class HIP-event timings on the engine's stream. Isolated: three forwards + decodes one after the other (nothing else
        on the device). In situ (`pipelined` = a callable that runs the timed legs' software pipeline): the same spans recorded while the
        decode kernels of the call before share the chip - a persistent recurrent launch then waits for the CUs their workgroups hold,
        which is real time of the step (review, round 5: rocprofv3's average over the timed region was 10 % above the isolated figure)."""
        enc = mdl._hip
        layout = enc.describe()
        situ = None
        if pipelined is not None:
            enc.profile(True)
            pipelined()
            situ = enc.profile_read()
            enc.profile(False)
        # encoder classes: three forwards with the span events on, nothing else on the device; the decode stage: three more calls with the
        # spans off (round 6: with five spans per transformer layer the span bookkeeping of a forward still in flight leaked into a decode
        # timed right behind it - 44 ms "decode" per sup batch in a 61 ms step), device idle when its clock starts
        nprof = 3
        enc.profile(True)
        for i in range(nprof):
            mdl(sigs[i % len(sigs)])
        torch.cuda.synchronize(dev)
        prof = enc.profile_read()
        enc.profile(False)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2 * nprof)]
        dec_ms = 0.0
        for i in range(nprof):
            scores = mdl(sigs[i % len(sigs)])
            torch.cuda.synchronize(dev)
            ev[2 * i].record()
            dec.submit(scores).result()
            ev[2 * i + 1].record()
        torch.cuda.synchronize(dev)
        for i in range(nprof):
            dec_ms += ev[2 * i].elapsed_time(ev[2 * i + 1])
        brk = {k: round(v[0] / nprof / per_call, 3) for k, v in prof.items() if v[1]}      # per step = per batch
        brk["decode_incl_d2h"] = round(dec_ms / nprof / per_call, 3)
        fl = flops(a.model, a.chunk)
        # the class that holds the most time among those that are ONE kernel per span ("attention" / "mlp" of the transformer are the
        # projections and norms around the attention kernel and fc1: several kernels per span, never the roofline's subject)
        cls = max((k for k in ("lstm_rec", "lstm_gemm", "crf_linear", "conv", "attention_core", "mlp_fc1") if k in fl),
                  key=lambda k: prof[k][0])
        ms, spans = prof[cls]
        launches_per_fwd = spans / nprof
        work = fl[cls]
        if cls == "lstm_rec" and prof["lstm_gemm"][1] == 0:
            work += fl["lstm_gemm"]          # fused kernel: the input projection runs inside the recurrence launch
        flops_per_launch = work * call_batch / launches_per_fwd
        avg_ms = ms / spans
        # a profiling span covers one layer of one call; the recurrent kernels of these widths serve at most 32 rings (64 paired) per
        # launch, so a call of more chunks is several launches inside the span: report per kernel launch
        per_span = 1
        if cls == "lstm_rec":
            rec = [ln for ln in layout.splitlines() if " lstm " in ln]
            if rec and "lstm_layer_wgx2_kernel" in rec[0]:
                per_span = -(-(call_batch // 16) // 64)
            elif rec and "lstm_layer_wgx_kernel" in rec[0]:
                per_span = -(-(call_batch // 16) // 32)
            elif rec and "lstm_layer_wide_kernel<32" in rec[0]:
                per_span = -(-call_batch // 256)          # 1024-wide layers: eight rings of 32 chunks (one per XCD) per launch
        flops_per_launch /= per_span
        avg_ms /= per_span
        launch_chunks = call_batch // per_span
        achieved = flops_per_launch / (avg_ms * 1e-3) / 1e12
        # the kernel names come from the engine itself (bh_encoder_describe), not from a guess about the dispatch
        kind = {"lstm_rec": " lstm ", "lstm_gemm": " lstm ", "crf_linear": " linearcrfencoder ", "conv": " conv ",
                "mlp_fc1": " transformer ", "attention_core": " transformer "}[cls]
        names = sorted({ln.split(": ", 1)[1] for ln in layout.splitlines() if kind in ln and ": " in ln})
        kernel = "; ".join(names)
        if cls == "mlp_fc1":
            kernel = "gemm_w4_kernel<0, true, 4, 0, true> (fc1 + SwiGLU epilogue, 512 -> 2 x 2048; 16x16x32 K-tile stream)"
        elif cls == "attention_core":
            kernel = "attention_ring_kernel"
        lstm_kernel = kernel if cls == "lstm_rec" else None
        q8 = "q8" in kernel
        peak = MFMA_I8_PEAK_TOPS if q8 else MFMA_F16_PEAK_TFLOPS
        traffic, traffic_src = pmc_traffic(lstm_kernel or kernel, a, launch_chunks)
        rf = {"kernel": kernel, "bound": "mfma", "achieved": round(achieved, 2), "peak": peak,
              "unit": "TOP/s" if q8 else "TFLOP/s", "frac": round(achieved / peak, 4), "traffic": traffic,
              "traffic_source": traffic_src, "avg_launch_ms": round(avg_ms, 4), "flops_per_launch": flops_per_launch,
              "chunks_per_launch": launch_chunks}
        if situ is not None and situ[cls][1] and len(lanes) == 1:          # with several lanes the launches of different lanes overlap: no per-launch meaning
            ms_s = situ[cls][0] / situ[cls][1] / per_span
            rf["avg_launch_ms_in_situ"] = round(ms_s, 4)
            rf["achieved_in_situ"] = round(flops_per_launch / (ms_s * 1e-3) / 1e12, 2)
            rf["frac_in_situ"] = round(flops_per_launch / (ms_s * 1e-3) / 1e12 / peak, 4)
            rf["in_situ_note"] = ("the same kernel's launches timed by HIP events on the encoder stream INSIDE the software pipeline of the timed "
                                  "legs (%d launches, decode kernels of the previous call sharing the chip); `frac` / `avg_launch_ms` are the "
                                  "kernel running alone" % (situ[cls][1] * per_span))
        return rf, brk

    roof = None
    breakdown = None
    if rank == 0:
        roof, breakdown = roofline_of(model, decs[0], signals, a.call_batch, a.per_call,
                                      pipelined=lambda: run(max(4 * len(lanes), min(a.steps // a.per_call, 12)), not a.no_h2d_leg))

    # ---- the product path's call shape: ONE batch per engine call (what `bonito basecaller --batchsize 512` hands the engine)
    per_call_1 = None
    if rank == 0 and world == 1 and not a.no_side_legs and a.per_call > 1:
        log("per_call_1 leg")
        lanes_main, per_call_main, call_batch_main = lanes, a.per_call, a.call_batch
        try:
            m1 = build_model(a.model, a.batch, a.chunk)
            m1.use_koi(batchsize=a.batch, chunksize=a.chunk, quantize=a.quantize)
            m1 = m1.half().to(dev)
            ln = Lane()
            ln.model = m1
            ln.enc_stream, ln.dec_stream = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
            sig1 = [sg[:a.batch] for sg in signals]
            ln.stage = [torch.empty_like(sig1[0]) for _ in range(2)]
            ln.stage_free, ln.staged, ln.tickets, ln.count = [None, None], [None, None], [None, None], 0
            sc0 = m1(sig1[0])
            torch.cuda.synchronize(dev)
            ln.decs = [decode.CRFDecoder(a.batch, sc0.shape[1], sc0.shape[2], dev, mode=a.decoder) for _ in range(2)]
            del sc0
            signals_main, host_main = signals, host_signals
            signals, host_signals = sig1, [sg.cpu().pin_memory() for sg in sig1]
            lanes, a.per_call, a.call_batch = [ln], 1, a.batch
            run(8)
            steps1 = max(8, min(a.steps, 40))
            barrier()
            marks = []
            t0 = time.perf_counter()
            run(steps1, False, marks)
            barrier()
            el1 = time.perf_counter() - t0
            m1._hip.check()
            gaps = [marks[i].elapsed_time(marks[i + 1]) for i in range(len(marks) - 1)]
            rf1, brk1 = roofline_of(m1, ln.decs[0], sig1, a.batch, 1)
            per_call_1 = {"value": a.batch * a.chunk * steps1 / el1, "ms_per_step": 1e3 * el1 / steps1, "steps": steps1,
                          "ms_per_step_median": statistics.median(gaps), "roofline": rf1, "kernel_ms_per_step": brk1,
                          "note": "one batch of %d chunks per engine call: the call shape of `bonito basecaller` at its default batchsize" % a.batch}
            signals, host_signals = signals_main, host_main
            del m1, ln
        except Exception as exc:          # a side leg must never take the headline down with it
            per_call_1 = {"error": repr(exc)}
        lanes, a.per_call, a.call_batch = lanes_main, per_call_main, call_batch_main

    if rank == 0:
        log("roofline leg done; cpu baseline + parity")
        samples = a.batch * a.chunk * a.steps * world
        cpu = par = None
        if not (a.no_cpu_baseline or world > 1):
            import tempfile
            with tempfile.TemporaryDirectory() as tmp:
                keep = os.path.join(tmp, "oracle_outputs.npz")
                cpu = cpu_baseline(a.model, a.chunk, a.decoder, keep=keep, n=a.parity_chunks) if a.parity_chunks else None
                if os.path.exists(keep):
                    try:
                        par = parity_leg(a, model, signals, decs[0], keep)
                    except Exception as exc:          # a side leg must never take the headline down with it
                        par = {"error": repr(exc)[:300]}
        out = {
            "metric": "signal samples/sec/GPU (chunk=10000, batch=512) + read accuracy vs ref",
            "value": samples / elapsed,
            "unit": "samples/s",
            "n_gpus": world,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": 1e3 * elapsed / a.steps,
            "ms_per_step_median": med,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "i8 recurrence (Q8-1) / f16" if a.quantize else "f16",
            "data": "synthetic",
            "config": {"workload": "dna_r10.4.1_e8.2_400bps_%s@v5.0.0-shaped model (seeded random weights), "
                                   "batch %d x chunk %d, %s%d distinct inputs rotating, %s decode, encoder/decoder "
                                   "software-pipelined on 2 HIP streams x %d batch lane(s) per GPU%s" %
                                   (a.model, a.batch, a.chunk,
                                    "%d batches per engine call (their rings paired in the recurrent kernels), " % a.per_call if a.per_call > 1 else "",
                                    N_BATCHES, a.decoder, a.lanes, ", --quantize" if a.quantize else ""),
                       "parallelism": "replicas x%d (shard-by-read, no collective)%s" % (world, " -- ranks SHARE devices (test mode)" if oversubscribed else "")},
            "value_definition": "`value`: input batches resident in HBM when the timed region starts (the bench contract: a PCIe-inclusive rate is "
                                "never `value`). SURVEY 8(d)'s metric counts the fp16 H2D inside the step: that is `value_with_h2d` (same steps, "
                                "same process, `h2d_over_resident` = the ratio of the two times). Each leg = the median of `repeats` regions of "
                                "exactly `steps` steps (`regions_ms_per_step`, `with_h2d.regions_ms_per_step`)",
            "repeats": a.repeats,
            "regions_ms_per_step": regs,
            "h2d_over_resident": (h2d["ms_per_step"] / (1e3 * elapsed / a.steps)) if h2d else None,
            "per_gpu": samples / elapsed / world,
            "value_with_h2d": h2d["value"] if h2d else None,
            "with_h2d": h2d,
            "batches_per_engine_call": a.per_call,
            "chunks_per_engine_call": a.call_batch,
            "roofline": roof,
            "kernel_ms_per_step": breakdown,
            "parity": par,
            "per_call_1": per_call_1,
            "e2e": None if (a.no_side_legs or world > 1 or a.model not in ("hac", "fast") or a.quantize) else e2e_leg(a.model),
            "other_configs": None if (a.no_side_legs or world > 1) else other_configs(a),
            "config1_ctc": None if (a.no_side_legs or world > 1) else config1_leg(),
            "cpu_baseline": cpu,
        }
        json_out.write(json.dumps(out) + "\n")
        json_out.flush()
    if world > 1:
        dist.barrier() if oversubscribed else dist.barrier(device_ids=[local])
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
