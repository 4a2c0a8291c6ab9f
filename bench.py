#!/usr/bin/env python3
"""
Headline benchmark: signal samples/sec/GPU at chunk=10000, batch=512 (BASELINE.json `metric`).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--model hac|fast] [--decoder viterbi|beam]

One "step" = one pass of the hot path over one resident synthetic batch: fp16 signal [512, 10000] in HBM
-> HIP encoder (conv x3, LSTM x5, LinearCRFEncoder) -> HIP CRF decode -> int8 moves/sequence/qstring on
the host.  Weights are seeded random-init tensors of the named architecture (no checkpoints offline).
For N > 1 launch with torch.distributed.run; each rank owns one GPU and the same per-GPU workload
(read chunks shard embarrassingly, no data-path collective) -> "scaling": "weak".

The JSON line also carries
  roofline     -- dominant kernel's algorithmic FLOP/s from HIP-event timings on the engine stream
  cpu_baseline -- the CPU oracle (PyTorch-CPU fp32 restatement of bonito/nn.py + C Viterbi) on a bounded sample
"""
import argparse
import json
import os
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
    return any(tok == "fast" or tok == "--model=fast" for tok in argv)       # the fast model defaults to 3 lanes


if _multi_lane(sys.argv[1:]):
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, ROOT)

import torch  # noqa: E402

MFMA_F16_PEAK_TFLOPS = 2500.0     # MI355X dense fp16/bf16 (MI355X_MICROARCH.md)
HBM_PEAK_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--model", default="hac", choices=["hac", "fast", "sup", "sup_lstm"])
    ap.add_argument("--batch", type=int, default=0, help="default: 512 (hac/fast), 256 (sup)")
    ap.add_argument("--chunk", type=int, default=0, help="default: 10000 (hac/fast), 12000 (sup)")
    ap.add_argument("--decoder", default="beam", choices=["viterbi", "beam"])
    ap.add_argument("--lanes", type=int, default=0,
                    help="independent batches in flight (each lane: own engine replica, encoder stream, decoder stream); "
                         "default 1; 3 for the narrow `fast` model whose kernels leave most CUs idle (1 lane 8.9 ms/step, "
                         "3 lanes 4.5 with GPU_MAX_HW_QUEUES=8); hac / sup kernels fill the chip and gain nothing")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--set", action="append", default=[], metavar="NAME=VALUE",
                    help="library tuning option (bh_set_option), e.g. beam_fork=1; for A/B runs, not part of the contract")
    a = ap.parse_args()
    a.batch = a.batch or (256 if a.model in ("sup", "sup_lstm") else 512)
    a.chunk = a.chunk or (12000 if a.model == "sup" else 20000 if a.model == "sup_lstm" else 10000)
    a.lanes = a.lanes or (3 if a.model == "fast" else 1)
    return a


def build_model(name, batch, chunk):
    from bonito_amd import synthetic
    if name == "sup":
        return synthetic.make_transformer_model(head_gain=4.0, batchsize=batch, chunksize=chunk)
    return synthetic.make_model(name, batchsize=batch, chunksize=chunk)


def flops(name, chunk):
    from bonito_amd import synthetic
    return synthetic.transformer_flops_per_chunk(chunksize=chunk) if name == "sup" else synthetic.flops_per_chunk(name, chunk)


def pmc_traffic(kernel, a):
    """HBM bytes per launch of the dominant kernel as measured by the committed PMC passes (profiles/pmc_traffic.json);
    None when no measurement exists for this kernel / workload."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as fh:
            ent = json.load(fh).get(kernel or "")
    except (OSError, ValueError):
        return None
    if not ent or ent.get("workload") != "%s %dx%d" % (a.model, a.batch, a.chunk):
        return None
    return ent["bytes_per_launch"]


def log(msg):
    sys.stderr.write("[bench %7.1fs] %s\n" % (time.perf_counter() - T_START, msg))
    sys.stderr.flush()


T_START = time.perf_counter()


def cpu_baseline_worker(name, chunk, seconds_budget=15.0):
    """CPU oracle timed on this host: oracle/nn_ref.py forward (fp32) + oracle/crf_oracle.c Viterbi."""
    from oracle import crf_ref, nn_ref
    model = build_model(name, 8, chunk)
    nn_ref.round_params_to_half_(model)
    from bonito_amd.util import effective_cpu_count
    ncores = max(1, min(effective_cpu_count(), 32))      # affinity capped by the cgroup quota; small matmuls stop scaling early
    torch.set_num_threads(ncores)
    n = 2 if name == "sup" else 8
    x = torch.randn(n, 1, chunk, generator=torch.Generator().manual_seed(25)).half().float()
    reps, t_total = 0, 0.0
    while t_total < seconds_budget and reps < 8:
        t0 = time.perf_counter()
        with torch.no_grad():
            y = nn_ref.forward(model.encoder, x, expand_blanks=False)
        sc = y.permute(1, 0, 2).contiguous().half().numpy()
        crf_ref.viterbi(sc, model.seqdist.state_len, blank=2.0)
        t_total += time.perf_counter() - t0
        reps += 1
    return {"value": n * chunk * reps / t_total, "unit": "samples/s", "cores": ncores, "kind": "port",
            "sample": "%d reps of %d chunks x %d samples, oracle/nn_ref.py fp32 forward + C Viterbi" % (reps, n, chunk)}


def cpu_baseline(name, chunk, hard_timeout=90.0):
    """Run the CPU leg in a child process with a hard wall-clock bound so it can never stall the bench."""
    import subprocess
    code = ("import json,sys; sys.path.insert(0, %r); import bench; "
            "print('CPU_BASELINE ' + json.dumps(bench.cpu_baseline_worker(%r, %d)))" % (ROOT, name, chunk))
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


def main():
    a = parse()
    import torch.distributed as dist
    from bonito_amd import parallel
    assert torch.cuda.is_available(), "bench.py needs a HIP device; there is no CPU fallback"
    rank, world, local = parallel.init("nccl")     # one process per GPU; RCCL only for barrier + MAX-reduce
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    from bonito_amd import decode, synthetic
    from bonito_amd.util import limit_host_threads
    log("host threads: %d" % limit_host_threads(4))
    for kv in a.set:
        name, _, value = kv.partition("=")
        decode.set_option(name, int(value))
    log("building model %s" % a.model)
    model = build_model(a.model, a.batch, a.chunk)
    model.use_koi(batchsize=a.batch, chunksize=a.chunk, quantize=False)
    model = model.half().to(dev)
    gen = torch.Generator(device=dev).manual_seed(25 + rank)
    signal = torch.randn(a.batch, 1, a.chunk, generator=gen, device=dev).half()

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
            ln.model = build_model(a.model, a.batch, a.chunk)
            ln.model.use_koi(batchsize=a.batch, chunksize=a.chunk, quantize=False)
            ln.model = ln.model.half().to(dev)
        ln.enc_stream, ln.dec_stream = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
        ln.tickets = [None, None]
        ln.count = 0
        lanes.append(ln)

    def encode(ln):
        with torch.cuda.stream(ln.enc_stream):
            sc = ln.model(signal)
            ev = torch.cuda.Event()
            ev.record(ln.enc_stream)
        return sc, ev

    # probe output geometry, build two decode contexts per lane (double buffered pinned outputs)
    for ln in lanes:
        sc0, ev0 = encode(ln)
        ev0.synchronize()
        T_out, C_out = sc0.shape[1], sc0.shape[2]
        ln.decs = [decode.CRFDecoder(a.batch, T_out, C_out, dev, mode=a.decoder) for _ in range(2)]
        del sc0
    decs = lanes[0].decs

    def run(steps):
        """`steps` passes of the hot path over one batch each, software-pipelined: inside a lane encoder(i+1) overlaps
        decode(i) on two HIP streams, and the lanes run round-robin. Every step's int8 outputs are on the host when this
        returns."""
        for ln in lanes:
            ln.tickets = [None, None]
            ln.count = 0
        for i in range(steps):
            ln = lanes[i % len(lanes)]
            sc, ev = encode(ln)
            k = ln.count & 1
            if ln.tickets[k] is not None:
                ln.tickets[k].result()             # its pinned buffers are about to be reused
            with torch.cuda.stream(ln.dec_stream):
                ln.dec_stream.wait_event(ev)
                sc.record_stream(ln.dec_stream)
                ln.tickets[k] = ln.decs[k].submit(sc)
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
            dist.barrier(device_ids=[local])
        torch.cuda.synchronize(dev)

    log("warmup")
    run(a.warmup)
    for ln in lanes:
        ln.model._hip.check()
    barrier()
    log("timed region")
    t0 = time.perf_counter()
    run(a.steps)
    barrier()
    elapsed = time.perf_counter() - t0
    elapsed = parallel.max_over_ranks(elapsed, device=dev)
    for ln in lanes:
        ln.model._hip.check()
    log("timed region done: %.1f ms/step" % (1e3 * elapsed / a.steps))

    # ---- roofline leg: per-kernel-class HIP-event timings on the engine's stream (after the timed region)
    roof = None
    breakdown = None
    if rank == 0:
        enc = model._hip
        enc.profile(True)
        nprof = 3
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2 * nprof)]
        dec_ms = 0.0
        for i in range(nprof):
            scores = model(signal)
            ev[2 * i].record()
            decs[0].submit(scores).result()
            ev[2 * i + 1].record()
        torch.cuda.synchronize(dev)
        for i in range(nprof):
            dec_ms += ev[2 * i].elapsed_time(ev[2 * i + 1])
        prof = enc.profile_read()
        enc.profile(False)
        breakdown = {k: round(v[0] / nprof, 3) for k, v in prof.items() if v[1]}
        breakdown["decode_incl_d2h"] = round(dec_ms / nprof, 3)
        fl = flops(a.model, a.chunk)
        cls = max((k for k in ("lstm_rec", "lstm_gemm", "crf_linear", "conv", "attention", "mlp") if k in fl),
                  key=lambda k: prof[k][0])
        ms, spans = prof[cls]
        launches_per_fwd = spans / nprof
        work = fl[cls]
        if cls == "lstm_rec" and prof["lstm_gemm"][1] == 0:
            work += fl["lstm_gemm"]          # fused kernel: the input projection runs inside the recurrence launch
        flops_per_launch = work * a.batch / launches_per_fwd
        avg_ms = ms / spans
        achieved = flops_per_launch / (avg_ms * 1e-3) / 1e12
        lstm_h = {"fast": 96, "hac": 384, "sup_lstm": 1024}.get(a.model, 0)
        lstm_kernel = ("lstm_layer_wide_kernel" if lstm_h > 512 else "lstm_layer_kernel" if prof["lstm_gemm"][1] else
                       "lstm_layer_wg_kernel" if lstm_h and (lstm_h % 48 == 0 or lstm_h in (64, 128, 256)) else "lstm_layer_fused_kernel")
        roof = {"kernel": {"lstm_rec": lstm_kernel, "lstm_gemm": "gemm_kernel", "crf_linear": "gemm_kernel",
                           "conv": "conv_igemm_kernel", "mlp": "gemm_kernel (fc1 gated + fc2) + rmsnorm_residual_kernel",
                           "attention": "gemm_kernel (Wqkv, out_proj) + attention_kernel + rmsnorm_residual_kernel"}[cls],
                "bound": "mfma", "achieved": round(achieved, 2), "peak": MFMA_F16_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": round(achieved / MFMA_F16_PEAK_TFLOPS, 4), "traffic": pmc_traffic(lstm_kernel if cls == "lstm_rec" else None, a),
                "avg_launch_ms": round(avg_ms, 4), "flops_per_launch": flops_per_launch}

    if rank == 0:
        log("roofline leg done; cpu baseline")
        samples = a.batch * a.chunk * a.steps * world
        out = {
            "metric": "signal samples/sec/GPU (chunk=10000, batch=512) + read accuracy vs ref",
            "value": samples / elapsed,
            "unit": "samples/s",
            "n_gpus": world,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": 1e3 * elapsed / a.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f16",
            "data": "synthetic",
            "config": {"workload": "dna_r10.4.1_e8.2_400bps_%s@v5.0.0-shaped model (seeded random weights), "
                                   "batch %d x chunk %d, %s decode, encoder/decoder software-pipelined on 2 HIP streams x %d batch lane(s) per GPU" %
                                   (a.model, a.batch, a.chunk, a.decoder, a.lanes),
                       "parallelism": "replicas x%d (shard-by-read, no collective)" % world},
            "per_gpu": samples / elapsed / world,
            "roofline": roof,
            "kernel_ms_per_step": breakdown,
            "cpu_baseline": None if (a.no_cpu_baseline or world > 1) else cpu_baseline(a.model, a.chunk),
        }
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
