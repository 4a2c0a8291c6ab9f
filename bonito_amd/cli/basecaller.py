"""
``python -m bonito_amd basecaller <model_directory> <reads_directory> > calls.fastq``

The thin CLI around the hot path, with the flags of /root/reference bonito/cli/basecaller.py:168-199 that make
sense without the alignment / BAM stack: it wires Reader -> load_model(use_koi=True) -> fuse_bn_ -> basecall ->
Writer and reports ``samples per second`` exactly like the reference (cli/basecaller.py:156-164: post-trim samples
of the written reads over the wall time of the writer loop).
"""
import os
import sys
from argparse import ArgumentDefaultsHelpFormatter, ArgumentParser
from datetime import timedelta
from time import perf_counter

import numpy as np

from bonito_amd import util
from bonito_amd.io import Writer
from bonito_amd.nn import fuse_bn_
from bonito_amd.reader import Reader


def parse_devices(spec):
    """"0-7" / "0,2,5" / "0-3,6" -> list of device indices (duplicates allowed: two ranks may share a GPU)."""
    out = []
    for part in str(spec).split(","):
        part = part.strip()
        if not part:
            continue
        if "-" in part:
            lo, hi = part.split("-", 1)
            out.extend(range(int(lo), int(hi) + 1))
        else:
            out.append(int(part))
    if not out:
        raise ValueError("--devices: no device in %r" % (spec,))
    return out


def child_device_mask(dev, environ):
    """HIP_VISIBLE_DEVICES for the worker that gets `--devices` entry `dev`. Indices are relative to what THIS process may see.
    HIP numbers its devices WITHIN the set ROCR_VISIBLE_DEVICES leaves (and that variable stays in the child's environment), so:
    parent mask in HIP_VISIBLE_DEVICES (with or without a ROCR mask underneath): entry `dev` of that list - its values already
    are indices into the ROCR set; ROCR mask only, or no mask: `dev` itself, bounds-checked against the ROCR list (advisor
    finding, round 3: translating through a ROCR-only mask gave the children indices of the physical numbering on top of the
    filtered set - no device)."""
    hip = [d.strip() for d in (environ.get("HIP_VISIBLE_DEVICES") or "").split(",") if d.strip()]
    rocr = [d.strip() for d in (environ.get("ROCR_VISIBLE_DEVICES") or "").split(",") if d.strip()]
    if hip:
        if dev >= len(hip):
            raise SystemExit("> error: --devices names device %d but only %d are visible (HIP_VISIBLE_DEVICES=%s)" % (dev, len(hip), ",".join(hip)))
        return hip[dev]
    if rocr and dev >= len(rocr):
        raise SystemExit("> error: --devices names device %d but only %d are visible (ROCR_VISIBLE_DEVICES=%s)" % (dev, len(rocr), ",".join(rocr)))
    return str(dev)


def launch(args, argv):
    """``--devices``: one worker process per listed GPU (the reference is single-device; SURVEY 8e: shard by read, no
    collective). Each worker sees only its GPU (HIP_VISIBLE_DEVICES), takes the reads whose index is congruent to its rank,
    formats its own records, and rank 0 merges the streams in input order and writes (bonito_amd/parallel.py)."""
    import socket
    import subprocess
    devices = parse_devices(args.devices)
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    child_argv, skip = [], False
    for tok in argv:                       # the workers get the same command line minus --devices / --device (each sees ONE device)
        if skip:
            skip = False
        elif tok in ("--devices", "--device"):
            skip = True
        elif not tok.startswith(("--devices=", "--device=")):
            child_argv.append(tok)
    procs = []
    import tempfile
    ready_dir = tempfile.mkdtemp(prefix="bonito_amd_")
    ready = os.path.join(ready_dir, "rendezvous_done")
    for rank, dev in enumerate(devices):
        dev = child_device_mask(dev, os.environ)
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(len(devices)), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HIP_VISIBLE_DEVICES=str(dev), BONITO_AMD_SPAWNED="1", BONITO_AMD_READY_FILE=ready)
        env.pop("CUDA_VISIBLE_DEVICES", None)
        pkg_root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        env["PYTHONPATH"] = pkg_root + (os.pathsep + env["PYTHONPATH"] if env.get("PYTHONPATH") else "")
        procs.append(subprocess.Popen([sys.executable, "-m", "bonito_amd", "basecaller", *child_argv], env=env,
                                      stdout=None if rank == 0 else subprocess.DEVNULL))
    # Two phases (advisor, round 5). BEFORE rank 0 reports that every rank has joined (parallel.rendezvous_done): a worker that exits -
    # out of memory, a bad device, an import error - can only leave the others waiting in the rendezvous, so everything is taken down
    # at once. AFTER: a worker other than rank 0 that dies does not take the run down - rank 0 keeps the records it had received from
    # it and basecalls the rest of that worker's shard itself (parallel.ordered_records, `rescue`), the other workers carry on. Rank 0
    # is the writer: if IT fails, the rest is taken down with it. Exit code: 0 = complete, every rank alive to the end; 3 = complete
    # (every read basecalled and written) but a worker was lost on the way - a scheduler can tell; 1 = failed.
    import shutil
    import time

    def take_down(rcs):
        for i, pr in enumerate(procs):
            if rcs[i] is None:
                pr.terminate()
        for i, pr in enumerate(procs):
            if rcs[i] is None:
                try:
                    rcs[i] = pr.wait(timeout=10)
                except Exception:
                    pr.kill()
                    rcs[i] = pr.wait()

    rcs = [None] * len(procs)
    try:
        while any(rc is None for rc in rcs):
            joined = os.path.exists(ready)
            for i, pr in enumerate(procs):
                if rcs[i] is None:
                    rcs[i] = pr.poll()
                    if rcs[i] not in (None, 0) and i != 0 and joined:
                        sys.stderr.write("> warning: worker rank %d (device %s) exited with code %d: rank 0 takes over its reads\n"
                                         % (i, devices[i], rcs[i]))
            early = [(i, rc) for i, rc in enumerate(rcs) if rc not in (None, 0)] if not joined else []
            if early:
                sys.stderr.write("> error: worker rank %d exited with code %d before every rank had joined: stopping all workers\n" % early[0])
                take_down(rcs)
                return 1
            if rcs[0] not in (None, 0):
                take_down(rcs)
                break
            time.sleep(0.05)
    finally:
        shutil.rmtree(ready_dir, ignore_errors=True)
    if rcs[0] != 0:
        sys.stderr.write("> error: the writing worker (rank 0) failed with code %s\n" % rcs[0])
        return 1
    bad = [(r, rc) for r, rc in enumerate(rcs) if rc != 0]
    if bad:
        sys.stderr.write("> warning: completed WITHOUT %s; every read was basecalled and written (exit code 3)\n"
                         % ", ".join("rank %d (rc %d)" % b for b in bad))
        return 3
    return 0


def main(args, argv=None):
    from bonito_amd import parallel
    # several lanes (narrow models, the 8-bit path) need more hardware queues than the runtime's default of four, or their streams
    # share queues and serialise; must be set before the HIP runtime starts. Measured harmless for the one-lane models.
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    if getattr(args, "devices", None) and "RANK" not in os.environ:
        return launch(args, list(argv if argv is not None else getattr(args, "_argv", sys.argv[2:])))
    rank, world, local = parallel.env_rank_world()
    out = sys.stdout
    if world > 1:
        # stdout carries the records: keep a private handle to it and point file descriptor 1 at stderr, so that nothing a
        # library prints (gloo announces its connections on stdout) can end up inside the FASTQ / SAM stream
        sys.stdout.flush()
        out = os.fdopen(os.dup(1), "w")
        os.dup2(2, 1)
        # one process per GPU: spawned by `launch` (sees one device) or by torchrun (device = LOCAL_RANK)
        if not os.environ.get("BONITO_AMD_SPAWNED") and args.device == "cuda":
            args.device = "cuda:%d" % local
        _fault_hook_early(rank)
        parallel.init("gloo")              # host objects only: there is no device collective on this path
        parallel.host_group()              # every rank joins here (bounded by the rendezvous timeout); rank 0 then tells the launcher
    util.init(args.seed, args.device)
    util.limit_host_threads(8)       # host work is small copies; never out-spin a container's CPU quota
    log = sys.stderr.write if rank == 0 else (lambda _msg: None)
    try:
        reader = Reader(args.reads_directory, args.recursive)
        log("> reading %s\n" % args.reads_directory)
    except FileNotFoundError as exc:
        sys.stderr.write("> error: %s\n" % exc)
        return 1
    log("> loading model %s\n" % args.model_directory)
    model = util.load_model(args.model_directory, args.device, weights=args.weights if args.weights > 0 else None,
                            chunksize=args.chunksize, overlap=args.overlap, batchsize=args.batchsize,
                            quantize=args.quantize, use_koi=True)
    model = model.apply(fuse_bn_)
    basecall = util.load_symbol(args.model_directory, "basecall")
    bc = model.config["basecaller"]
    read_ids = None
    if args.read_ids:
        with open(args.read_ids) as fh:
            read_ids = {line.strip().split()[0] for line in fh if line.strip()}
    reads = reader.get_reads(read_ids=read_ids, skip=args.skip, do_trim=not args.no_trim,
                             scaling_strategy=model.config.get("scaling"),
                             norm_params=model.config.get("standardisation") if (model.config.get("scaling") or {}).get(
                                 "strategy") == "pa" else model.config.get("normalisation"),
                             n_max=args.max_reads or None, raw=args.device_ingest, rank=rank, world=world)
    mode = "sam" if args.sam else ("fasta" if args.fasta else "fastq")
    pa = (model.config.get("scaling") or {}).get("strategy") == "pa"
    raw_kw = dict(scaling_strategy=model.config.get("scaling"),
                  norm_params=model.config.get("standardisation") if pa else model.config.get("normalisation"),
                  do_trim=not args.no_trim) if args.device_ingest else None    # int16 reads: pA scaling, normalisation, trim, chunking on the GPU
    import importlib
    records_fn = getattr(importlib.import_module(basecall.__module__), "basecall_records", None)
    if records_fn is None and args.device_ingest:
        raise SystemExit("> error: --device-ingest needs a CRF model")

    def make_records(mdl, rds):
        if records_fn is not None:
            # CRF family: stitching, string compaction and the record text of a read are ONE library call (crf/basecall.py
            # records_from_planes); the triples are what io.format_record makes of basecall()'s results, byte for byte
            return records_fn(mdl, rds, mode, reverse=args.revcomp, rna=args.rna, batchsize=bc["batchsize"],
                              chunksize=bc["chunksize"], overlap=bc["overlap"], lanes=args.lanes, per_call=args.per_call,
                              min_qscore=args.min_qscore, raw=raw_kw)
        names = basecall.__code__.co_varnames
        kw = {"lanes": args.lanes} if args.lanes > 1 and "lanes" in names else {}
        results = basecall(mdl, rds, reverse=args.revcomp, rna=args.rna, batchsize=bc["batchsize"],
                           chunksize=bc["chunksize"], overlap=bc["overlap"], **kw)
        return parallel.format_stream(results, mode, args.min_qscore)

    records = make_records(model, reads)
    records = _fault_hook(records, rank, world)
    t0 = perf_counter()
    lost_ranks = []
    if world > 1:
        def rescue(r, k):
            """Rank r is gone after k records: the rest of ITS shard, from this process - a second engine (own workspace) on this
            rank's GPU, fed by the same reader shard the dead worker had (records are idempotent and keyed by their index)."""
            import itertools
            sys.stderr.write("> warning: rank %d is gone after %d records; rank 0 basecalls the rest of its reads\n" % (r, k))
            lost_ranks.append(r)
            m2 = util.load_model(args.model_directory, args.device, weights=args.weights if args.weights > 0 else None,
                                 chunksize=args.chunksize, overlap=args.overlap, batchsize=args.batchsize,
                                 quantize=args.quantize, use_koi=True).apply(fuse_bn_)
            theirs = reader.get_reads(read_ids=read_ids, skip=args.skip, do_trim=not args.no_trim,
                                      scaling_strategy=model.config.get("scaling"),
                                      norm_params=model.config.get("standardisation") if pa else model.config.get("normalisation"),
                                      n_max=args.max_reads or None, raw=args.device_ingest, rank=r, world=world)
            try:
                yield from make_records(m2, itertools.islice(theirs, k, None))
            finally:
                # the stand-in engine (its workspace, its share of the score tensors) goes when its shard is done: two resident engines for
                # the rest of the run could exhaust rank 0's memory, and rank 0 is the only writer (advisor, round 5)
                eng = getattr(m2, "_hip", None)
                if eng is not None:
                    eng.close()
                del m2
                import torch
                torch.cuda.empty_cache()

        # every rank formats its own records; rank 0 merges the streams in input order and is the only writer. The streams end with a
        # closing message from rank 0 (no barrier: it would wait for a worker that died)
        records = parallel.ordered_records(records, rank, world, rescue=rescue, packed=True)
        if rank != 0:
            parallel.shutdown()
            return 0
    writer = Writer(mode, records, fd=out, summary_path=None if args.no_summary else args.summary, preformatted=True)
    writer.start()
    writer.join()
    duration = perf_counter() - t0
    if writer.error is not None:
        raise writer.error
    num_samples = sum(n for _, n in writer.log)
    sys.stderr.write("> completed reads: %s\n" % len(writer.log))
    sys.stderr.write("> duration: %s\n" % timedelta(seconds=np.round(duration)))
    sys.stderr.write("> samples per second %.1E\n" % (num_samples / max(duration, 1e-9)))
    if world > 1:
        sys.stderr.write("> devices: %d (one process per GPU, reads sharded round-robin)%s\n"
                         % (world, "; lost on the way: rank(s) %s" % sorted(set(lost_ranks)) if lost_ranks else ""))
    sys.stderr.write("> done\n")
    if world > 1:
        parallel.shutdown()
    return 0


def _fault_hook(records, rank, world):
    """TEST HOOK of the failure-detection path (tests/test_gpu_basecall.py): with BONITO_AMD_TEST_HOOKS=1 AND BONITO_AMD_FAULT_INJECT=
    "<rank>:<count>" that worker exits hard after `count` records. Without the first variable the second is ignored; a malformed value
    is an error at start-up (not a crash in every worker half-way), and an armed hook says so on stderr (advisor, round 5)."""
    fault = os.environ.get("BONITO_AMD_FAULT_INJECT", "")
    if not fault:
        return records
    if os.environ.get("BONITO_AMD_TEST_HOOKS") != "1":
        sys.stderr.write("> warning: BONITO_AMD_FAULT_INJECT is set but BONITO_AMD_TEST_HOOKS is not 1: ignored\n")
        return records
    if fault.endswith(":early"):
        return records                     # handled before the rendezvous (_fault_hook_early)
    try:
        victim, count = (int(tok) for tok in fault.split(":"))
        if victim < 0 or count < 0:
            raise ValueError
    except ValueError:
        raise SystemExit("> error: BONITO_AMD_FAULT_INJECT must be <rank>:<records> or <rank>:early, got %r" % fault)
    if world < 2 or victim != rank:
        return records
    sys.stderr.write("> WARNING: TEST HOOK ARMED - rank %d will exit hard after %d records (BONITO_AMD_FAULT_INJECT)\n" % (rank, count))

    def dying(recs):
        for k, rec in enumerate(recs):
            if k == count:
                os._exit(7)
            yield rec
    return dying(records)


def _fault_hook_early(rank):
    """TEST HOOK: BONITO_AMD_FAULT_INJECT="<rank>:early" (with BONITO_AMD_TEST_HOOKS=1) - that worker exits BEFORE joining the rendezvous,
    the case in which the launcher must stop everything at once (tests/test_cli_cpu.py)."""
    fault = os.environ.get("BONITO_AMD_FAULT_INJECT", "")
    if os.environ.get("BONITO_AMD_TEST_HOOKS") == "1" and fault.endswith(":early") and fault.split(":")[0] == str(rank):
        sys.stderr.write("> WARNING: TEST HOOK ARMED - rank %d exits before the rendezvous (BONITO_AMD_FAULT_INJECT)\n" % rank)
        os._exit(9)


def argparser():
    parser = ArgumentParser(formatter_class=ArgumentDefaultsHelpFormatter, add_help=False)
    parser.add_argument("model_directory")
    parser.add_argument("reads_directory")
    parser.add_argument("--device", default="cuda")
    parser.add_argument("--devices", default=None,
                        help="multi-GPU: device list such as 0-7 or 0,2,5; one worker process per GPU, reads sharded "
                             "round-robin, rank 0 writes all records in input order")
    parser.add_argument("--seed", default=25, type=int)
    parser.add_argument("--weights", default=0, type=int)
    parser.add_argument("--read-ids")
    parser.add_argument("--skip", action="store_true", default=False)
    parser.add_argument("--no-trim", action="store_true", default=False)
    parser.add_argument("--revcomp", action="store_true", default=False)
    parser.add_argument("--rna", action="store_true", default=False)
    parser.add_argument("--recursive", action="store_true", default=False)
    parser.add_argument("--device-ingest", action="store_true", default=False,
                        help="int16 .npy reads: scale / normalise / trim / chunk on the GPU instead of in numpy")
    quant = parser.add_mutually_exclusive_group()
    quant.add_argument("--quantize", dest="quantize", action="store_true")
    quant.add_argument("--no-quantize", dest="quantize", action="store_false")
    parser.set_defaults(quantize=None)
    parser.add_argument("--overlap", default=None, type=int)
    parser.add_argument("--chunksize", default=None, type=int)
    parser.add_argument("--batchsize", default=None, type=int)
    parser.add_argument("--max-reads", default=0, type=int)
    parser.add_argument("--lanes", default=0, type=int,
                        help="batches in flight in the encoder (engine replicas per GPU); 0 = automatic: 2 with the 8-bit recurrent "
                             "path (--quantize at 384 hidden units: the kernels of two lanes share every CU), 3 for the narrow (64 / 96 / 128 "
                             "wide) models whose recurrent kernel fills an eighth of the chip, 1 otherwise")
    parser.add_argument("--per-call", default=0, type=int,
                        help="batches of --batchsize chunks per engine call; 0 = automatic (calls of up to 2048 chunks for the "
                             "192...512-wide fp16 models: the recurrent kernel pairs rings, 2.97 -> 1.8 ms per layer and 512 chunks at "
                             "384 hidden units, and two such launches per layer keep the pipeline full; results do not depend on it)")
    parser.add_argument("--min-qscore", default=0.0, type=float)
    parser.add_argument("--sam", action="store_true", default=False, help="write unaligned SAM instead of FASTQ")
    parser.add_argument("--fasta", action="store_true", default=False)
    parser.add_argument("--summary", default="summary.tsv")
    parser.add_argument("--no-summary", action="store_true", default=False)
    return parser
