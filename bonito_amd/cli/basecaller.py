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


def main(args):
    util.init(args.seed, args.device)
    util.limit_host_threads(8)       # host work is small copies; never out-spin a container's CPU quota
    try:
        reader = Reader(args.reads_directory, args.recursive)
        sys.stderr.write("> reading %s\n" % args.reads_directory)
    except FileNotFoundError as exc:
        sys.stderr.write("> error: %s\n" % exc)
        return 1
    sys.stderr.write("> loading model %s\n" % args.model_directory)
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
                             n_max=args.max_reads or None, raw=args.device_ingest)
    if args.device_ingest:            # int16 reads: pA scaling, normalisation, trim and chunking on the GPU
        from bonito_amd.crf.basecall import basecall_raw
        pa = (model.config.get("scaling") or {}).get("strategy") == "pa"
        results = basecall_raw(model, reads, reverse=args.revcomp, rna=args.rna, batchsize=bc["batchsize"],
                               chunksize=bc["chunksize"], overlap=bc["overlap"], scaling_strategy=model.config.get("scaling"),
                               norm_params=model.config.get("standardisation") if pa else model.config.get("normalisation"),
                               do_trim=not args.no_trim)
    else:
        results = basecall(model, reads, reverse=args.revcomp, rna=args.rna, batchsize=bc["batchsize"],
                           chunksize=bc["chunksize"], overlap=bc["overlap"])
    mode = "sam" if args.sam else ("fasta" if args.fasta else "fastq")
    writer = Writer(mode, results, fd=sys.stdout, min_qscore=args.min_qscore,
                    summary_path=None if args.no_summary else args.summary)
    t0 = perf_counter()
    writer.start()
    writer.join()
    duration = perf_counter() - t0
    if writer.error is not None:
        raise writer.error
    num_samples = sum(n for _, n in writer.log)
    sys.stderr.write("> completed reads: %s\n" % len(writer.log))
    sys.stderr.write("> duration: %s\n" % timedelta(seconds=np.round(duration)))
    sys.stderr.write("> samples per second %.1E\n" % (num_samples / max(duration, 1e-9)))
    sys.stderr.write("> done\n")
    return 0


def argparser():
    parser = ArgumentParser(formatter_class=ArgumentDefaultsHelpFormatter, add_help=False)
    parser.add_argument("model_directory")
    parser.add_argument("reads_directory")
    parser.add_argument("--device", default="cuda")
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
    parser.add_argument("--min-qscore", default=0.0, type=float)
    parser.add_argument("--sam", action="store_true", default=False, help="write unaligned SAM instead of FASTQ")
    parser.add_argument("--fasta", action="store_true", default=False)
    parser.add_argument("--summary", default="summary.tsv")
    parser.add_argument("--no-summary", action="store_true", default=False)
    return parser
