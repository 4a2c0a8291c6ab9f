"""
CRF basecalling pipeline on the MI355X engine: chunk -> batch -> HIP encoder + HIP decode -> unbatch ->
stitch -> format. Same function names, arguments and result dictionaries as /root/reference
bonito/crf/basecall.py (stitch_results 13-24, compute_scores 27-45, fmt 48-55, basecall 58-82) so
``load_symbol(config, "basecall")`` callers (cli/basecaller.py:71,131-136) need no change.
"""
import numpy as np
import torch

from bonito_amd import decode as hip_decode
from bonito_amd.decode import to_str
from bonito_amd.multiprocessing import thread_iter
from bonito_amd.util import chunk, stitch, batchify, unbatchify


def stitch_results(results, length, size, overlap, stride, reverse=False):
    """Stitch per-chunk decode outputs ([n_chunks, T] int8 each) back into one read."""
    if isinstance(results, dict):
        return {k: stitch_results(v, length, size, overlap, stride, reverse=reverse) for k, v in results.items()}
    if length < size:
        return results[0, :int(np.floor(length / stride))]
    return stitch(results, size, overlap, length, stride, reverse=reverse)


def compute_scores(model, batch, beam_width=32, beam_cut=100.0, scale=1.0, offset=0.0, blank_score=2.0,
                   reverse=False, decoder="beam"):
    """fp16 forward on the HIP engine followed by the HIP decoder; returns CPU int8 [N, T] tensors
    `sequence`, `qstring`, `moves` (zero where nothing is emitted), as koi.decode.beam_search does."""
    with torch.inference_mode():
        device = next(model.parameters()).device
        scores = model(batch.to(torch.float16).to(device))
        if reverse:
            scores = model.seqdist.reverse_complement(scores)
        with torch.cuda.device(scores.device):
            if decoder == "viterbi":
                moves, path = hip_decode.viterbi(scores, blank_score=blank_score)
                sequence = hip_decode.path_to_sequence(path)
                qstring = torch.where(sequence != 0, torch.tensor(33 + 20, dtype=torch.int8), torch.tensor(0, dtype=torch.int8))
            else:
                sequence, qstring, moves = hip_decode.beam_search(
                    scores, beam_width=beam_width, beam_cut=beam_cut, scale=scale, offset=offset,
                    blank_score=blank_score)
        return {"moves": moves, "qstring": qstring, "sequence": sequence}


def fmt(stride, attrs, rna=False):
    flip = (lambda s: s[::-1]) if rna else (lambda s: s)
    return {
        "stride": stride,
        "moves": attrs["moves"].numpy(),
        "qstring": flip(to_str(attrs["qstring"])),
        "sequence": flip(to_str(attrs["sequence"])),
    }


class _Pipeline:
    """Two-stage device pipeline: the encoder runs on one HIP stream (thread 1), the CRF decode + int8 D2H
    on another (thread 2), so decode of batch i overlaps the encoder of batch i+1. The reference gets the
    same overlap from its per-stage ThreadIterators (bonito/crf/basecall.py:63-82) with koi returning CPU
    tensors; here the split is explicit because both halves are ours."""

    def __init__(self, model, decoder="beam", reverse=False, **decode_kw):
        self.model, self.mode, self.kw, self.reverse = model, decoder, decode_kw, reverse
        self.device = next(model.parameters()).device
        self.enc_stream = torch.cuda.Stream(self.device)
        self.dec_stream = torch.cuda.Stream(self.device)
        self.decoders = {}

    def encode(self, batch):
        with torch.inference_mode(), torch.cuda.stream(self.enc_stream):
            scores = self.model(batch.to(torch.float16).to(self.device, non_blocking=True))
            if self.reverse:
                scores = self.model.seqdist.reverse_complement(scores)
            ready = torch.cuda.Event()
            ready.record(self.enc_stream)
        return scores, ready

    def decode(self, scores, ready):
        key = tuple(scores.shape[1:])
        dec = self.decoders.get(key)
        if dec is None or dec.N < scores.shape[0]:
            cfg = getattr(self.model, "config", None) or {}
            nmax = max(scores.shape[0], int(cfg.get("basecaller", {}).get("batchsize", 0) or 0))
            dec = self.decoders[key] = hip_decode.CRFDecoder(nmax, key[0], key[1], self.device, mode=self.mode, **self.kw)
        with torch.inference_mode(), torch.cuda.stream(self.dec_stream):
            self.dec_stream.wait_event(ready)
            scores.record_stream(self.dec_stream)
            ticket = dec.submit(scores)
        sequence, qstring, moves = ticket.result()
        if self.mode == "viterbi":
            path = qstring            # plane 1 carries the path for the Viterbi decoder
            sequence = hip_decode.path_to_sequence(path)
            qstring = torch.where(sequence != 0, torch.tensor(33 + 20, dtype=torch.int8), torch.tensor(0, dtype=torch.int8))
        return {"moves": moves, "qstring": qstring, "sequence": sequence}


def basecall(model, reads, chunksize=4000, overlap=100, batchsize=32, reverse=False, rna=False, decoder="beam"):
    """Basecalls a set of reads: yields (read, {sequence, qstring, moves, stride})."""
    pipe = _Pipeline(model, decoder=decoder, reverse=reverse)
    chunks = thread_iter(
        ((read, 0, read.signal.shape[-1]), chunk(torch.from_numpy(read.signal), chunksize, overlap))
        for read in reads
    )
    batches = thread_iter(batchify(chunks, batchsize=batchsize))
    encoded = thread_iter((keys, pipe.encode(batch)) for keys, batch in batches)
    scores = thread_iter((keys, pipe.decode(*enc)) for keys, enc in encoded)
    results = thread_iter(
        (read, stitch_results(sc, end - start, chunksize, overlap, model.stride, reverse))
        for ((read, start, end), sc) in unbatchify(scores)
    )
    return thread_iter((read, fmt(model.stride, attrs, rna)) for read, attrs in results)
