"""
CTC basecalling pipeline on the MI355X engine -- same signature and results dictionary as /root/reference
bonito/ctc/basecall.py (basecall 14-29, compute_scores 32-40, decode 43-61): chunk -> batch -> HIP forward
-> unbatch -> stitch the per-step log-probabilities -> CTC decode of every stitched read.
"""
from functools import partial

import torch

from bonito_amd.util import mean_qscore_from_qstring
from bonito_amd.util import chunk, stitch, batchify, unbatchify, permute


def basecall(model, reads, beamsize=5, chunksize=0, overlap=0, batchsize=1, qscores=False, reverse=None, rna=None):
    """Basecalls a set of reads (`reverse` / `rna` are accepted and ignored like the reference's `reverse`)."""
    chunks = ((read, chunk(torch.tensor(read.signal), chunksize, overlap)) for read in reads)
    scores = unbatchify((k, compute_scores(model, v)) for k, v in batchify(chunks, batchsize))
    scores = ((read, {"scores": stitch(v, chunksize, overlap, len(read.signal), model.stride)}) for read, v in scores)
    if hasattr(model, "alphabet") and hasattr(model, "qscale"):
        return decode_grouped(model, scores, beamsize=beamsize, qscores=qscores)
    decoder = partial(decode, decode=model.decode, beamsize=beamsize, qscores=qscores, stride=model.stride)
    return ((read, decoder(v)) for read, v in scores)


def decode_grouped(model, scores, beamsize=5, qscores=False, group=64):
    """`decode` below for up to `group` stitched reads per launch: the HIP CTC decoders run one lane per read (csrc/ctc.hip), so a launch
    per read - the reference's call pattern, a Rust call per read on the CPU - leaves 63 lanes idle and pays a serial search per read.
    Same results, same order (every read is decoded independently; tests/test_gpu_ctc.py compares the two paths)."""
    from bonito_amd.ctc import decode as ctc_decode

    def flush(pending):
        lps = [v["scores"] for _, v in pending]
        greedy = ctc_decode.viterbi_search_batch(lps, model.alphabet, model.qscale, model.qbias)
        beams = None
        if not (qscores or beamsize == 1):
            beams = ctc_decode.beam_search_batch(lps, model.alphabet, beamsize, 1e-3)
        for i, (read, _) in enumerate(pending):
            seq, qstring, path = greedy[i]
            mean_qscore_from_qstring(qstring)
            if beams is not None:
                seq, path, qstring = beams[i][0], None, "*"
            yield read, {"sequence": seq, "qstring": qstring, "stride": model.stride, "moves": path}

    pending = []
    for item in scores:
        pending.append(item)
        if len(pending) >= group:
            yield from flush(pending)
            pending = []
    if pending:
        yield from flush(pending)


def compute_scores(model, batch):
    """fp16 forward on the HIP engine; returns float32 log-probabilities [N, T, n_labels] on the host."""
    with torch.no_grad():
        device = next(model.parameters()).device
        probs = permute(model(batch.to(torch.half).to(device)), "TNC", "NTC")
    return probs.cpu().to(torch.float32)


def decode(scores, decode, beamsize=5, qscores=False, stride=1):
    """Greedy decode for a qstring / move path, then (optionally) prefix beam search for the sequence."""
    seq, path = decode(scores["scores"], beamsize=1, qscores=True, return_path=True)
    seq, qstring = seq[:len(path)], seq[len(path):]
    mean_qscore = mean_qscore_from_qstring(qstring)  # noqa: F841  (kept for parity with the reference flow)
    if not (qscores or beamsize == 1):
        try:
            seq = decode(scores["scores"], beamsize=beamsize)
            path = None
            qstring = "*"
        except NotImplementedError:
            pass
    return {"sequence": seq, "qstring": qstring, "stride": stride, "moves": path}
