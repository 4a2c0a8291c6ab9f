"""
CRF decoding entry points with the call surface of ``koi.decode`` (the un-vendored dependency the
reference calls at /root/reference bonito/crf/basecall.py:7,36-40,48-55), backed by the HIP kernels in
bonito_amd/csrc/crf.hip through the C ABI (``bh_crf_viterbi`` ...).
"""
import ctypes as C

import numpy as np
import torch

from bonito_amd import _lib

_ALPHABET = np.frombuffer(b"NACGT", dtype=np.uint8)


def _check_scores(scores):
    if scores.dtype != torch.float16:
        raise TypeError("Expected fp16 but received %s" % scores.dtype)   # koi raises TypeError too
    if not scores.is_cuda:
        raise _lib.HipEngineError("scores must live on a HIP device (no CPU decode fallback)")
    if not scores.is_contiguous():
        raise AssertionError("scores must be contiguous [N, T, C]")


def state_len_of(C_, n_base=4):
    """state_len such that C == n_base^(state_len+1) (koi layout, expand_blanks=False)."""
    sl, size = 0, n_base
    while size < C_:
        size *= n_base
        sl += 1
    if size != C_:
        raise ValueError("score width %d is not a power of %d" % (C_, n_base))
    return sl


class CRFDecoder:
    """Pre-allocated decode context (workspace, device + pinned host output planes) for batches of up
    to (max_batch, T, C). ``submit(scores)`` enqueues the HIP decode and the int8 D2H copy on the current
    stream and returns a ticket; ``ticket.result()`` waits and returns CPU tensors. Used by the pipelined
    basecaller / bench so that decode of batch i overlaps the encoder of batch i+1 on another stream."""

    def __init__(self, max_batch, T, C, device, mode="beam", beam_width=32, beam_cut=100.0, scale=1.0, offset=0.0,
                 blank_score=2.0):
        lib = _lib.lib()
        self.mode, self.device = mode, torch.device(device)
        self.N, self.T, self.C = int(max_batch), int(T), int(C)
        self.sl = state_len_of(self.C)
        self.args = (int(beam_width), float(beam_cut), float(blank_score), float(scale), float(offset))
        nbytes = (lib.bh_beam_search_workspace(self.N, self.T, self.sl) if mode == "beam"
                  else lib.bh_crf_viterbi_workspace(self.N, self.T, self.sl))
        self.ws = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        self.dev_out = torch.empty((3, self.N, self.T), dtype=torch.int8, device=self.device)
        self.host_out = torch.empty((3, self.N, self.T), dtype=torch.int8).pin_memory()
        self.done = torch.cuda.Event()

    class Ticket:
        def __init__(self, dec, n):
            self.dec, self.n = dec, n

        def result_planes(self):
            """CPU int8 [3, N, T] = (sequence, qstring, moves) stacked (a copy, safe to keep)."""
            self.dec.done.synchronize()
            h = self.dec.host_out[:, : self.n]
            # plain pageable copy, single-threaded on purpose: .clone() of a pinned tensor would hipHostMalloc a new
            # pinned block (~6 ms), and torch's parallel CPU copy wakes the whole intra-op pool for 2.5 MB
            return torch.from_numpy(np.array(h.numpy(), copy=True))

        def result(self):
            """(sequence, qstring, moves) CPU int8 [N, T] (copies, safe to keep)."""
            out = self.result_planes()
            return out[0], out[1], out[2]

    def submit(self, scores):
        _check_scores(scores)
        N, T, Cc = scores.shape
        if N > self.N or T != self.T or Cc != self.C:
            raise ValueError("decoder built for (<=%d, %d, %d), got %s" % (self.N, self.T, self.C, tuple(scores.shape)))
        lib = _lib.lib()
        bw, cut, blank, scale, offset = self.args
        out = self.dev_out
        with torch.cuda.device(self.device):
            st = _lib.stream_ptr(self.device)
            if self.mode == "beam":
                _lib.check(lib.bh_beam_search(_lib.ptr(scores), N, T, self.sl, bw, cut, blank, scale, offset,
                                              _lib.ptr(self.ws), _lib.ptr(out[0]), _lib.ptr(out[1]), _lib.ptr(out[2]),
                                              None, st), "bh_beam_search")
                for pl in range(3):      # one dense copy per plane: a [3, :N] slice is strided when N < max_batch
                    self.host_out[pl, :N].copy_(out[pl, :N], non_blocking=True)
            else:
                # viterbi: plane 2 = moves, plane 1 = path (0..4); sequence/qstring are derived on the host
                _lib.check(lib.bh_crf_viterbi(_lib.ptr(scores), N, T, self.sl, 0, blank, T * Cc, Cc, _lib.ptr(self.ws),
                                              _lib.ptr(out[2]), _lib.ptr(out[1]), None, st), "bh_crf_viterbi")
                for pl in (1, 2):
                    self.host_out[pl, :N].copy_(out[pl, :N], non_blocking=True)
            self.done.record(torch.cuda.current_stream(self.device))
        return CRFDecoder.Ticket(self, N)


def beam_search(scores, beam_width=32, beam_cut=100.0, scale=1.0, offset=0.0, blank_score=2.0, return_qfloat=False):
    """koi.decode.beam_search replacement (same arguments and defaults, bonito/crf/basecall.py:27,36-40).
    scores: cuda fp16 contiguous [N, T, 4^(state_len+1)].  Returns CPU int8 tensors
    (sequence, qstring, moves), each [N, T] with zeros where no base is emitted."""
    _check_scores(scores)
    N, T, Cc = scores.shape
    sl = state_len_of(Cc)
    lib = _lib.lib()
    dev = scores.device
    ws = torch.empty(lib.bh_beam_search_workspace(N, T, sl), dtype=torch.uint8, device=dev)
    out = torch.empty((3, N, T), dtype=torch.int8, device=dev)
    qf = torch.empty((N, T), dtype=torch.float32, device=dev) if return_qfloat else None
    with torch.cuda.device(dev):
        _lib.check(lib.bh_beam_search(_lib.ptr(scores), N, T, sl, int(beam_width), float(beam_cut), float(blank_score),
                                      float(scale), float(offset), _lib.ptr(ws), _lib.ptr(out[0]), _lib.ptr(out[1]),
                                      _lib.ptr(out[2]), _lib.ptr(qf), _lib.stream_ptr(dev)), "bh_beam_search")
    host = out.cpu()          # one D2H copy of the three int8 planes (koi also returns CPU tensors)
    if return_qfloat:
        return host[0], host[1], host[2], qf.cpu()
    return host[0], host[1], host[2]


def viterbi(scores, blank_score=2.0, return_score=False):
    """Max-semiring best path of the CTC-CRF (CTC_CRF.viterbi, bonito/crf/model.py:98-103) on koi-layout
    scores fp16 [N, T, 4^(state_len+1)].  Returns CPU int8 tensors (moves [N,T] in {0,1},
    path [N,T] in {0..4}), like koi's decoders return CPU int8."""
    _check_scores(scores)
    N, T, Cc = scores.shape
    sl = state_len_of(Cc)
    lib = _lib.lib()
    dev = scores.device
    ws = torch.empty(lib.bh_crf_viterbi_workspace(N, T, sl), dtype=torch.uint8, device=dev)
    moves = torch.empty((N, T), dtype=torch.int8, device=dev)
    path = torch.empty((N, T), dtype=torch.int8, device=dev)
    best = torch.empty((N,), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        _lib.check(lib.bh_crf_viterbi(_lib.ptr(scores), N, T, sl, 0, float(blank_score), T * Cc, Cc, _lib.ptr(ws),
                                      _lib.ptr(moves), _lib.ptr(path), _lib.ptr(best), _lib.stream_ptr(dev)),
                   "bh_crf_viterbi")
    if return_score:
        return moves.cpu(), path.cpu(), best.cpu()
    return moves.cpu(), path.cpu()


def viterbi_5s(scores_tnc, state_len, return_score=False):
    """Same, on the reference's expand_blanks layout: fp16 [T, N, 5*4^state_len] (crf/model.py:49)."""
    if scores_tnc.dtype != torch.float16 or not scores_tnc.is_cuda:
        raise TypeError("expected cuda fp16 scores")
    scores_tnc = scores_tnc.contiguous()
    T, N, Cc = scores_tnc.shape
    lib = _lib.lib()
    dev = scores_tnc.device
    ws = torch.empty(lib.bh_crf_viterbi_workspace(N, T, state_len), dtype=torch.uint8, device=dev)
    moves = torch.empty((N, T), dtype=torch.int8, device=dev)
    path = torch.empty((N, T), dtype=torch.int8, device=dev)
    best = torch.empty((N,), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        _lib.check(lib.bh_crf_viterbi(_lib.ptr(scores_tnc), N, T, state_len, 1, 0.0, Cc, N * Cc, _lib.ptr(ws),
                                      _lib.ptr(moves), _lib.ptr(path), _lib.ptr(best), _lib.stream_ptr(dev)),
                   "bh_crf_viterbi")
    if return_score:
        return moves.cpu(), path.cpu(), best.cpu()
    return moves.cpu(), path.cpu()


def path_to_sequence(path):
    """int8 path in {0..4} -> int8 ASCII bytes (0 where nothing is emitted): koi's `sequence` layout,
    so ``to_str`` and ``stitch`` (crf/basecall.py:13-24,48-55) work on it unchanged."""
    p = path.numpy() if isinstance(path, torch.Tensor) else np.asarray(path)
    out = np.where(p != 0, _ALPHABET[p.astype(np.int64)], 0).astype(np.int8)
    return torch.from_numpy(out)


def to_str(x, encoding="ascii"):
    """Non-zero bytes decoded as text (koi.decode.to_str)."""
    a = x.numpy() if isinstance(x, torch.Tensor) else np.asarray(x)
    a = a[a != 0]
    return a.astype(np.uint8).tobytes().decode(encoding)


def reverse_complement(scores):
    """CTC_CRF.reverse_complement (bonito/crf/model.py:84-96) on koi-layout scores cuda fp16 [N, T, 4S]."""
    _check_scores(scores)
    N, T, Cc = scores.shape
    out = torch.empty_like(scores)
    with torch.cuda.device(scores.device):
        _lib.check(_lib.lib().bh_crf_reverse_complement(_lib.ptr(scores), _lib.ptr(out), N, T, state_len_of(Cc), 0,
                                                        T * Cc, Cc, _lib.stream_ptr(scores.device)), "bh_crf_reverse_complement")
    return out


def reverse_complement_5s(scores_tnc, state_len):
    """Same on the reference layout [T, N, 5S] (cuda fp16)."""
    scores_tnc = scores_tnc.contiguous()
    T, N, Cc = scores_tnc.shape
    out = torch.empty_like(scores_tnc)
    with torch.cuda.device(scores_tnc.device):
        _lib.check(_lib.lib().bh_crf_reverse_complement(_lib.ptr(scores_tnc), _lib.ptr(out), N, T, int(state_len), 1,
                                                        Cc, N * Cc, _lib.stream_ptr(scores_tnc.device)), "bh_crf_reverse_complement")
    return out


def logz(scores, blank_score=2.0):
    """Log-partition function per chunk (CTC_CRF.logZ, crf/model.py:47-52) of koi-layout scores -> CPU float64 [N]."""
    _check_scores(scores)
    N, T, Cc = scores.shape
    sl = state_len_of(Cc)
    lib = _lib.lib()
    ws = torch.empty(lib.bh_beam_search_workspace(N, T, sl), dtype=torch.uint8, device=scores.device)
    out = torch.empty(N, dtype=torch.float64, device=scores.device)
    with torch.cuda.device(scores.device):
        _lib.check(lib.bh_crf_logz(_lib.ptr(scores), N, T, sl, float(blank_score), _lib.ptr(ws), _lib.ptr(out),
                                   _lib.stream_ptr(scores.device)), "bh_crf_logz")
    return out.cpu()


def posterior_viterbi(scores, blank_score=2.0):
    """SeqdistModel.decode_batch's decoder (crf/model.py:196-199): best path over log edge posteriors.
    scores cuda fp16 [N, T, 4S] -> CPU int8 (moves, path)."""
    _check_scores(scores)
    N, T, Cc = scores.shape
    sl = state_len_of(Cc)
    lib = _lib.lib()
    dev = scores.device
    ws = torch.empty(lib.bh_crf_posterior_viterbi_workspace(N, T, sl), dtype=torch.uint8, device=dev)
    moves = torch.empty((N, T), dtype=torch.int8, device=dev)
    path = torch.empty((N, T), dtype=torch.int8, device=dev)
    with torch.cuda.device(dev):
        _lib.check(lib.bh_crf_posterior_viterbi(_lib.ptr(scores), N, T, sl, float(blank_score), _lib.ptr(ws),
                                                _lib.ptr(moves), _lib.ptr(path), _lib.stream_ptr(dev)), "bh_crf_posterior_viterbi")
    return moves.cpu(), path.cpu()


def set_option(name, value):
    """Process-wide engine knob (bh_set_option), e.g. set_option("beam_fork", 0)."""
    _lib.check(_lib.lib().bh_set_option(name.encode(), int(value)), "bh_set_option")
