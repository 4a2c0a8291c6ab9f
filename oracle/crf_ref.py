"""
CPU ORACLE -- TEST INFRASTRUCTURE ONLY.  Python face of oracle/crf_oracle.c plus two independent
restatements used to pin it:

* ``viterbi_autograd``  -- literal torch restatement of the reference's formulation: the Max-semiring
  "posteriors" are d logZ / d Ms (one-hot along the best path), then ``a = argmax``; ``moves = a % 5 != 0``;
  ``paths = 1 + (a // 5) % 4`` (/root/reference bonito/crf/model.py:47-52,98-103; koi's
  ``SequenceDist.posteriors`` is the autograd of logZ).
* ``viterbi_bruteforce`` -- exhaustive enumeration over all (start state, transition) sequences for tiny T/S.

PARITY UNPINNED by reference tests (none exist for this path); pinned by agreement of these three.
"""
import ctypes as C
import itertools
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def _lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            raise RuntimeError("oracle/liboracle.so missing: run `python build.py`")
        _LIB = C.CDLL(path)
    return _LIB


def _as_half_bits(scores):
    a = np.ascontiguousarray(np.asarray(scores, dtype=np.float16))
    return a, a.view(np.uint16)


def viterbi(scores, state_len, layout_5s=False, blank=2.0, time_major=False):
    """scores: float16 array [N,T,C] (or [T,N,C] if time_major). Returns (moves, path, best)."""
    a, bits = _as_half_bits(scores)
    if time_major:
        T, N, Cc = a.shape
        s_n, s_t = Cc, N * Cc
    else:
        N, T, Cc = a.shape
        s_n, s_t = T * Cc, Cc
    moves = np.zeros((N, T), np.int8)
    path = np.zeros((N, T), np.int8)
    best = np.zeros((N,), np.float32)
    rc = _lib().oracle_crf_viterbi(
        bits.ctypes.data_as(C.c_void_p), N, T, int(state_len), int(layout_5s), C.c_float(blank),
        C.c_long(s_n), C.c_long(s_t), moves.ctypes.data_as(C.c_void_p), path.ctypes.data_as(C.c_void_p),
        best.ctypes.data_as(C.c_void_p))
    if rc:
        raise RuntimeError("oracle_crf_viterbi failed")
    return moves, path, best


def logz(scores, state_len, layout_5s=False, blank=2.0, time_major=False):
    a, bits = _as_half_bits(scores)
    if time_major:
        T, N, Cc = a.shape
        s_n, s_t = Cc, N * Cc
    else:
        N, T, Cc = a.shape
        s_n, s_t = T * Cc, Cc
    out = np.zeros((N,), np.float32)
    rc = _lib().oracle_crf_logz(bits.ctypes.data_as(C.c_void_p), N, T, int(state_len), int(layout_5s),
                                C.c_float(blank), C.c_long(s_n), C.c_long(s_t), out.ctypes.data_as(C.c_void_p))
    if rc:
        raise RuntimeError("oracle_crf_logz failed")
    return out


def expand_blanks(scores_4s, blank):
    """koi layout [..., 4S] -> reference layout [..., 5S] (bonito/nn.py:291-297)."""
    a = np.asarray(scores_4s)
    x = a.reshape(*a.shape[:-1], -1, 4)
    pad = np.full((*x.shape[:-1], 1), blank, dtype=a.dtype)
    return np.concatenate([pad, x], axis=-1).reshape(*a.shape[:-1], -1)


def idx_table(state_len, n_base=4):
    S = n_base ** state_len
    j = np.arange(S)
    cols = [j] + [r * (S // n_base) + j // n_base for r in range(n_base)]
    return np.stack(cols, axis=1).astype(np.int64)


def viterbi_autograd(scores_tnc_5s, state_len):
    """Reference formulation with torch autograd (fp64 to keep the argmax stable); returns path [T,N]."""
    import torch
    x = torch.as_tensor(np.asarray(scores_tnc_5s, dtype=np.float64)).requires_grad_(True)
    T, N, _ = x.shape
    S = 4 ** state_len
    idx = torch.as_tensor(idx_table(state_len))
    Ms = x.reshape(T, N, S, 5)
    alpha = torch.zeros(N, S, dtype=torch.float64)
    for t in range(T):
        alpha = (Ms[t] + alpha[:, idx]).max(dim=-1).values
    alpha.max(dim=-1).values.sum().backward()
    tb = x.grad.reshape(T, N, -1)
    a = tb.argmax(2)
    moves = (a % 5) != 0
    paths = 1 + (a // 5) % 4
    return torch.where(moves, paths, torch.zeros_like(paths)).numpy()


def viterbi_bruteforce(scores_tnc_5s, state_len):
    """Exhaustive search; returns (best score [N], path [T,N]) for tiny problems."""
    x = np.asarray(scores_tnc_5s, dtype=np.float64)
    T, N, _ = x.shape
    S = 4 ** state_len
    idx = idx_table(state_len)
    Ms = x.reshape(T, N, S, 5)
    best = np.full(N, -np.inf)
    paths = np.zeros((T, N), np.int64)
    # a path = final state + the transition index k_t at every step, walked backwards
    for n in range(N):
        for final in range(S):
            for ks in itertools.product(range(5), repeat=T):
                st, sc = final, 0.0
                for t in range(T - 1, -1, -1):
                    sc += Ms[t, n, st, ks[t]]
                    st = idx[st, ks[t]]
                if sc > best[n]:
                    best[n] = sc
                    st = final
                    for t in range(T - 1, -1, -1):
                        paths[t, n] = 0 if ks[t] == 0 else 1 + (st % 4)
                        st = idx[st, ks[t]]
    return best, paths


def map_sequence_bruteforce(scores_t4s, state_len, blank=2.0):
    """EXACT sequence posterior of a tiny problem: {sequence: ln sum of exp(path score) over every path spelling it} for koi-layout
    scores [T, 4S] (fp64 arithmetic, unpruned prefix search: every (sequence, state) pair is kept). The target a beam search
    approximates - independent of BS-1 / BS-2 / the guide; used by tests only (cost ~ 5^T)."""
    x = np.asarray(scores_t4s, dtype=np.float64)
    T, S = x.shape[0], 4 ** state_len
    cur = {("", s): 0.0 for s in range(S)}
    for t in range(T):
        new = {}
        for (seq, s), lp in cur.items():
            cands = [((seq, s), lp + blank)]
            for b in range(4):
                ns = ((s << 2) | b) & (S - 1)
                cands.append(((seq + "ACGT"[b], ns), lp + x[t, ns * 4 + s // (S // 4)]))
            for k, v in cands:
                new[k] = np.logaddexp(new[k], v) if k in new else v
        cur = new
    total = {}
    for (seq, _), lp in cur.items():
        total[seq] = np.logaddexp(total[seq], lp) if seq in total else lp
    return total


def beam_search(scores, state_len, beam_width=32, beam_cut=100.0, blank=2.0, scale=1.0, offset=0.0):
    """BS-2 decode (linear-domain guide, fp64 posteriors; BS-1 = the table-lse2 guide of rounds 1-4) of koi-layout scores [N,T,4S] (float16). Returns (sequence, qstring, moves, qfloat)."""
    a, bits = _as_half_bits(scores)
    N, T, _ = a.shape
    seq = np.zeros((N, T), np.int8)
    qs = np.zeros((N, T), np.int8)
    mv = np.zeros((N, T), np.int8)
    qf = np.zeros((N, T), np.float32)
    rc = _lib().oracle_beam_search(
        bits.ctypes.data_as(C.c_void_p), N, T, int(state_len), int(beam_width), C.c_float(beam_cut),
        C.c_float(blank), C.c_float(scale), C.c_float(offset), seq.ctypes.data_as(C.c_void_p),
        qs.ctypes.data_as(C.c_void_p), mv.ctypes.data_as(C.c_void_p), qf.ctypes.data_as(C.c_void_p))
    if rc:
        raise RuntimeError("oracle_beam_search failed (%d)" % rc)
    return seq, qs, mv, qf


def beam_search_bs1(scores, state_len, beam_width=32, beam_cut=100.0, blank=2.0):
    """BS-1 (the decoder of rounds 1-4: Log-semiring table-lse2 guide, candidate order 5 e + j), kept ONLY as the quality reference the
    product decoder BS-2 is guarded against. Returns (sequence, moves) [N,T] int8."""
    a, bits = _as_half_bits(scores)
    N, T, _ = a.shape
    seq = np.zeros((N, T), np.int8)
    mv = np.zeros((N, T), np.int8)
    rc = _lib().oracle_beam_search_bs1(bits.ctypes.data_as(C.c_void_p), N, T, int(state_len), int(beam_width), C.c_float(beam_cut),
                                       C.c_float(blank), seq.ctypes.data_as(C.c_void_p), mv.ctypes.data_as(C.c_void_p))
    if rc:
        raise RuntimeError("oracle_beam_search_bs1 failed (%d)" % rc)
    return seq, mv


def seq_logprob(scores_t4s, state_len, seq, blank=2.0):
    """(ln sum over every alignment of `seq` of exp(path score), logZ) of ONE chunk's koi-layout scores [T,4S] in fp64 - the model's exact
    sequence likelihood (crf/model.py:30-108 semantics), independent of any decoder. `seq`: bytes / str over ACGT, or an int8 plane of a
    decoder (zeros = nothing emitted). ln P(seq | scores) = the difference of the two."""
    a, bits = _as_half_bits(scores_t4s)
    T = a.shape[0]
    if isinstance(seq, str):
        seq = seq.encode()
    raw = np.frombuffer(seq, np.uint8) if isinstance(seq, (bytes, bytearray)) else np.asarray(seq).astype(np.uint8)
    raw = raw[raw != 0]
    lut = np.full(256, -1, np.int8)
    for i, ch in enumerate(b"ACGT"):
        lut[ch] = i
    idx = np.ascontiguousarray(lut[raw])
    if (idx < 0).any():
        raise ValueError("sequence holds bytes outside ACGT")
    lp, lz = C.c_double(), C.c_double()
    rc = _lib().oracle_seq_logprob_f64(bits.ctypes.data_as(C.c_void_p), int(T), int(state_len), C.c_float(blank),
                                       idx.ctypes.data_as(C.c_void_p), int(len(idx)), C.byref(lp), C.byref(lz))
    if rc:
        raise RuntimeError("oracle_seq_logprob_f64 failed")
    return lp.value, lz.value


def backward(scores, state_len, blank=2.0):
    """-> (beta~ [N,T+1,S] f32, Bcum [N,T+1] f64, logZ [N] f64)"""
    a, bits = _as_half_bits(scores)
    N, T, _ = a.shape
    S = 4 ** state_len
    beta = np.zeros((N, T + 1, S), np.float32)
    B = np.zeros((N, T + 1), np.float64)
    lz = np.zeros((N,), np.float64)
    rc = _lib().oracle_crf_backward(bits.ctypes.data_as(C.c_void_p), N, T, int(state_len), C.c_float(blank),
                                    beta.ctypes.data_as(C.c_void_p), B.ctypes.data_as(C.c_void_p),
                                    lz.ctypes.data_as(C.c_void_p))
    if rc:
        raise RuntimeError("oracle_crf_backward failed")
    return beta, B, lz


def forward_post(scores, state_len, beta, B, lz, blank=2.0):
    a, bits = _as_half_bits(scores)
    N, T, _ = a.shape
    P = np.zeros((N, T, 4), np.float32)
    rc = _lib().oracle_crf_forward_post(bits.ctypes.data_as(C.c_void_p), N, T, int(state_len), C.c_float(blank),
                                        beta.ctypes.data_as(C.c_void_p), B.ctypes.data_as(C.c_void_p),
                                        lz.ctypes.data_as(C.c_void_p), P.ctypes.data_as(C.c_void_p))
    if rc:
        raise RuntimeError("oracle_crf_forward_post failed")
    return P


def lse2(a, b):
    f = _lib().oracle_lse2
    f.restype = C.c_float
    return f(C.c_float(a), C.c_float(b))


def _rc_digits(j, k):
    r = 0
    for _ in range(k):
        r = (r << 2) | (3 - (j & 3))
        j >>= 2
    return r


def reverse_complement(scores, state_len, layout_5s=True):
    """Index-level restatement of CTC_CRF.reverse_complement (bonito/crf/model.py:84-96).
    layout_5s: [T,N,5S]; else koi layout [N,T,4S]. Pinned against tests/golden/crf_rc.npz (reference output)."""
    a = np.asarray(scores)
    S = 4 ** state_len
    out = np.empty_like(a)
    if layout_5s:
        T = a.shape[0]
        for j2 in range(S):
            out[:, :, j2 * 5] = a[::-1, :, _rc_digits(j2, state_len) * 5]
            for r2 in range(4):
                src = _rc_digits(r2 * S + j2, state_len + 1)
                r, j = divmod(src, S)
                out[:, :, j2 * 5 + 1 + r2] = a[::-1, :, j * 5 + 1 + r]
    else:
        for j2 in range(S):
            for r2 in range(4):
                src = _rc_digits(r2 * S + j2, state_len + 1)
                r, j = divmod(src, S)
                out[:, :, j2 * 4 + r2] = a[:, ::-1, j * 4 + r]
    return out


def posterior_viterbi(scores, state_len, blank=2.0):
    """decode_batch's decoder on koi-layout scores [N,T,4S]: (moves, path)."""
    a, bits = _as_half_bits(scores)
    N, T, _ = a.shape
    moves = np.zeros((N, T), np.int8)
    path = np.zeros((N, T), np.int8)
    rc = _lib().oracle_crf_posterior_viterbi(bits.ctypes.data_as(C.c_void_p), N, T, int(state_len), C.c_float(blank),
                                             moves.ctypes.data_as(C.c_void_p), path.ctypes.data_as(C.c_void_p))
    if rc:
        raise RuntimeError("oracle_crf_posterior_viterbi failed")
    return moves, path


def posterior_viterbi_autograd(scores_4s, state_len, blank=2.0):
    """Literal restatement of crf/model.py:196-199 with torch autograd posteriors (fp64): path [N,T]."""
    import torch
    x5 = expand_blanks(np.asarray(scores_4s, dtype=np.float64), blank).transpose(1, 0, 2).copy()      # [T,N,5S]
    x = torch.as_tensor(x5).requires_grad_(True)
    T, N, _ = x.shape
    S = 4 ** state_len
    idx = torch.as_tensor(idx_table(state_len))
    Ms = x.reshape(T, N, S, 5)
    alpha = torch.zeros(N, S, dtype=torch.float64)
    for t in range(T):
        alpha = torch.logsumexp(Ms[t] + alpha[:, idx], dim=-1)
    torch.logsumexp(alpha, dim=-1).sum().backward()
    post = x.grad + 1e-8
    return viterbi_autograd(post.log().numpy(), state_len).T


def bs2_exp(x):
    f = _lib().oracle_bs2_exp
    f.restype, f.argtypes = C.c_float, [C.c_float]
    return float(f(float(x)))


def bs2_log(v):
    f = _lib().oracle_bs2_log
    f.restype, f.argtypes = C.c_float, [C.c_float]
    return float(f(float(v)))


def bs2_backward(scores, state_len, blank=2.0):
    """-> b [N,T+1,S] f32: the linear-domain guide of BS-2 (every row scaled by a power of two so that its maximum lies in [1, 2))."""
    a, bits = _as_half_bits(scores)
    N, T, _ = a.shape
    S = 4 ** state_len
    b = np.zeros((N, T + 1, S), np.float32)
    rc = _lib().oracle_bs2_backward(bits.ctypes.data_as(C.c_void_p), N, T, int(state_len), C.c_float(blank), b.ctypes.data_as(C.c_void_p))
    if rc:
        raise RuntimeError("oracle_bs2_backward failed")
    return b


def posteriors_f64(scores, state_len, blank=2.0):
    """-> P [N,T,4] f32: the model's true class posteriors at the boundaries u = 1..T (fp64 forward / backward, libm)."""
    a, bits = _as_half_bits(scores)
    N, T, _ = a.shape
    P = np.zeros((N, T, 4), np.float32)
    rc = _lib().oracle_crf_posteriors_f64(bits.ctypes.data_as(C.c_void_p), N, T, int(state_len), C.c_float(blank), P.ctypes.data_as(C.c_void_p))
    if rc:
        raise RuntimeError("oracle_crf_posteriors_f64 failed")
    return P
