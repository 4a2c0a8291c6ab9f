"""
CPU ORACLE -- TEST INFRASTRUCTURE ONLY.  Greedy CTC decode as bonito.ctc uses it: the reference calls the
Rust crate ``fast_ctc_decode.viterbi_search(probs, alphabet, qstring, qscale, qbias)``
(/root/reference bonito/ctc/model.py:39-42; crate un-pinned in requirements.txt:3 and absent here), whose
published behaviour is restated: per-step argmax, collapse repeats, drop blank (label 0); `path` = step
index of each emitted base; per-base quality through bonito's own ``phred`` formula (bonito/util.py:105-111)
of the label's MEAN probability over its run ([EXT]: mean-vs-max is not verifiable offline -> PARITY UNPINNED).
"""
import numpy as np


def phred(prob, scale=1.0, bias=0.0):
    p = max(1.0 - float(prob), 1e-4)
    q = -10.0 * np.log10(np.float32(p)) * scale + bias
    return int(np.rint(np.float32(q))) + 33


def viterbi_search(logp, alphabet, qscale=1.0, qbias=0.0):
    """logp: [T, C] float array of log-probabilities -> (sequence, qstring, path)."""
    lp = np.asarray(logp, dtype=np.float32)
    T = lp.shape[0]
    labels = lp.argmax(axis=1)          # ties: lowest label (numpy argmax)
    seq, qs, path = [], [], []
    t = 0
    prev = 0
    while t < T:
        lab = int(labels[t])
        if lab != 0 and lab != prev:
            u = t
            while u < T and labels[u] == lab:
                u += 1
            prob = float(np.exp(lp[t:u, lab].astype(np.float64)).mean())
            seq.append(alphabet[lab])
            qs.append(chr(phred(prob, qscale, qbias)))
            path.append(t)
        prev = lab
        t += 1
    return "".join(seq), "".join(qs), path


def beam_search(logp, alphabet, beam_size=5, beam_cut_threshold=1e-3):
    """PB-1 prefix beam search (oracle/crf_oracle.c::oracle_ctc_prefix_beam) -> (sequence, path)."""
    import ctypes as C
    from oracle.crf_ref import _lib
    lp = np.ascontiguousarray(np.asarray(logp, dtype=np.float32))
    T, Cc = lp.shape
    labels = np.zeros(max(T, 1), np.int8)
    path = np.zeros(max(T, 1), np.int32)
    cnt = C.c_int(0)
    rc = _lib().oracle_ctc_prefix_beam(lp.ctypes.data_as(C.c_void_p), T, Cc, int(beam_size), C.c_float(beam_cut_threshold),
                                       labels.ctypes.data_as(C.c_void_p), path.ctypes.data_as(C.c_void_p), C.byref(cnt))
    if rc:
        raise RuntimeError("oracle_ctc_prefix_beam failed (%d)" % rc)
    n = cnt.value
    return "".join(alphabet[i] for i in labels[:n]), path[:n].tolist()
