"""
CTC decoders with the call surface of the ``fast_ctc_decode`` crate used at /root/reference
bonito/ctc/model.py:11,39-46, running on the HIP device (bonito_amd/csrc/ctc.hip).
"""
import ctypes as C

import numpy as np
import torch

from bonito_amd import _lib


def viterbi_search_batch(logps, alphabet, qscale=1.0, qbias=0.0, device="cuda"):
    """Greedy decode of a list of [T_r, n_labels] log-probability tensors in ONE launch.
    Returns a list of (sequence, qstring, path) per read."""
    if not logps:
        return []
    dev = torch.device(device)
    lens = [int(x.shape[0]) for x in logps]
    offs = torch.tensor(np.concatenate([[0], np.cumsum(lens)]), dtype=torch.int64)
    flat = torch.cat([torch.as_tensor(x, dtype=torch.float32) for x in logps]).contiguous().to(dev)
    total, Cc = flat.shape
    offs_d = offs.to(dev)
    lab = torch.empty(total, dtype=torch.int8, device=dev)
    qual = torch.empty(total, dtype=torch.int8, device=dev)
    path = torch.empty(total, dtype=torch.int32, device=dev)
    cnt = torch.empty(len(lens), dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        _lib.check(_lib.lib().bh_ctc_greedy_decode(_lib.ptr(flat), _lib.ptr(offs_d), len(lens), Cc, float(qscale),
                                                   float(qbias), _lib.ptr(lab), _lib.ptr(qual), _lib.ptr(path),
                                                   _lib.ptr(cnt), _lib.stream_ptr(dev)), "bh_ctc_greedy_decode")
    lab, qual, path, cnt = lab.cpu().numpy(), qual.cpu().numpy(), path.cpu().numpy(), cnt.cpu().numpy()
    letters = np.frombuffer("".join(alphabet).encode(), dtype=np.uint8)
    out = []
    for r, n in enumerate(cnt):
        o = int(offs[r])
        seq = letters[lab[o:o + n].astype(np.int64)].tobytes().decode()
        qs = qual[o:o + n].astype(np.uint8).tobytes().decode()
        out.append((seq, qs, path[o:o + n].astype(np.int64).tolist()))
    return out


def viterbi_search(logp, alphabet, qstring=False, qscale=1.0, qbias=0.0):
    """(sequence [+ qstring], path) of one read, like fast_ctc_decode.viterbi_search -- except that the
    input is the LOG-probability tensor (the reference exponentiates on the CPU first, ctc/model.py:40)."""
    (seq, qs, path), = viterbi_search_batch([logp], alphabet, qscale, qbias)
    return (seq + qs if qstring else seq), path


def beam_search_batch(logps, alphabet, beam_size=5, beam_cut_threshold=1e-3, device="cuda"):
    """Prefix beam search (PB-1) of a list of [T_r, n_labels] log-probability tensors in ONE launch
    (one lane per read). Returns a list of (sequence, path)."""
    if not logps:
        return []
    dev = torch.device(device)
    lens = [int(x.shape[0]) for x in logps]
    offs = torch.tensor(np.concatenate([[0], np.cumsum(lens)]), dtype=torch.int64)
    flat = torch.cat([torch.as_tensor(x, dtype=torch.float32) for x in logps]).contiguous().to(dev)
    total, Cc = flat.shape
    lib = _lib.lib()
    offs_d = offs.to(dev)
    ws = torch.empty(lib.bh_ctc_beam_search_workspace(total, len(lens), Cc, int(beam_size)), dtype=torch.uint8, device=dev)
    lab = torch.empty(max(total, 1), dtype=torch.int8, device=dev)
    path = torch.empty(max(total, 1), dtype=torch.int32, device=dev)
    cnt = torch.empty(len(lens), dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        _lib.check(lib.bh_ctc_beam_search(_lib.ptr(flat), _lib.ptr(offs_d), len(lens), Cc, int(beam_size),
                                          float(beam_cut_threshold), _lib.ptr(ws), _lib.ptr(lab), _lib.ptr(path),
                                          _lib.ptr(cnt), _lib.stream_ptr(dev)), "bh_ctc_beam_search")
    lab, path, cnt = lab.cpu().numpy(), path.cpu().numpy(), cnt.cpu().numpy()
    letters = np.frombuffer("".join(alphabet).encode(), dtype=np.uint8)
    out = []
    for r, n in enumerate(cnt):
        o = int(offs[r])
        out.append((letters[lab[o:o + n].astype(np.int64)].tobytes().decode(), path[o:o + n].astype(np.int64).tolist()))
    return out


def beam_search(logp, alphabet, beam_size=5, beam_cut_threshold=1e-3):
    """(sequence, path) of one read, like fast_ctc_decode.beam_search (input: LOG-probabilities)."""
    (seq, path), = beam_search_batch([logp], alphabet, beam_size, beam_cut_threshold)
    return seq, path
