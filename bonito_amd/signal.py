"""
Signal ingest on the device for raw int16 reads: pA scaling, quantile normalisation, adapter trim and chunking into the
encoder's fp16 input batches (libbonito_hip.so: bh_signal_normalise / bh_signal_chunks, csrc/signal.hip).

Device counterpart of `bonito_amd.reader.Read.__init__` (itself a mirror of /root/reference bonito/reader.py:122-166 and
bonito/pod5.py:52-67) followed by `util.chunk` + `batchify` + the fp16 cast: same shift / scale / trim per read and the
same fp16 chunk rows, bit for bit (tests/test_gpu_signal.py), with 2 bytes per sample crossing PCIe once.
"""
import ctypes as C

import numpy as np
import torch

from bonito_amd import _lib
from bonito_amd.reader import __default_norm_params__


def chunk_table(lengths, trims, chunksize, overlap):
    """Chunk origins of `util.chunk` (bonito/util.py:142-161) for reads of `lengths` samples whose first `trims` samples are
    cut: arrays (read index, first sample, samples available). A read shorter than the chunk yields one row with
    available < chunksize (the kernel tiles it); otherwise windows advance by chunksize - overlap from offset `stub`, with
    an extra first chunk at 0 when stub > 0. Reads with nothing left after the trim yield no rows."""
    reads, starts, avail = [], [], []
    for r, (n, t0) in enumerate(zip(lengths, trims)):
        t0 = int(t0)
        T = int(n) - t0
        if T <= 0:
            continue
        if T < chunksize:
            reads.append(r); starts.append(t0); avail.append(T)
            continue
        step = chunksize - overlap
        stub = (T - overlap) % step
        for s in ([t0] if stub > 0 else []) + list(range(t0 + stub, t0 + T - chunksize + 1, step)):
            reads.append(r); starts.append(s); avail.append(chunksize)
    return (np.asarray(reads, dtype=np.int32), np.asarray(starts, dtype=np.int64), np.asarray(avail, dtype=np.int64))


class RawBatch:
    """Raw reads resident on the device: concatenated int16 samples + per-read calibration."""

    def __init__(self, raws, scalings, offsets, device="cuda"):
        assert len(raws) == len(scalings) == len(offsets) and len(raws) > 0
        self.device = torch.device(device)
        lens = np.array([len(r) for r in raws], dtype=np.int64)
        self.lengths = lens
        offs = np.zeros(len(raws) + 1, dtype=np.int64)
        np.cumsum(lens, out=offs[1:])
        host = torch.from_numpy(np.concatenate([np.asarray(r, dtype=np.int16) for r in raws]))
        self.raw = host.to(self.device)
        self.offs = torch.from_numpy(offs).to(self.device)
        self.cal_scale = torch.tensor(scalings, dtype=torch.float32, device=self.device)
        self.cal_offset = torch.tensor(offsets, dtype=torch.float32, device=self.device)
        self.R = len(raws)
        self.shift = self.scale = self.weak = self.trim = None

    def normalise(self, scaling_strategy=None, norm_params=None, do_trim=True):
        """Per-read (shift, scale, trim) on the device; returns them as numpy arrays (fp64, fp64, int32)."""
        strategy, qa, qb, sm, cm, fs, fc = 0, 0.0, 0.0, 0.0, 0.0, 0.0, 1.0
        if scaling_strategy and scaling_strategy.get("strategy") == "pa":
            strategy = 1
            if norm_params and norm_params.get("standardise") == 1:
                fs, fc = float(norm_params.get("mean")), float(norm_params.get("stdev"))
            elif norm_params and norm_params.get("standardise") == 0:
                fs, fc = 0.0, 1.0
            else:
                raise ValueError("Picoampere scaling requested, but standardisation flag not provided")
        elif scaling_strategy is None or scaling_strategy.get("strategy") == "quantile":
            prm = norm_params or __default_norm_params__
            qa, qb, sm, cm = (float(prm[k]) for k in ("quantile_a", "quantile_b", "shift_multiplier", "scale_multiplier"))
        else:
            raise ValueError("Scaling strategy %s not supported; choose quantile or pa." % scaling_strategy.get("strategy"))
        dev = self.device
        self.shift = torch.empty(self.R, dtype=torch.float64, device=dev)
        self.scale = torch.empty(self.R, dtype=torch.float64, device=dev)
        self.weak = torch.empty(self.R, dtype=torch.int32, device=dev)
        self.trim = torch.empty(self.R, dtype=torch.int32, device=dev)
        with torch.cuda.device(dev):
            _lib.check(_lib.lib().bh_signal_normalise(
                _lib.ptr(self.raw), _lib.ptr(self.offs), _lib.ptr(self.cal_scale), _lib.ptr(self.cal_offset), self.R, strategy,
                qa, qb, sm, cm, fs, fc, int(bool(do_trim)), _lib.ptr(self.shift), _lib.ptr(self.scale), _lib.ptr(self.weak),
                _lib.ptr(self.trim), _lib.stream_ptr(dev)), "bh_signal_normalise")
        return self.shift.cpu().numpy(), self.scale.cpu().numpy(), self.trim.cpu().numpy()

    def chunk_table(self, chunksize, overlap, trims=None):
        """(read, start, available) per chunk in util.chunk order (stub chunk first, short reads tiled), host side."""
        trims = self.trim.cpu().numpy() if trims is None else np.asarray(trims)
        return chunk_table(self.lengths, trims, chunksize, overlap)

    def chunks(self, table, chunksize, lo=0, hi=None):
        """fp16 [n, 1, chunksize] device tensor of table rows lo:hi (call `normalise` first)."""
        reads, starts, avail = (t[lo:hi] for t in table)
        n = len(reads)
        dev = self.device
        out = torch.empty((n, 1, chunksize), dtype=torch.float16, device=dev)
        if n == 0:
            return out
        d_reads = torch.from_numpy(np.ascontiguousarray(reads)).to(dev)
        d_starts = torch.from_numpy(np.ascontiguousarray(starts)).to(dev)
        d_avail = torch.from_numpy(np.ascontiguousarray(avail)).to(dev)
        with torch.cuda.device(dev):
            _lib.check(_lib.lib().bh_signal_chunks(
                _lib.ptr(self.raw), _lib.ptr(self.offs), _lib.ptr(self.cal_scale), _lib.ptr(self.cal_offset), _lib.ptr(self.shift),
                _lib.ptr(self.scale), _lib.ptr(self.weak), _lib.ptr(d_reads), _lib.ptr(d_starts), _lib.ptr(d_avail), n,
                int(chunksize), _lib.ptr(out), _lib.stream_ptr(dev)), "bh_signal_chunks")
        return out
