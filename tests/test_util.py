"""Host logic (bonito_amd/util.py) vs fixtures produced by the reference's bonito/util.py (CPU only)."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from bonito_amd import util

with open(os.path.join(GOLDEN, "util_cases.json")) as fh:
    CASES = json.load(fh)


@pytest.mark.parametrize("case", CASES["chunk_stitch"], ids=lambda c: "T%d_c%d_o%d" % (c["T"], c["chunksize"], c["overlap"]))
def test_chunk_and_stitch(case):
    T, cs, ov, stride = case["T"], case["chunksize"], case["overlap"], case["stride"]
    sig = torch.arange(T, dtype=torch.float32)
    ch = util.chunk(sig, cs, ov)
    assert list(ch.shape) == case["chunk_shape"]
    assert ch[:, 0, 0].tolist() == case["chunk_first"]
    assert ch[:, 0, -1].tolist() == case["chunk_last"]
    if not cs:
        return
    n, _, L = ch.shape
    steps = L // stride
    per = torch.stack([ch[i, 0, ::stride][:steps] for i in range(n)]).to(torch.int64)
    if T < cs:
        st = per[0, : int(np.floor(T / stride))]
    else:
        st = util.stitch(per, cs, ov, T, stride)
    assert st.tolist() == case["stitched"]
    if "stitched_rev" in case:
        assert util.stitch(per, cs, ov, T, stride, reverse=True).tolist() == case["stitched_rev"]


def test_stitch_covers_read_once():
    """size-independent property: stitched steps are strictly increasing sample offsets covering the read."""
    T, cs, ov, stride = 50000, 9996, 492, 6
    sig = torch.arange(T, dtype=torch.float32)
    ch = util.chunk(sig, cs, ov)
    per = torch.stack([c[0, ::stride][: cs // stride] for c in ch]).to(torch.int64)
    st = util.stitch(per, cs, ov, T, stride)
    d = np.diff(st.numpy())
    assert (d > 0).all() and d.max() <= 2 * stride
    assert st[0] == 0 and st[-1] >= T - stride - 1


def test_batchify_unbatchify_golden():
    g = CASES["batchify"]
    items = [("r%d" % i, torch.arange(n * 3, dtype=torch.float32).reshape(n, 1, 3) + 100 * i)
             for i, n in enumerate([3, 1, 7, 2, 5])]
    batches = list(util.batchify(iter(items), 4))
    assert [[[k, list(r)] for k, r in ks] for ks, _ in batches] == g["batch_keys"]
    assert [list(v.shape) for _, v in batches] == g["batch_shapes"]
    assert [float(v.sum()) for _, v in batches] == g["batch_sums"]
    rebuilt = list(util.unbatchify(batches))
    assert [[k, list(v.shape), float(v.sum())] for k, v in rebuilt] == g["unbatch"]
    for (k0, v0), (k1, v1) in zip(items, rebuilt):
        assert k0 == k1 and torch.equal(v0, v1)


def test_unbatchify_dict_values_and_empty():
    """crf.basecall feeds unbatchify with dict-of-tensors batches keyed like batchify's output."""
    assert list(util.batchify(iter([]), 4)) == []
    keys0 = (("a", (0, 3)), ("b", (3, 4)))
    keys1 = (("b", (0, 2)),)
    b0 = {"x": np.arange(4), "y": np.arange(4) * 2}
    b1 = {"x": np.arange(4, 6), "y": np.arange(4, 6) * 2}
    out = list(util.unbatchify(iter([(keys0, b0), (keys1, b1)])))
    assert [k for k, _ in out] == ["a", "b"]
    assert out[0][1]["y"].tolist() == [0, 2, 4]
    assert out[1][1]["x"].tolist() == [3, 4, 5]


def test_phred_and_mean_qscore():
    assert util.phred(0.0) == "!" and util.phred(1.0) == chr(33 + 40)
    assert util.phred(0.9) == chr(33 + 10)
    assert abs(util.mean_qscore_from_qstring("5555") - 20.0) < 1e-6
    assert util.mean_qscore_from_qstring("") == 0.0


def test_config_defaults_and_rounding():
    cfg = util.set_config_defaults({}, None, None, None)
    assert cfg["basecaller"] == {"chunksize": 4000, "overlap": 500, "batchsize": 64, "quantize": False}
    cfg = util.set_config_defaults({"basecaller": {"chunksize": 10000, "overlap": 500, "batchsize": 96}}, batchsize=512)
    assert cfg["basecaller"]["batchsize"] == 512 and cfg["basecaller"]["chunksize"] == 10000


def test_match_names_by_shape_order():
    from bonito_amd import nn as bnn
    from conftest import load_nn_fixture
    cfg, sd, _, _ = load_nn_fixture("lstm32_sl2")
    model = bnn.from_dict(cfg)
    renamed = {"foo.%d" % i: v for i, (k, v) in enumerate(sd.items())}
    remap = util.match_names(renamed, model)
    assert list(remap.values()) == list(sd.keys())


def test_load_symbol_maps_reference_packages():
    from bonito_amd.crf import Model, basecall
    cfg = {"model": {"package": "bonito.crf"}}
    assert util.load_symbol(cfg, "Model") is Model
    assert util.load_symbol(cfg, "basecall") is basecall


def test_chunk_batches_equals_chunk_plus_batchify():
    """The product path's fused chunk -> batch -> fp16 generator yields exactly what the reference-shaped pair yields."""
    import torch
    from bonito_amd.crf.basecall import chunk_batches
    from bonito_amd.util import batchify, chunk

    class R:
        def __init__(self, i, x):
            self.read_id, self.signal = i, x

    rng = np.random.default_rng(0)
    for cs, ov, bs in ((1000, 100, 7), (996, 498, 16), (400, 0, 5)):
        reads = [R(i, rng.standard_normal(int(n)).astype(np.float32)) for i, n in enumerate(rng.integers(50, 6000, 30))]
        ref = list(batchify((((r, 0, len(r.signal)), chunk(torch.from_numpy(r.signal), cs, ov)) for r in reads), batchsize=bs))
        got = [(k, b.clone()) for k, b in chunk_batches(reads, cs, ov, bs, nbuf=2)]
        assert len(ref) == len(got)
        for (rk, rb), (gk, gb) in zip(ref, got):
            assert len(rk) == len(gk)
            for a, b in zip(rk, gk):
                assert a[0][0] is b[0][0] and a[0][1:] == b[0][1:] and tuple(a[1]) == tuple(b[1])
            assert torch.equal(rb.to(torch.float16), gb)


def test_stitch_vectorised_path_equals_slice_list():
    import torch
    from bonito_amd.util import concat, stitch
    rng = np.random.default_rng(3)
    cs, ov, stride = 996, 498, 6
    for length in (996, 2000, 5555, 9000):
        step = cs - ov
        stub = (length - ov) % step
        n = (length - stub - ov) // step + (1 if stub > 0 else 0)
        chunks = torch.from_numpy(rng.integers(0, 100, (n, cs // stride)).astype(np.int8))
        got = stitch(chunks, cs, ov, length, stride)
        if n == 1:
            want = chunks[0]
        else:
            semi = ov // 2
            start, end = semi // stride, (cs - semi) // stride
            first_end = (stub + semi) // stride if stub > 0 else end
            want = concat([chunks[0, :first_end], *chunks[1:-1, start:end], chunks[-1, start:]])
        assert torch.equal(got, want)


@pytest.mark.parametrize("reverse", [False, True])
def test_stitch_planes_equals_stitch_results(reverse):
    import torch
    from bonito_amd.crf.basecall import stitch_planes, stitch_results
    rng = np.random.default_rng(4)
    cs, ov, stride = 996, 498, 6
    for length in (300, 996, 2000, 5555, 9000):
        step = cs - ov
        if length < cs:
            n = 1
        else:
            stub = (length - ov) % step
            n = (length - stub - ov) // step + (1 if stub > 0 else 0)
        planes = torch.from_numpy(rng.integers(0, 100, (3, n, cs // stride)).astype(np.int8))
        want = stitch_results({"sequence": planes[0], "qstring": planes[1], "moves": planes[2]}, length, cs, ov, stride, reverse=reverse)
        got = stitch_planes(planes, length, cs, ov, stride, reverse=reverse)
        for i, k in enumerate(("sequence", "qstring", "moves")):
            assert torch.equal(got[i], want[k]), (length, k)


def test_signal_chunk_table_equals_util_chunk():
    """Host half of the device ingest: chunk origins must be util.chunk's (checked on an index ramp as the signal)."""
    import torch
    from bonito_amd.signal import chunk_table
    from bonito_amd.util import chunk
    lengths = [12000, 4000, 3999, 700, 8001, 4500, 10]
    trims = [250, 0, 10, 10, 1, 500, 10]
    for cs, ov in ((4000, 500), (996, 498), (1000, 0)):
        reads, starts, avail = chunk_table(lengths, trims, cs, ov)
        row = 0
        for r, (n, t0) in enumerate(zip(lengths, trims)):
            sig = torch.arange(t0, n, dtype=torch.float32)
            if len(sig) == 0:
                continue
            want = chunk(sig, cs, ov)[:, 0]
            for w in want:
                assert reads[row] == r
                if avail[row] >= cs:
                    assert starts[row] == int(w[0]) and int(w[-1]) == starts[row] + cs - 1
                else:                       # tiled short read
                    assert starts[row] == t0 and avail[row] == n - t0
                    assert torch.equal(w, torch.arange(cs, dtype=torch.float32) % avail[row] + t0)
                row += 1
        assert row == len(reads)


@pytest.mark.parametrize("rna", [False, True])
def test_fmt_planes_equals_fmt(rna):
    # shared-index fast path (decoder outputs: base and quality exactly where moves == 1) and the generic fallback
    from bonito_amd.crf.basecall import fmt, fmt_planes
    rng = np.random.default_rng(3)
    L = 5000
    m = (rng.random(L) < 0.55).astype(np.int8)
    seq = rng.choice(np.frombuffer(b"ACGT", np.int8), L) * m
    qs = rng.integers(34, 83, L).astype(np.int8) * m
    consistent = torch.from_numpy(np.stack([seq, qs, m]))
    odd = consistent.clone()
    odd[0, 17] = 0 if odd[0, 17] != 0 else 65          # a base without a move / a move without a base
    empty = torch.zeros((3, 40), dtype=torch.int8)
    for planes in (consistent, odd, empty):
        want = fmt(6, {"sequence": planes[0], "qstring": planes[1], "moves": planes[2]}, rna)
        got = fmt_planes(6, planes, rna)
        assert got["sequence"] == want["sequence"] and got["qstring"] == want["qstring"] and got["stride"] == 6
        assert np.array_equal(got["moves"], want["moves"])


def test_host_chunk_rows_cast_equals_torch_half():
    # the library's chunk gather: same rows as util.chunk, same fp32 -> fp16 rounding as torch (ties, subnormals, overflow)
    import ctypes as C
    from bonito_amd import _lib
    rng = np.random.default_rng(11)
    special = np.array([0.0, -0.0, 1.0, 65504.0, 65519.9, 65520.0, 1e6, -1e6, np.inf, -np.inf, 6.1e-5, 6.0e-5, 5.96e-8, 2.98e-8,
                        2.9802322e-8, 3e-8, 1e-10, 1.00048828125, 1.0009765625 + 2 ** -12, 0.1, -0.3333333], np.float32)
    sig = np.concatenate([special, (rng.standard_normal(5000) * np.exp(rng.uniform(-12, 12, 5000))).astype(np.float32)])
    T, chunksize, overlap = sig.shape[0], 700, 130
    want = util.chunk(torch.from_numpy(sig), chunksize, overlap).reshape(-1, chunksize).to(torch.float16)
    got = np.empty((want.shape[0], chunksize), np.float16)
    n = _lib.lib().bh_host_chunk_rows(sig.ctypes.data, T, chunksize, overlap, 0, want.shape[0], got.ctypes.data)
    assert n == want.shape[0]
    assert np.array_equal(got.view(np.uint16), want.numpy().view(np.uint16))
    part = np.empty((3, chunksize), np.float16)
    assert _lib.lib().bh_host_chunk_rows(sig.ctypes.data, T, chunksize, overlap, 2, 3, part.ctypes.data) == 3
    assert np.array_equal(part.view(np.uint16), want[2:5].numpy().view(np.uint16))
    assert _lib.lib().bh_host_chunk_rows(sig.ctypes.data, T, chunksize, overlap, want.shape[0] - 1, 2, part.ctypes.data) < 0


def test_limit_host_threads_shares_the_budget_between_local_ranks(monkeypatch):
    before = torch.get_num_threads()
    try:
        torch.set_num_threads(max(before, 4))
        monkeypatch.setattr(util, "effective_cpu_count", lambda: 16)
        monkeypatch.setenv("LOCAL_WORLD_SIZE", "8")
        assert util.limit_host_threads(4) == min(2, torch.get_num_threads())
        monkeypatch.setenv("LOCAL_WORLD_SIZE", "64")
        assert util.limit_host_threads(4) == 1
    finally:
        torch.set_num_threads(before)


def test_host_compact_and_chunk_rows_edge_cases():
    from bonito_amd import _lib
    lib = _lib.lib()
    for src in (np.zeros(17, np.int8), np.full(9, 65, np.int8), np.array([0, 71, 0, 0, 84, 65, 0], np.int8), np.zeros(0, np.int8)):
        dst = np.zeros(max(1, src.size), np.uint8)
        n = lib.bh_host_compact(src.ctypes.data if src.size else None, src.size, dst.ctypes.data)
        assert n == np.count_nonzero(src) and bytes(dst[:n]) == src[src != 0].astype(np.uint8).tobytes()
    # no stub (T - overlap divisible by the step), a single chunk (T == chunksize), and the rejected shapes
    for T, cs, ov in ((100 + 4 * 900, 1000, 100), (1000, 1000, 100), (2500, 1000, 0)):
        sig = np.arange(T, dtype=np.float32) * 0.25
        want = util.chunk(torch.from_numpy(sig), cs, ov).reshape(-1, cs).to(torch.float16).numpy()
        got = np.empty_like(want)
        assert lib.bh_host_chunk_rows(sig.ctypes.data, T, cs, ov, 0, want.shape[0], got.ctypes.data) == want.shape[0]
        assert np.array_equal(got.view(np.uint16), want.view(np.uint16))
    sig = np.zeros(500, np.float32)
    out = np.empty((1, 1000), np.float16)
    assert lib.bh_host_chunk_rows(sig.ctypes.data, 500, 1000, 100, 0, 1, out.ctypes.data) < 0     # T < chunksize: Python path
    assert lib.bh_host_chunk_rows(sig.ctypes.data, 500, 100, 100, 0, 1, out.ctypes.data) < 0      # overlap >= chunksize


@pytest.mark.parametrize("mode", ["fastq", "fasta", "sam"])
@pytest.mark.parametrize("reverse,rna", [(False, False), (True, False), (False, True)])
def test_fused_record_formatter_writes_the_bytes_of_the_python_path(mode, reverse, rna):
    """`records_from_planes` (one `bh_host_format_read` call per read: stitch + to_str + record text in the library) against
    unbatchify -> stitch_planes -> fmt_planes -> io.format_record on the same decoded planes: identical text, summary rows and log
    entries - reads shorter than a chunk, exactly one chunk, many chunks, reads split over several engine calls, empty calls
    (filtered), a q-score filter, overlap 0 and an overlap that is not a multiple of the stride."""
    import importlib
    from bonito_amd import io as bio
    bc = importlib.import_module("bonito_amd.crf.basecall")

    class Read:
        filename, channel, mux, start, duration, template_start, template_duration = "f.pod5", 3, 1, 0.5, 2.0, 0.1, 1.9

        def __init__(self, i, n, run):
            self.read_id, self.num_samples, self.signal_len, self.run_id, self.trimmed_samples = "read-%d" % i, n + 7 * i, n, run, 7 * i
        signal = None

    rng = np.random.default_rng(11)
    for chunksize, overlap, stride, batch in [(1200, 120, 6, 7), (1000, 0, 5, 5), (1210, 121, 11, 16), (600, 60, 6, 3)]:
        T = chunksize // stride
        lens = [chunksize // 3, chunksize, chunksize + 1, 2 * chunksize - overlap, 5 * chunksize + 17, 3 * chunksize, 11 * chunksize + 333, chunksize - 1]
        reads = [Read(i, n, "run%d" % (i % 2) if i % 3 else None) for i, n in enumerate(lens)]

        def n_chunks(n):
            if n < chunksize:
                return 1
            step = chunksize - overlap
            stub = (n - overlap) % step
            return (n - stub - chunksize) // step + 1 + (1 if stub > 0 else 0)

        def batches():
            keys, pos, planes = [], 0, None
            for r in reads:
                n, lo = n_chunks(r.signal_len), 0
                key = (r, 0, r.signal_len)
                while lo < n:
                    if planes is None:
                        mv = (rng.random((batch, T)) < 0.4).astype(np.int8)
                        if r.read_id == "read-5":
                            mv[:] = 0                                       # a read that calls nothing: filtered out
                        seq = np.where(mv != 0, np.array([65, 67, 71, 84], np.int8)[rng.integers(0, 4, (batch, T))], 0).astype(np.int8)
                        qs = np.where(mv != 0, rng.integers(34, 80, (batch, T)), 0).astype(np.int8)
                        planes = torch.from_numpy(np.stack([seq, qs, mv]))
                    take = min(n - lo, batch - pos)
                    keys.append((key, (pos, pos + take)))
                    pos += take
                    lo += take
                    if pos == batch:
                        yield tuple(keys), planes
                        keys, pos, planes = [], 0, None
            if pos:
                yield tuple(keys), planes[:, :pos]

        cached = list(batches())
        for min_q in (0.0, 14.0):
            want = [bio.format_record(read, bc.fmt_planes(stride, bc.stitch_planes(sc, end - start, chunksize, overlap, stride, reverse), rna), mode, min_q)
                    for ((read, start, end), sc) in util.unbatchify(iter(cached), dim=1)]
            got = list(bc.records_from_planes(iter(cached), chunksize, overlap, stride, mode, min_q, reverse, rna))
            assert len(got) == len(want) == len(reads)
            for g, w in zip(got, want):
                assert g == w
            assert min_q > 0 or any(t is not None for t, _, _ in got)
        assert any(t is None for t, _, _ in got)                            # ... and the q-score filter did drop reads


@pytest.mark.parametrize("mode", ["fastq", "sam", "fasta"])
def test_fused_record_formatter_edge_cases_follow_format_record(mode):
    """Advisor findings (rounds 3 and 4) on `bh_host_format_read` vs `io.format_record`: a decoded qstring "*" beside ONE base is a real
    quality (Q9: mean 9.0, written unchanged - the reference's semantics, bonito/io.py:431-433), an EMPTY qstring beside a sequence
    means "no qualities" ('!' per base in FASTQ, '*' in SAM, mean 0.0), a read may span more than
    64 engine calls (tiny batches / `--per-call 1` / ultra-long reads), and a non-ASCII read id is carried through as UTF-8."""
    import importlib
    from bonito_amd import io as bio
    bc = importlib.import_module("bonito_amd.crf.basecall")

    class Read:
        filename, channel, mux, start, duration, template_start, template_duration = "f.pod5", 3, 1, 0.5, 2.0, 0.1, 1.9
        signal, run_id, trimmed_samples = None, "runX", 0

        def __init__(self, rid, n):
            self.read_id, self.num_samples, self.signal_len = rid, n, n

    chunksize, overlap, stride = 600, 60, 6
    T = chunksize // stride

    def planes_for(seq_row, qs_row):
        mv = (np.asarray(seq_row) != 0).astype(np.int8)
        return torch.from_numpy(np.stack([np.asarray(seq_row, np.int8)[None], np.asarray(qs_row, np.int8)[None], mv[None]]))

    cases = []
    one_base = np.zeros(T, np.int8); one_base[5] = 65
    star = np.zeros(T, np.int8); star[5] = ord("*")                       # a single base with q = 9: the qstring IS "*"
    cases.append((Read("star", chunksize), planes_for(one_base, star)))
    three = np.zeros(T, np.int8); three[[3, 9, 40]] = [65, 67, 71]
    cases.append((Read("noqual", chunksize), planes_for(three, np.zeros(T, np.int8))))     # sequence without any quality
    qs3 = np.zeros(T, np.int8); qs3[[3, 9, 40]] = [40, 50, 60]
    cases.append((Read("réad-ü", chunksize), planes_for(three, qs3)))                      # non-ASCII id
    for read, planes in cases:
        keys = (((read, 0, read.signal_len), (0, 1)),)
        for min_q in (0.0, 5.0):
            want = [bio.format_record(read, bc.fmt_planes(stride, bc.stitch_planes(planes, read.signal_len, chunksize, overlap, stride), False), mode, min_q)]
            got = list(bc.records_from_planes(iter([(keys, planes)]), chunksize, overlap, stride, mode, min_q))
            assert got == want, (read.read_id, min_q)
            if read.read_id == "star":           # Q9 passes a threshold of 5 and is written as what it is
                assert got[0][0] is not None and (mode == "fasta" or "*" in got[0][0]) and abs(got[0][1][-1] - 9.0) < 1e-6
            if read.read_id == "noqual":
                if min_q > 0:
                    assert got[0][0] is None
                elif mode == "fastq":
                    assert got[0][0].endswith("ACG\n+\n!!!\n")
                elif mode == "sam":
                    assert "\tACG\t*\tNM:i:0" in got[0][0]

    # one read of 150 chunks delivered as 150 engine calls of one chunk each (> 64 pieces)
    rng = np.random.default_rng(5)
    n = 150
    length = chunksize + (n - 1) * (chunksize - overlap)
    read = Read("long", length)
    mv = (rng.random((n, T)) < 0.4).astype(np.int8)
    seq = np.where(mv != 0, np.array([65, 67, 71, 84], np.int8)[rng.integers(0, 4, (n, T))], 0).astype(np.int8)
    qs = np.where(mv != 0, rng.integers(34, 80, (n, T)), 0).astype(np.int8)
    allp = torch.from_numpy(np.stack([seq, qs, mv]))
    key = (read, 0, length)
    calls = [(((key, (0, 1)),), allp[:, i:i + 1].contiguous()) for i in range(n)]
    want = bio.format_record(read, bc.fmt_planes(stride, bc.stitch_planes(allp, length, chunksize, overlap, stride), False), mode, 0.0)
    got = list(bc.records_from_planes(iter(calls), chunksize, overlap, stride, mode, 0.0))
    assert got == [want] and want[0] is not None
