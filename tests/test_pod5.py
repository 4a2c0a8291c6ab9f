"""pod5 ingest without the pod5 wheel (reference seam /root/reference bonito/pod5.py:52-67,113-124): container footer,
Arrow tables, VBZ codec. FORMAT UNPINNED - no .pod5 exists under /root/reference, so the fixtures are written by tests/pod5_fixture.py
from the published layout; what IS checked: the codec against an independent numpy restatement and hand-made byte strings, the
container round trip, and that a Reader over a .pod5 yields the very reads (ids, metadata, normalised signal) it yields from .npy."""
import json
import os
import uuid

import numpy as np
import pytest

from bonito_amd import pod5, reader
from pod5_fixture import build_footer, svb16_encode, vbz_encode, write_pod5


def _signals(rng, n, lo=200, hi=30000):
    out = []
    for i in range(n):
        m = int(rng.integers(lo, hi))
        x = np.cumsum(rng.integers(-40, 41, m)) + 400 + (rng.standard_normal(m) * 15)
        if i % 3 == 0:
            x[50:300] += 700                                   # an adapter-like peak so that trim() has something to find
        out.append(np.clip(x, -2048, 2047).astype(np.int16))
    return out


def test_svb16_known_bytes_and_edge_values():
    # by hand: samples 3, 2, 300 -> deltas 3, -1, 298 -> zig-zag 6, 1, 596 = 0x0254 -> keys 0b100, data 06 01 54 02
    assert svb16_encode(np.array([3, 2, 300], np.int16)) == bytes([0b100, 6, 1, 0x54, 0x02])
    import ctypes as C
    from bonito_amd import _lib
    raw = bytes([0b100, 6, 1, 0x54, 0x02])
    out = np.zeros(3, np.int16)
    src = np.frombuffer(raw, np.uint8)
    used = _lib.lib().bh_host_svb16_decode(src.ctypes.data_as(C.c_void_p), len(raw), 3, out.ctypes.data_as(C.c_void_p))
    assert used == 5 and out.tolist() == [3, 2, 300]
    # truncated input is an error, not a read past the end
    assert _lib.lib().bh_host_svb16_decode(src.ctypes.data_as(C.c_void_p), 4, 3, out.ctypes.data_as(C.c_void_p)) == -1
    # extremes and wrap-around of the 16-bit differences
    edge = np.array([32767, -32768, -1, 0, 1, -32768, 32767, 255, 256, -128, -129], np.int16)
    rng = np.random.default_rng(0)
    for x in (edge, rng.integers(-32768, 32768, 1001).astype(np.int16), np.zeros(0, np.int16), np.array([7], np.int16)):
        enc = svb16_encode(x)
        assert np.array_equal(pod5.svb16_decode_numpy(enc, len(x)), x)
        assert np.array_equal(pod5.vbz_decode(vbz_encode(x), len(x)), x)


def test_footer_flatbuffer_round_trip():
    contents = [(24, 1000, 1), (1048, 200, 4), (1272, 4096, 0)]
    fb = build_footer(contents, "0000-id", software="sw", version="0.3.2")
    got = pod5.parse_footer(memoryview(fb))
    assert got["file_identifier"] == "0000-id" and got["software"] == "sw" and got["pod5_version"] == "0.3.2"
    assert [(c["offset"], c["length"], c["content_type"]) for c in got["contents"]] == contents
    assert all(c["format"] == 0 for c in got["contents"])


@pytest.mark.parametrize("compress,batch_reads", [(True, 0), (False, 0), (True, 3)])
def test_pod5_file_round_trip(tmp_path, compress, batch_reads):
    rng = np.random.default_rng(3)
    sigs = _signals(rng, 7)
    reads = [{"read_id": str(uuid.UUID(int=1000 + i)), "signal": s, "offset": -240.0 + i, "scale": 0.1755 + 0.001 * i, "channel": 10 + i,
              "well": 1 + i % 4, "start": 5000 * i, "read_number": 100 + i} for i, s in enumerate(sigs)]
    path = write_pod5(str(tmp_path / "a.pod5"), reads, compress=compress, rows=4096, sample_rate=4000, batch_reads=batch_reads)
    with pod5.Reader(path) as fh:
        assert fh.num_reads == 7 and fh.footer["pod5_version"] == "0.3.2"
        got = list(fh.reads())
        assert [g.read_id for g in got] == [r["read_id"] for r in reads]
        for g, r in zip(got, reads):
            assert np.array_equal(g.signal, r["signal"]) and g.signal.dtype == np.int16 and g.sample_count == len(r["signal"])
            assert g.pore.channel == r["channel"] and g.pore.well == r["well"] and g.start_sample == r["start"] and g.read_number == r["read_number"]
            assert abs(g.calibration.offset - r["offset"]) < 1e-4 and abs(g.calibration.scale - r["scale"]) < 1e-6
            assert g.run_info.acquisition_id == "acq-test-0001" and g.run_info.sample_rate == 4000
            assert g.run_info.context_tags["sample_frequency"] == "4000" and g.run_info.sample_id == "sample-1"
        sel = [reads[5]["read_id"], reads[1]["read_id"]]
        assert [g.read_id for g in fh.reads(selection=sel)] == [reads[1]["read_id"], reads[5]["read_id"]]      # file order
        with pytest.raises(KeyError):
            list(fh.reads(selection=[str(uuid.UUID(int=5))], missing_ok=False))


def test_not_a_pod5_file_is_refused(tmp_path):
    p = tmp_path / "x.pod5"
    p.write_bytes(b"\x89PNG\r\n\x1a\n" + b"\0" * 100)
    with pytest.raises(pod5.Pod5FormatError):
        pod5.Reader(str(p))
    good = write_pod5(str(tmp_path / "g.pod5"), [{"read_id": str(uuid.UUID(int=1)), "signal": np.arange(100, dtype=np.int16)}])
    data = bytearray(open(good, "rb").read())
    data[-30] ^= 0xFF                                           # the tail's section marker no longer matches the head's
    p.write_bytes(bytes(data))
    with pytest.raises(pod5.Pod5FormatError):
        pod5.Reader(str(p))


def test_reader_yields_the_same_reads_from_pod5_and_from_npy(tmp_path):
    """bonito_amd.reader.Reader over a .pod5 == over the equivalent .npy + side-car files: ids, metadata, shift / scale, trim, the
    normalised signal bit for bit (the per-read flow of bonito/pod5.py:52-67), the rank / world shard and the raw int16 path."""
    rng = np.random.default_rng(8)
    sigs = _signals(rng, 9)
    ids = [str(uuid.UUID(int=77 + i)) for i in range(9)]
    d_npy, d_pod = tmp_path / "npy", tmp_path / "pod"
    d_npy.mkdir(); d_pod.mkdir()
    recs = []
    for i, (rid, s) in enumerate(zip(ids, sigs)):
        meta = {"read_id": rid, "offset": -230.0, "scale": 0.18, "channel": 3 + i, "mux": 2, "start": (1000.0 * i) / 5000.0,
                "sample_rate": 5000.0, "run_id": "acq-test-0001"}
        np.save(d_npy / ("r%02d.npy" % i), s)
        (d_npy / ("r%02d.json" % i)).write_text(json.dumps(meta))
        recs.append({"read_id": rid, "signal": s, "offset": -230.0, "scale": 0.18, "channel": 3 + i, "well": 2, "start": 1000 * i})
    write_pod5(str(d_pod / "reads.pod5"), recs, rows=5000)
    for kw in ({}, {"do_trim": False}, {"rank": 1, "world": 3}, {"read_ids": {ids[2], ids[6]}}, {"read_ids": {ids[2]}, "skip": True}, {"n_max": 4}):
        a = list(reader.Reader(str(d_npy)).get_reads(**kw))
        b = list(reader.Reader(str(d_pod)).get_reads(**kw))
        assert [r.read_id for r in a] == [r.read_id for r in b] and len(a) > 0
        for x, y in zip(a, b):
            assert np.array_equal(x.signal, y.signal) and x.signal.dtype == y.signal.dtype == np.float32
            assert (x.shift, x.scale, x.trimmed_samples, x.num_samples) == (y.shift, y.scale, y.trimmed_samples, y.num_samples)
            assert (x.channel, x.mux, x.run_id, x.sample_rate) == (y.channel, y.mux, y.run_id, y.sample_rate)
            assert abs(x.start - y.start) < 1e-12 and abs(x.template_start - y.template_start) < 1e-12
    a = list(reader.Reader(str(d_npy)).get_reads(raw=True))
    b = list(reader.Reader(str(d_pod)).get_reads(raw=True))
    for x, y in zip(a, b):                                       # device ingest: the int16 samples and the calibration travel as they are
        assert np.array_equal(x.raw, y.raw) and y.raw.dtype == np.int16 and abs(x.scaling - y.scaling) < 1e-7 and x.offset == y.offset
