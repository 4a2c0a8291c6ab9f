"""Test helper: WRITES small ``.pod5`` files of the published layout (bonito_amd/pod5.py's docstring), so that the reader has something
to read - there is no ``.pod5`` under /root/reference and no pod5 wheel here. The writer side exists only in tests/ (the product reads).

    write_pod5(path, reads, compress=True, rows=..., sample_rate=5000)

`reads`: list of dicts {read_id (uuid str), signal (int16 array), offset, scale, channel, well, start, read_number}. The signal of a read
is split into rows of `rows` samples (MinKNOW uses 102 400); with `compress` every row is VBZ: zig-zag first differences ->
streamvbyte-16 -> zstd. The footer is a hand-built FlatBuffers table of the schema the format publishes.
"""
import struct
import uuid

import numpy as np
import pyarrow as pa

from bonito_amd.pod5 import FOOTER_MAGIC, SIGNATURE


def svb16_encode(samples):
    """int16 array -> bytes: keys (1 bit per value, LSB first) then 1 or 2 data bytes per value; zig-zag code of the first differences."""
    x = np.asarray(samples, np.int16).astype(np.uint16)
    if len(x) == 0:
        return b""
    prev = np.concatenate([[np.uint16(0)], x[:-1]])
    d = (x - prev).astype(np.uint16).view(np.int16).astype(np.int32)
    v = ((d << 1) ^ (d >> 15)).astype(np.uint16)                 # zig-zag in 16 bits
    two = (v > 0xFF).astype(np.uint8)
    keys = np.packbits(two, bitorder="little")
    lo, hi = (v & 0xFF).astype(np.uint8), (v >> 8).astype(np.uint8)
    data = np.empty(int(len(v) + two.sum()), np.uint8)
    off = np.concatenate([[0], np.cumsum(1 + two.astype(np.int64))[:-1]])
    data[off] = lo
    data[(off + 1)[two == 1]] = hi[two == 1]
    return keys.tobytes() + data.tobytes()


def vbz_encode(samples):
    return pa.Codec("zstd").compress(svb16_encode(samples), asbytes=True)


# ---- FlatBuffers, write side: forward layout, every offset points to a higher address ---------------------------------------------------
def _pad(buf, n):
    while len(buf) % n:
        buf += b"\x00"


def build_footer(contents, file_identifier, software="bonito_amd tests", version="0.3.2"):
    """contents: list of (offset, length, content_type). -> FlatBuffers bytes of table Footer."""
    buf = bytearray(4)                                            # root offset, patched below
    # root vtable: 4 fields
    vt_root = len(buf)
    buf += struct.pack("<HHHHHH", 12, 20, 4, 8, 12, 16)           # vtable size, table size, field offsets
    _pad(buf, 4)
    root = len(buf)
    buf += struct.pack("<i", root - vt_root)
    f_ident, f_soft, f_ver, f_vec = root + 4, root + 8, root + 12, root + 16
    buf += b"\x00" * 16
    struct.pack_into("<I", buf, 0, root)
    # contents vector
    _pad(buf, 4)
    vec = len(buf)
    struct.pack_into("<I", buf, f_vec, vec - f_vec)
    buf += struct.pack("<I", len(contents)) + b"\x00" * (4 * len(contents))
    # one shared vtable for the element tables: offset int64 @8, length int64 @16, format short @24, content_type short @26
    # (table: soffset 4 bytes, 4 bytes padding, then the fields)
    _pad(buf, 2)
    vt_el = len(buf)
    buf += struct.pack("<HHHHHH", 12, 28, 8, 16, 24, 26)
    for i, (off, length, ctype) in enumerate(contents):
        _pad(buf, 8)
        tab = len(buf)
        buf += struct.pack("<i", tab - vt_el) + b"\x00" * 4 + struct.pack("<qqhh", off, length, 0, ctype)
        slot = vec + 4 + 4 * i
        struct.pack_into("<I", buf, slot, tab - slot)
    for slot, text in ((f_ident, file_identifier), (f_soft, software), (f_ver, version)):
        _pad(buf, 4)
        pos = len(buf)
        raw = text.encode()
        struct.pack_into("<I", buf, slot, pos - slot)
        buf += struct.pack("<I", len(raw)) + raw + b"\x00"
    _pad(buf, 8)
    return bytes(buf)


def _ipc(table):
    sink = pa.BufferOutputStream()
    with pa.ipc.new_file(sink, table.schema) as w:
        w.write_table(table)
    return sink.getvalue().to_pybytes()


def write_pod5(path, reads, compress=True, rows=4096, sample_rate=5000, acquisition_id="acq-test-0001", batch_reads=0):
    sig_ids, sig_blocks, sig_samples, read_rows = [], [], [], []
    for r in reads:
        x = np.asarray(r["signal"], np.int16)
        idx = []
        for lo in range(0, max(len(x), 1), rows):
            part = x[lo:lo + rows]
            idx.append(len(sig_blocks))
            sig_ids.append(uuid.UUID(r["read_id"]).bytes)
            sig_blocks.append(vbz_encode(part) if compress else part)
            sig_samples.append(len(part))
        read_rows.append(idx)
    if compress:
        sig_col = pa.array(sig_blocks, pa.large_binary())
    else:
        sig_col = pa.array([b.tolist() for b in sig_blocks], pa.large_list(pa.int16()))
    signal_table = pa.table({"read_id": pa.array(sig_ids, pa.binary(16)), "signal": sig_col,
                             "samples": pa.array(sig_samples, pa.uint32())})
    n = len(reads)
    reads_table = pa.table({
        "read_id": pa.array([uuid.UUID(r["read_id"]).bytes for r in reads], pa.binary(16)),
        "signal": pa.array(read_rows, pa.list_(pa.uint64())),
        "read_number": pa.array([r.get("read_number", i) for i, r in enumerate(reads)], pa.uint32()),
        "start": pa.array([r.get("start", 0) for r in reads], pa.uint64()),
        "median_before": pa.array([r.get("median_before", 200.0) for r in reads], pa.float32()),
        "channel": pa.array([r.get("channel", 1) for r in reads], pa.uint16()),
        "well": pa.array([r.get("well", 1) for r in reads], pa.uint8()),
        "pore_type": pa.array(["not_set"] * n).dictionary_encode(),
        "calibration_offset": pa.array([r.get("offset", 0.0) for r in reads], pa.float32()),
        "calibration_scale": pa.array([r.get("scale", 1.0) for r in reads], pa.float32()),
        "end_reason": pa.array(["signal_positive"] * n).dictionary_encode(),
        "end_reason_forced": pa.array([False] * n),
        "run_info": pa.array([acquisition_id] * n).dictionary_encode(),
        "num_samples": pa.array([len(r["signal"]) for r in reads], pa.uint64()),
    })
    if batch_reads:                                              # several record batches in the reads table
        reads_table = pa.Table.from_batches(reads_table.to_batches(max_chunksize=batch_reads))
    run_table = pa.table({
        "acquisition_id": pa.array([acquisition_id]),
        "acquisition_start_time": pa.array([1700000000000], pa.timestamp("ms", tz="UTC")),
        "adc_max": pa.array([2047], pa.int16()), "adc_min": pa.array([-2048], pa.int16()),
        "context_tags": pa.array([[("sample_frequency", str(sample_rate)), ("experiment_type", "genomic_dna")]], pa.map_(pa.string(), pa.string())),
        "experiment_name": pa.array(["bonito_amd fixture"]), "flow_cell_id": pa.array(["FAX00000"]),
        "flow_cell_product_code": pa.array(["FLO-MIN114"]), "protocol_name": pa.array(["sequencing"]),
        "protocol_run_id": pa.array(["proto-1"]), "protocol_start_time": pa.array([1700000000000], pa.timestamp("ms", tz="UTC")),
        "sample_id": pa.array(["sample-1"]), "sample_rate": pa.array([sample_rate], pa.uint16()),
        "sequencing_kit": pa.array(["sqk-lsk114"]), "sequencer_position": pa.array(["MN00000"]),
        "sequencer_position_type": pa.array(["minion"]), "software": pa.array(["tests/pod5_fixture.py"]),
        "system_name": pa.array(["host"]), "system_type": pa.array(["linux"]),
        "tracking_id": pa.array([[("run_id", acquisition_id), ("exp_start_time", "2023-11-14T22:13:20Z")]], pa.map_(pa.string(), pa.string())),
    })
    marker = uuid.uuid4().bytes
    out = bytearray(SIGNATURE + marker)
    contents = []
    for table, ctype in ((signal_table, 1), (run_table, 4), (reads_table, 0)):
        raw = _ipc(table)
        contents.append((len(out), len(raw), ctype))
        out += raw
        _pad(out, 8)
        out += marker
    out += FOOTER_MAGIC
    footer = build_footer(contents, str(uuid.uuid4()))
    out += footer
    out += struct.pack("<q", len(footer)) + marker + SIGNATURE
    with open(path, "wb") as fh:
        fh.write(out)
    return path
