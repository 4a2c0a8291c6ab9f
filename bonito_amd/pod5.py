"""
A reader for ``.pod5`` files that does not need the ``pod5`` wheel (absent here, and its decoder is compiled code): the container
is parsed directly, the tables are read with ``pyarrow``, and the signal codec is undone with pyarrow's zstd binding plus one loop
in the host library (``bh_host_svb16_decode``). It yields records with the attribute names /root/reference bonito/pod5.py:16-67
reads from ``pod5.Reader(...).reads()`` (``read_id``, ``signal``, ``calibration.offset/.scale``, ``pore.channel/.well``,
``start_sample``, ``read_number``, ``sample_count``, ``run_info.acquisition_id/.sample_rate/.context_tags/...``), so that
``bonito_amd.reader.Reader`` treats both sources alike ("needs a pod5 reader (Arrow IPC + VBZ codec)").

FORMAT UNPINNED: there is no ``.pod5`` file anywhere under /root/reference and no wheel to write one with, so nothing here can be
checked against a file MinKNOW or the pod5 library produced. What is implemented is the PUBLISHED layout of the format (pod5
file-format specification, v0.3 "flattened reads table"), restated:

* file  = signature ``\\x8bPOD\\r\\n\\x1a\\n`` | section marker (16 bytes) | embedded Arrow IPC files, each padded to 8 bytes and followed by
  the section marker | ``FOOTER\\0\\0`` | footer (FlatBuffers, padded to 8) | footer length (int64 LE) | section marker | signature;
* footer = table Footer { file_identifier: string; software: string; pod5_version: string; contents: [EmbeddedFile] },
  EmbeddedFile { offset: int64; length: int64; format: short (0 = FeatherV2); content_type: short (0 reads, 1 signal, 2 read-id
  index, 3 other index, 4 run info) };
* signal table: ``read_id`` (16-byte uuid), ``signal`` (large_binary, VBZ: zstd over streamvbyte-16 over zig-zag first differences -
  or a large_list<int16> when uncompressed), ``samples`` (uint32);
* reads table: ``read_id``, ``signal`` (list<uint64>: the read's rows of the signal table, in order), ``read_number``, ``start``,
  ``channel``, ``well``, ``calibration_offset``, ``calibration_scale``, ``num_samples``, ``run_info`` (dictionary<string>: acquisition id), ...;
* run-info table: ``acquisition_id``, ``sample_rate``, ``sample_id``, ``flow_cell_id``, ``sequencer_position``, ``acquisition_start_time``,
  ``context_tags`` / ``tracking_id`` (maps), ...

tests/pod5_fixture.py writes files of exactly this layout (the writer side exists only there); tests/test_pod5.py round-trips them and
checks that basecalling a ``.pod5`` equals basecalling the same reads from ``.npy``.
"""
import mmap
import struct
import uuid
from types import SimpleNamespace

import numpy as np

SIGNATURE = b"\x8bPOD\r\n\x1a\n"
FOOTER_MAGIC = b"FOOTER\x00\x00"
CONTENT_READS, CONTENT_SIGNAL, CONTENT_READ_ID_INDEX, CONTENT_OTHER_INDEX, CONTENT_RUN_INFO = range(5)


class Pod5FormatError(ValueError):
    pass


# ---- FlatBuffers, read side: just enough for the footer -------------------------------------------------------------------------------
def _fb_table(buf, pos):
    """-> (table position, list of field offsets relative to it; 0 = absent / default)"""
    soff = struct.unpack_from("<i", buf, pos)[0]
    vt = pos - soff
    vsize = struct.unpack_from("<H", buf, vt)[0]
    nf = (vsize - 4) // 2
    return pos, list(struct.unpack_from("<%dH" % nf, buf, vt + 4)) if nf else []


def _fb_indirect(buf, pos):
    return pos + struct.unpack_from("<I", buf, pos)[0]


def _fb_string(buf, tab, fields, k):
    if k >= len(fields) or not fields[k]:
        return ""
    p = _fb_indirect(buf, tab + fields[k])
    n = struct.unpack_from("<I", buf, p)[0]
    return bytes(buf[p + 4:p + 4 + n]).decode("utf-8", "replace")


def _fb_scalar(buf, tab, fields, k, fmt, default=0):
    if k >= len(fields) or not fields[k]:
        return default
    return struct.unpack_from(fmt, buf, tab + fields[k])[0]


def parse_footer(buf):
    """FlatBuffers Footer -> dict(file_identifier, software, pod5_version, contents=[dict(offset, length, format, content_type)])."""
    root = struct.unpack_from("<I", buf, 0)[0]
    tab, fields = _fb_table(buf, root)
    out = {"file_identifier": _fb_string(buf, tab, fields, 0), "software": _fb_string(buf, tab, fields, 1),
           "pod5_version": _fb_string(buf, tab, fields, 2), "contents": []}
    if len(fields) > 3 and fields[3]:
        vec = _fb_indirect(buf, tab + fields[3])
        n = struct.unpack_from("<I", buf, vec)[0]
        for i in range(n):
            et, ef = _fb_table(buf, _fb_indirect(buf, vec + 4 + 4 * i))
            out["contents"].append({"offset": _fb_scalar(buf, et, ef, 0, "<q"), "length": _fb_scalar(buf, et, ef, 1, "<q"),
                                    "format": _fb_scalar(buf, et, ef, 2, "<h"), "content_type": _fb_scalar(buf, et, ef, 3, "<h")})
    return out


# ---- the signal codec --------------------------------------------------------------------------------------------------------------------
def vbz_decode(block, count, out=None):
    """One VBZ-compressed signal block -> int16[count] (written into `out` when given): zstd, then streamvbyte-16 / zig-zag / first
    differences in the host library (a table-driven loop, 1 ns per sample; both library calls drop the interpreter lock)."""
    import ctypes as C

    import pyarrow as pa

    from bonito_amd import _lib
    count = int(count)
    if out is None:
        out = np.empty(count, np.int16)
    if count == 0:
        return out
    # an svb16 block of `count` values is at most ceil(count / 8) + 2 count bytes (+ 16 so that the decoder's wide path may over-read).
    # libzstd itself (this image ships libzstd.so.1) needs no size up front; pyarrow's binding does, and gets the frame header's content size
    bound = (count + 7) // 8 + 2 * count
    raw, n = _zstd_decompress(block, bound + 16)
    if raw is None:
        data = pa.Codec("zstd").decompress(block, decompressed_size=_zstd_content_size(block, bound), asbytes=True)
        src = np.frombuffer(data, np.uint8)
        ptr, n = src.ctypes.data_as(C.c_void_p), len(src)
    else:
        ptr = C.cast(raw, C.c_void_p)
    used = _lib.lib().bh_host_svb16_decode(ptr, n, count, out.ctypes.data_as(C.c_void_p))
    if used < 0:
        raise Pod5FormatError("VBZ block too short for %d samples" % count)
    return out


def _zstd_content_size(block, bound):
    """Frame_Content_Size of a zstd frame header (RFC 8878 3.1.1.1) or `bound` when the frame does not carry it."""
    b = bytes(block[:18])
    if len(b) < 6 or b[:4] != b"\x28\xb5\x2f\xfd":
        raise Pod5FormatError("signal block is not a zstd frame")
    desc = b[4]
    fcs_flag, single, dict_flag = desc >> 6, (desc >> 5) & 1, desc & 3
    pos = 5 + (0 if single else 1) + (0, 1, 2, 4)[dict_flag]
    size = {0: 1 if single else 0, 1: 2, 2: 4, 3: 8}[fcs_flag]
    if size == 0:
        return bound
    v = int.from_bytes(b[pos:pos + size], "little")
    return v + 256 if size == 2 else v


_ZSTD = None
_ZSTD_BUF = {}          # per thread: a reusable output buffer (grown on demand)


def _zstd_decompress(data, bound):
    """libzstd through ctypes where the shared object is on the box (this image: libzstd.so.1): no size needed up front. Returns (a
    ctypes buffer that stays valid until this thread's next call, bytes decompressed) or (None, 0) when the library is not there."""
    import ctypes as C
    import ctypes.util
    import threading
    global _ZSTD
    if _ZSTD is None:
        name = ctypes.util.find_library("zstd")
        if not name:
            _ZSTD = False
        else:
            lib = C.CDLL(name)
            lib.ZSTD_decompress.restype = C.c_size_t
            lib.ZSTD_decompress.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
            lib.ZSTD_isError.restype = C.c_uint
            lib.ZSTD_isError.argtypes = [C.c_size_t]
            _ZSTD = lib
    if not _ZSTD:
        return None, 0
    tid = threading.get_ident()
    dst = _ZSTD_BUF.get(tid)
    if dst is None or len(dst) < bound:
        dst = _ZSTD_BUF[tid] = C.create_string_buffer(max(bound, 1 << 18))
    src = np.frombuffer(data, np.uint8)          # zero-copy view of the arrow buffer / bytes
    n = _ZSTD.ZSTD_decompress(dst, len(dst), src.ctypes.data_as(C.c_void_p), len(src))
    if _ZSTD.ZSTD_isError(n):
        raise Pod5FormatError("zstd: corrupt signal block")
    return dst, int(n)


def svb16_decode_numpy(raw, count):
    """The same inner codec in numpy (tests compare it with the library's loop)."""
    raw = np.frombuffer(raw, np.uint8)
    if count == 0:
        return np.zeros(0, np.int16)
    nkeys = (count + 7) // 8
    two = np.unpackbits(raw[:nkeys], bitorder="little")[:count].astype(np.int64)
    off = nkeys + np.concatenate([[0], np.cumsum(1 + two)[:-1]])
    lo = raw[off].astype(np.uint16)
    hi = np.where(two == 1, raw[np.minimum(off + 1, len(raw) - 1)], 0).astype(np.uint16)
    v = lo | (hi << 8)
    delta = (v >> 1) ^ (np.uint16(0) - (v & 1))
    return np.cumsum(delta.astype(np.uint16), dtype=np.uint16).view(np.int16)


# ---- the file ------------------------------------------------------------------------------------------------------------------------------
class Reader:
    """``with Reader(path) as fh: for read in fh.reads(): ...`` - the subset of pod5.Reader the basecaller uses."""

    def __init__(self, path):
        self.path = str(path)
        self._fh = open(self.path, "rb")
        self._mm = mmap.mmap(self._fh.fileno(), 0, access=mmap.ACCESS_READ)
        mm = self._mm
        if len(mm) < 2 * len(SIGNATURE) + 32 + 8 or mm[:8] != SIGNATURE or mm[-8:] != SIGNATURE:
            raise Pod5FormatError("%s: not a pod5 file (signature)" % self.path)
        self.section_marker = bytes(mm[8:24])
        if mm[-24:-8] != self.section_marker:
            raise Pod5FormatError("%s: section markers of head and tail differ" % self.path)
        flen = struct.unpack_from("<q", mm, len(mm) - 32)[0]
        fend = len(mm) - 32
        fstart = fend - flen
        if flen <= 0 or fstart < 24 + 8 or mm[fstart - 8:fstart] != FOOTER_MAGIC:
            raise Pod5FormatError("%s: footer not found" % self.path)
        self.footer = parse_footer(memoryview(mm)[fstart:fend])
        self._tables = {}
        self._signal_cols = None
        self._run_info = None

    # -- context manager / teardown
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def close(self):
        self._tables.clear()
        self._signal_cols = None
        try:
            self._mm.close()
        except (BufferError, ValueError):          # arrow buffers still point into the map: the OS unmaps at exit
            pass
        self._fh.close()

    def _embedded(self, content_type):
        import pyarrow as pa
        if content_type not in self._tables:
            ent = [c for c in self.footer["contents"] if c["content_type"] == content_type]
            if not ent:
                raise Pod5FormatError("%s: no embedded table of content type %d" % (self.path, content_type))
            off, n = ent[0]["offset"], ent[0]["length"]
            if off < 24 or off + n > len(self._mm):
                raise Pod5FormatError("%s: embedded table outside the file" % self.path)
            buf = pa.py_buffer(memoryview(self._mm)[off:off + n])
            self._tables[content_type] = pa.ipc.open_file(buf)
        return self._tables[content_type]

    @property
    def num_reads(self):
        f = self._embedded(CONTENT_READS)
        return sum(f.get_batch(i).num_rows for i in range(f.num_record_batches))

    def run_infos(self):
        """{acquisition_id: namespace(acquisition_id, sample_rate, sample_id, flow_cell_id, sequencer_position, acquisition_start_time,
        context_tags, tracking_id, system_name, ...)}"""
        if self._run_info is None:
            out = {}
            tab = self._embedded(CONTENT_RUN_INFO).read_all()
            cols = {name: tab.column(name).to_pylist() for name in tab.column_names}
            for i in range(tab.num_rows):
                row = {k: v[i] for k, v in cols.items()}
                for k in ("context_tags", "tracking_id"):
                    if isinstance(row.get(k), list):            # arrow map -> list of (key, value)
                        row[k] = dict(row[k])
                    elif row.get(k) is None:
                        row[k] = {}
                if not row.get("sample_rate"):
                    row["sample_rate"] = int(row["context_tags"].get("sample_frequency", 0) or 0)
                out[row.get("acquisition_id", "")] = SimpleNamespace(**row)
            self._run_info = out
        return self._run_info

    def _signal_rows(self):
        """(kind, column accessors) of the signal table, batches concatenated lazily: row i -> (bytes-like | int16 array, samples)"""
        if self._signal_cols is None:
            f = self._embedded(CONTENT_SIGNAL)
            batches = [f.get_batch(i) for i in range(f.num_record_batches)]
            starts = np.cumsum([0] + [b.num_rows for b in batches])
            self._signal_cols = (batches, starts)
        return self._signal_cols

    def _signal_of(self, rows, total):
        import pyarrow as pa
        batches, starts = self._signal_rows()
        out = np.empty(int(total), np.int16)
        pos = 0
        for r in rows:
            b = int(np.searchsorted(starts, r, side="right") - 1)
            batch = batches[b]
            k = int(r - starts[b])
            n = int(batch.column(batch.schema.get_field_index("samples"))[k].as_py())
            col = batch.column(batch.schema.get_field_index("signal"))
            if isinstance(col, pa.ExtensionArray):
                col = col.storage
            if pa.types.is_large_binary(col.type) or pa.types.is_binary(col.type):
                vbz_decode(col[k].as_buffer(), n, out=out[pos:pos + n])
            else:                                               # uncompressed: list<int16>
                out[pos:pos + n] = np.asarray(col[k].values.to_numpy(zero_copy_only=False), np.int16)[:n]
            pos += n
        if pos != total:
            raise Pod5FormatError("%s: a read's signal rows hold %d samples, its record says %d" % (self.path, pos, total))
        return out

    def reads(self, selection=None, missing_ok=True, preload=None):
        """Records in file order (`selection`: an iterable of read ids, str or UUID, to keep). Signal rows are decoded per read."""
        import pyarrow as pa
        want = None if selection is None else {str(s) for s in selection}
        infos = self.run_infos()
        f = self._embedded(CONTENT_READS)
        found = set()
        for bi in range(f.num_record_batches):
            batch = f.get_batch(bi)
            names = batch.schema.names

            def col(name, default=None):
                if name not in names:
                    return default
                c = batch.column(names.index(name))
                if isinstance(c, pa.ExtensionArray):
                    c = c.storage
                if pa.types.is_dictionary(c.type):
                    c = c.dictionary_decode()
                return c

            ids = col("read_id")
            sig = col("signal")
            run = col("run_info")
            opt = {k: col(k) for k in ("read_number", "start", "channel", "well", "calibration_offset", "calibration_scale", "num_samples",
                                       "median_before", "end_reason", "pore_type")}
            for i in range(batch.num_rows):
                rid = str(uuid.UUID(bytes=ids[i].as_py()))
                if want is not None and rid not in want:
                    continue
                found.add(rid)
                rows = sig[i].as_py()
                get = lambda k, d=0: (opt[k][i].as_py() if opt[k] is not None else d)
                info = infos.get(run[i].as_py() if run is not None else "", None) or SimpleNamespace(
                    acquisition_id="", sample_rate=0, context_tags={}, tracking_id={}, sample_id="", flow_cell_id="",
                    sequencer_position="", acquisition_start_time=None)
                n = get("num_samples", None)
                if n is None:
                    batches, starts = self._signal_rows()
                    n = 0
                    for r in rows:
                        b = int(np.searchsorted(starts, r, side="right") - 1)
                        n += int(batches[b].column(batches[b].schema.get_field_index("samples"))[int(r - starts[b])].as_py())
                yield Pod5Read(self, rid, rows, int(n), info, SimpleNamespace(channel=get("channel"), well=get("well"), pore_type=get("pore_type", "")),
                               SimpleNamespace(offset=float(get("calibration_offset", 0.0)), scale=float(get("calibration_scale", 1.0))),
                               int(get("start")), int(get("read_number")), float(get("median_before", 0.0) or 0.0), get("end_reason", ""))
        if want is not None and not missing_ok and found != want:
            raise KeyError("read ids not in %s: %s" % (self.path, sorted(want - found)[:5]))


class Pod5Read:
    """One record; ``signal`` (int16 ADC samples) is decoded on first access."""
    __slots__ = ("_reader", "read_id", "_rows", "sample_count", "run_info", "pore", "calibration", "start_sample", "read_number",
                 "median_before", "end_reason", "_signal")

    def __init__(self, reader, read_id, rows, sample_count, run_info, pore, calibration, start_sample, read_number, median_before, end_reason):
        self._reader, self.read_id, self._rows, self.sample_count = reader, read_id, rows, sample_count
        self.run_info, self.pore, self.calibration = run_info, pore, calibration
        self.start_sample, self.read_number, self.median_before, self.end_reason = start_sample, read_number, median_before, end_reason
        self._signal = None

    @property
    def signal(self):
        if self._signal is None:
            self._signal = self._reader._signal_of(self._rows, self.sample_count)
        return self._signal

    @property
    def signal_pa(self):
        return self.calibration.scale * (self.signal.astype(np.float32) + self.calibration.offset)

