"""
Basecall output: FASTQ / FASTA / unaligned SAM text and ``summary.tsv`` -- the host-side formatting of
/root/reference bonito/io.py (encode_moves 57-70, write_fastq 97-105, sam_record 135-166, summary 169-226,
Writer 400-469) without pysam / mappy (alignment and BAM/CRAM need those wheels and are out of scope).
SAM tags follow documentation/SAM.md:39-55: ``mv:B:c,<stride>,<moves...>``, ``qs:f`` (mean q-score),
``ns:i`` (samples incl. trimmed), ``ts:i`` (trimmed samples), ``RG:Z``.
"""
import csv
import sys
from threading import Thread

import numpy as np

from bonito_amd.util import mean_qscore_from_qstring

__version__ = "0.1.0"

summary_field_names = ["filename", "read_id", "run_id", "channel", "mux", "start_time", "duration", "template_start",
                       "template_duration", "sequence_length_template", "mean_qscore_template"]


def mean_qscore(qstring):
    """`util.mean_qscore_from_qstring` through the library's helper: the Python records and the ones `bh_host_format_read` writes
    (crf/basecall.py `records_from_planes`) carry the same `qs:f` / summary values because they share the arithmetic."""
    try:
        from bonito_amd import _lib
        raw = qstring.encode("latin-1")
        return float(_lib.lib().bh_host_mean_qscore(raw, len(raw)))
    except OSError:                      # library not built: the numpy restatement (differs in the last bit at most)
        return float(mean_qscore_from_qstring(qstring))


def encode_moves(moves, stride, sep=","):
    """np.array([0,1,0,1,1]), 5 -> '5,0,1,0,1,1' (single-digit moves only)."""
    moves = np.asarray(moves)
    out = np.full(2 * moves.size, ord(sep), dtype=np.uint8)
    out[1::2] = moves.astype(np.uint8) + ord("0")
    return "%d%s" % (stride, out.tobytes().decode("ascii"))


def write_fasta(header, sequence, fd=sys.stdout):
    fd.write(">%s\n%s\n" % (header, sequence))


def write_fastq(header, sequence, qstring, fd=sys.stdout, tags=None, sep="\t"):
    fd.write("@%s%s\n" % (header, (" " + sep.join(tags)) if tags is not None else ""))
    fd.write("%s\n+\n%s\n" % (sequence, qstring))


def sam_header(groups, sep="\t"):
    hd = sep.join(["@HD", "VN:1.5", "SO:unknown", "ob:%s" % __version__])
    pg = sep.join(["@PG", "ID:basecaller", "PN:bonito_amd", "VN:%s" % __version__, "CL:%s" % " ".join(sys.argv),
                   "DS:MI355X engine"])
    return "\n".join([hd, pg, *groups]) + "\n"


def sam_record(read_id, sequence, qstring, mapping=None, tags=None, sep="\t"):
    if mapping is not None:
        raise NotImplementedError("alignment is out of scope for this build (needs mappy)")
    record = [read_id, 4, "*", 0, 0, "*", "*", 0, 0, sequence, qstring, "NM:i:0"]
    if tags is not None:
        record.extend(tags)
    return sep.join(map(str, record))


def read_tags(read, res, mean_q):
    tags = ["RG:Z:%s" % (read.run_id or "unknown"), "qs:f:%0.2f" % mean_q, "ns:i:%d" % read.num_samples,
            "ts:i:%d" % read.trimmed_samples]
    if res.get("moves") is not None and len(res["moves"]):
        tags.append("mv:B:c,%s" % encode_moves(res["moves"], res["stride"]))
    return tags


def summary_row(read, seqlen, qscore):
    return [read.filename, read.read_id, read.run_id, read.channel, read.mux, read.start, read.duration,
            read.template_start, read.template_duration, seqlen, qscore]


def signal_samples(read):
    """Post-trim samples of a read: what the reference logs as ``len(read.signal)`` (io.py:437-440) and the CLI divides
    by the wall time for its `samples per second` line (cli/basecaller.py:156-164)."""
    sig = getattr(read, "signal", None)
    if sig is not None:
        return len(sig)
    n = getattr(read, "signal_len", None)                 # RawRead: filled in by the device ingest
    return int(n) if n is not None else int(read.num_samples) - int(getattr(read, "trimmed_samples", 0))


def format_record(read, res, mode, min_qscore=0.0):
    """One basecalled read -> (text to write or None, summary row or None, (read_id, post-trim samples)).

    The single place that turns a (read, result) pair into output bytes: the in-process `Writer` and the per-rank workers of
    the multi-GPU path (bonito_amd/parallel.py) both call it, so N-rank output is byte-identical to 1-rank output by
    construction. The log entry is produced for EVERY read, before the q-score / empty-sequence filters, like the reference
    (io.py:437-442)."""
    seq, qstring = res["sequence"], res.get("qstring", "*")
    # Reference semantics (bonito/io.py:431-433, util.py:114-121): the mean q-score is ALWAYS computed from the q-string - 9.0 for the
    # "*" sentinel as for a real one-base read with Q9 (whose q-string IS "*"), 0.0 for an empty one. Only the formatting differs: the
    # sentinel / an empty q-string beside a sequence mean "no qualities" ('!' per base in FASTQ, where the reference would write a
    # malformed record; '*' in SAM). A decoded "*" beside ONE base is a quality and is written unchanged (advisor finding, round 4).
    no_quals = len(qstring) == 0 or (qstring == "*" and len(seq) != 1)
    mean_q = mean_qscore(qstring) if len(qstring) else 0.0
    log = (read.read_id, signal_samples(read))
    if mean_q < min_qscore or not len(seq):
        return None, None, log
    if mode == "fasta":
        text = ">%s\n%s\n" % (read.read_id, seq)
    elif mode == "fastq":
        tags = read_tags(read, res, mean_q)
        text = "@%s %s\n%s\n+\n%s\n" % (read.read_id, "\t".join(tags), seq, "!" * len(seq) if no_quals else qstring)
    else:
        text = sam_record(read.read_id, seq, "*" if no_quals else qstring, tags=read_tags(read, res, mean_q)) + "\n"
    return text, summary_row(read, len(seq), mean_q), log


class Writer(Thread):
    """Consumes the basecall iterator (this drives the whole pipeline) and writes records to `fd`.

    `iterator` yields (read, result) pairs, or -- `preformatted=True`, the multi-GPU merge of bonito_amd/parallel.py --
    the (text, summary_row, log) triples `format_record` made of them on the rank that basecalled the read. A pre-formatted
    text may be `bytes` / a memoryview and a row the rendered TSV line as `bytes` (the packed blocks of parallel.ordered_records, the
    library's record text taken as it is): they go to the binary layer of `fd` without a str round trip."""

    def __init__(self, mode, iterator, fd=sys.stdout, min_qscore=0.0, summary_path=None, groups=(), preformatted=False):
        super().__init__()
        assert mode in ("fastq", "fasta", "sam")
        self.mode, self.iterator, self.fd = mode, iterator, fd
        self.min_qscore, self.summary_path, self.groups = min_qscore, summary_path, groups
        self.preformatted = preformatted
        self.log = []            # (read_id, post-trim samples) of every basecalled read: feeds the samples/s report
        self.error = None

    def run(self):
        try:
            summary = open(self.summary_path, "w", newline="") if self.summary_path else None
            tsv = csv.writer(summary, delimiter="\t") if summary else None
            if tsv:
                tsv.writerow(summary_field_names)
            if self.mode == "sam":
                self.fd.write(sam_header(self.groups))
            raw = getattr(self.fd, "buffer", None)          # the binary layer under a text file (sys.stdout, open(..., "w")); None for StringIO
            dirty = self.mode == "sam"                      # text written through the str layer and not yet flushed
            for item in self.iterator:
                text, row, log = item if self.preformatted else format_record(item[0], item[1], self.mode, self.min_qscore)
                self.log.append(log)
                if text is None:
                    continue
                if isinstance(text, str):
                    self.fd.write(text)
                    dirty = True
                elif raw is not None:
                    if dirty:
                        self.fd.flush()
                        dirty = False
                    raw.write(text)
                else:
                    self.fd.write(bytes(text).decode("utf-8"))
                if tsv:
                    if isinstance(row, (bytes, bytearray, memoryview)):
                        summary.write(bytes(row).decode("utf-8"))          # rendered by the csv module on the producing rank
                    else:
                        tsv.writerow(row)
            if raw is not None:
                raw.flush()
            self.fd.flush()
            if summary:
                summary.close()
        except BaseException as exc:       # surfaced by the caller after join()
            self.error = exc
