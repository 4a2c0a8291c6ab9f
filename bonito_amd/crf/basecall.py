"""
CRF basecalling pipeline on the MI355X engine: chunk -> batch -> HIP encoder + HIP decode -> unbatch ->
stitch -> format. Same function names, arguments and result dictionaries as /root/reference
bonito/crf/basecall.py (stitch_results 13-24, compute_scores 27-45, fmt 48-55, basecall 58-82) so
``load_symbol(config, "basecall")`` callers (cli/basecaller.py:71,131-136) need no change.
"""
import numpy as np
import torch

from bonito_amd import _lib, decode as hip_decode
from bonito_amd.decode import to_str
from bonito_amd.multiprocessing import thread_iter
from bonito_amd.util import chunk, stitch, batchify, unbatchify


def stitch_results(results, length, size, overlap, stride, reverse=False):
    """Stitch per-chunk decode outputs ([n_chunks, T] int8 each) back into one read."""
    if isinstance(results, dict):
        return {k: stitch_results(v, length, size, overlap, stride, reverse=reverse) for k, v in results.items()}
    if length < size:
        return results[0, :int(np.floor(length / stride))]
    return stitch(results, size, overlap, length, stride, reverse=reverse)


def stitch_planes(planes, length, size, overlap, stride, reverse=False):
    """`stitch_results` for the three decode outputs stacked as one int8 tensor [3, n_chunks, T] (sequence, qstring,
    moves): one set of slicing / concatenation calls per read instead of one per output. Same pieces as
    util.stitch (tests compare the two)."""
    if length < size:
        return planes[:, 0, :int(np.floor(length / stride))]
    if planes.shape[1] == 1:
        return planes[:, 0]
    semi = overlap // 2
    start, end = semi // stride, (size - semi) // stride
    stub = (length - overlap) % (size - overlap)
    first_end = (stub + semi) // stride if stub > 0 else end
    k = planes.shape[0]
    if reverse:
        return torch.cat([planes[:, -1, :-start], planes[:, 1:-1, -end:-start].flip(1).reshape(k, -1),
                          planes[:, 0, -first_end:]], dim=1)
    return torch.cat([planes[:, 0, :first_end], planes[:, 1:-1, start:end].reshape(k, -1), planes[:, -1, start:]], dim=1)


def fmt_planes(stride, planes, rna=False):
    """`fmt` for the stacked planes (sequence, qstring, moves) of one read: the two strings are compacted by the
    library's host helper (`bh_host_compact`, one pass each) instead of a numpy mask + compress + cast per string."""
    a = planes.numpy() if isinstance(planes, torch.Tensor) else np.asarray(planes)
    if a.dtype != np.int8 or not a.flags.c_contiguous:
        a = np.ascontiguousarray(a, dtype=np.int8)
    n = a.shape[1]
    buf = np.empty((2, n), np.uint8)
    src, dst = a.ctypes.data, buf.ctypes.data
    compact = _lib.lib().bh_host_compact
    n_seq, n_qs = compact(src, n, dst), compact(src + n, n, dst + n)
    seq, qs = buf[0, :n_seq].tobytes().decode("ascii"), buf[1, :n_qs].tobytes().decode("ascii")
    if rna:
        seq, qs = seq[::-1], qs[::-1]
    return {"stride": stride, "moves": a[2], "qstring": qs, "sequence": seq}


def compute_scores(model, batch, beam_width=32, beam_cut=100.0, scale=1.0, offset=0.0, blank_score=2.0,
                   reverse=False, decoder="beam"):
    """fp16 forward on the HIP engine followed by the HIP decoder; returns CPU int8 [N, T] tensors
    `sequence`, `qstring`, `moves` (zero where nothing is emitted), as koi.decode.beam_search does."""
    with torch.inference_mode():
        device = next(model.parameters()).device
        scores = model(batch.to(torch.float16).to(device))
        if reverse:
            scores = model.seqdist.reverse_complement(scores)
        with torch.cuda.device(scores.device):
            if decoder == "viterbi":
                moves, path = hip_decode.viterbi(scores, blank_score=blank_score)
                sequence = hip_decode.path_to_sequence(path)
                qstring = torch.where(sequence != 0, torch.tensor(33 + 20, dtype=torch.int8), torch.tensor(0, dtype=torch.int8))
            else:
                sequence, qstring, moves = hip_decode.beam_search(
                    scores, beam_width=beam_width, beam_cut=beam_cut, scale=scale, offset=offset,
                    blank_score=blank_score)
        eng = getattr(model, "_hip", None)
        if eng is not None:      # the decoders returned CPU tensors: the forward is complete, its timeout flag is on the host
            eng.poll()
        return {"moves": moves, "qstring": qstring, "sequence": sequence}


def fmt(stride, attrs, rna=False):
    flip = (lambda s: s[::-1]) if rna else (lambda s: s)
    return {
        "stride": stride,
        "moves": attrs["moves"].numpy(),
        "qstring": flip(to_str(attrs["qstring"])),
        "sequence": flip(to_str(attrs["sequence"])),
    }


class _Pipeline:
    """Two-stage device pipeline: the encoder runs on one HIP stream (thread 1), the CRF decode + int8 D2H
    on another (thread 2), so decode of batch i overlaps the encoder of batch i+1. The reference gets the
    same overlap from its per-stage ThreadIterators (bonito/crf/basecall.py:63-82) with koi returning CPU
    tensors; here the split is explicit because both halves are ours."""

    def __init__(self, model, decoder="beam", reverse=False, lanes=1, **decode_kw):
        self.model, self.mode, self.kw, self.reverse = model, decoder, decode_kw, reverse
        self.device = next(model.parameters()).device
        # `lanes` batches in flight in the encoder: lane k has its own engine replica (same weights, own workspace) and its own
        # stream, batches go round-robin. With the 8-bit recurrent kernels compiled for two workgroups per CU
        # (bh_set_option "lstm_q8_variant" 2) the persistent kernels of two lanes share every CU and each hides the other's
        # exchange round trip (hac-sized model: 18.8 -> 15.7 ms per batch); the fp16 kernels fill the register file and gain nothing.
        quantize = _resolved_quantize(model)
        self.lanes = max(1, min(int(lanes) or auto_lanes(model, quantize), max_lanes(model, quantize)))
        if self.lanes > 1 and quantize:
            hip_decode.set_option("lstm_q8_variant", 2)    # the 8-bit kernels compiled for two workgroups per CU
        self.enc_streams = [torch.cuda.Stream(self.device) for _ in range(self.lanes)]
        self.replicas = [None] * self.lanes            # lane 0 runs the model's own engine
        self.n_encoded = 0
        self.retries = 0                               # batches that were run a second time after an exchange timeout
        self.dec_stream = torch.cuda.Stream(self.device)
        self.copy_stream = torch.cuda.Stream(self.device)
        self.decoders = {}
        import threading
        self.enc_lock = threading.RLock()              # encoder thread vs. a serial retry from the decode thread

    def engine(self, lane):
        return getattr(self.model, "_hip", None) if lane == 0 else self.replicas[lane]

    def check_forward(self, lane, ticket):
        """Raises HipEngineError iff forward `ticket` of lane `lane`'s engine hit the spin bound of a persistent kernel. Only that
        forward's flag is read (`bh_encoder_error_flag_at`): the batches behind it in the pipeline keep theirs (advisor finding,
        round 3: with one sticky flag per engine, the retry of batch i erased the evidence against batches i+1 and i+2)."""
        eng = self.engine(lane)
        if eng is not None:
            eng.poll_ticket(ticket)

    def _retry_serially(self, dev_batch, lane=0):
        """An exchange timeout (a persistent recurrent kernel whose peers were not co-resident in time: a second process on the GPU,
        more lanes than the kernels can co-host) invalidates the batch, not the run: keep the encoder thread from issuing anything
        new (`enc_lock`), let what is in flight finish, and run encoder + decoder of THIS batch again on the engine that produced
        it, alone on the device; only a second failure is an error (bonito/crf/basecall.py:58-82 has nothing that can time out, so
        a drop-in must not abort where the reference would have carried on). The batches that were in flight meanwhile are checked
        against their own tickets when their turn comes, and retried the same way."""
        with self.enc_lock:
            torch.cuda.synchronize(self.device)
            with torch.inference_mode():
                scores = self._forward(lane, dev_batch)
                ticket = self.engine(lane).last_ticket
                if self.reverse:
                    scores = self.model.seqdist.reverse_complement(scores)
                torch.cuda.synchronize(self.device)
                self.check_forward(lane, ticket)          # raises HipEngineError if the serial run timed out as well
                key = tuple(scores.shape[1:])
                planes = self.decoders[key].submit(scores).result_planes()
        self.retries += 1
        return planes

    def _forward(self, lane, x):
        if lane == 0:
            return self.model(x)
        if self.replicas[lane] is None:
            self.replicas[lane] = self.model.engine_replica(x)
        return self.replicas[lane](x)

    def _encode_on(self, lane, enc_stream, x):
        """forward (+ reverse complement) of one device batch on its lane's stream -> (scores, ready event, ticket). The lock keeps
        a serial retry (decode thread) and this thread from driving an engine - one workspace, one flag array - at the same time."""
        with self.enc_lock:
            scores = self._forward(lane, x)
            ticket = self.engine(lane).last_ticket
            if self.reverse:
                scores = self.model.seqdist.reverse_complement(scores)
            ready = torch.cuda.Event()
            ready.record(enc_stream)
        return scores, ready, ticket

    def encode(self, batch):
        lane = self.n_encoded % self.lanes
        self.n_encoded += 1
        enc_stream = self.enc_streams[lane]
        if batch.is_cuda:                 # produced on the device (basecall_raw): no copy, just order the streams
            with torch.inference_mode(), torch.cuda.stream(enc_stream):
                enc_stream.wait_stream(torch.cuda.default_stream(self.device))
                batch.record_stream(enc_stream)
                scores, ready, ticket = self._encode_on(lane, enc_stream, batch)
            return scores, ready, batch, (lane, ticket)
        # H2D on its own stream, waited for on the host: the (recycled, pinned) batch buffer is free again when this
        # method returns, and the copy never queues behind the previous batch's encoder.
        with torch.inference_mode(), torch.cuda.stream(self.copy_stream):
            dev_batch = batch.to(torch.float16).to(self.device, non_blocking=True)
            copied = torch.cuda.Event()
            copied.record(self.copy_stream)
        copied.synchronize()
        with torch.inference_mode(), torch.cuda.stream(enc_stream):
            enc_stream.wait_event(copied)
            dev_batch.record_stream(enc_stream)
            scores, ready, ticket = self._encode_on(lane, enc_stream, dev_batch)
        return scores, ready, dev_batch, (lane, ticket)

    def decode(self, scores, ready, dev_batch=None, origin=None):
        key = tuple(scores.shape[1:])
        dec = self.decoders.get(key)
        if dec is None or dec.N < scores.shape[0]:
            cfg = getattr(self.model, "config", None) or {}
            nmax = max(scores.shape[0], int(cfg.get("basecaller", {}).get("batchsize", 0) or 0))
            dec = self.decoders[key] = hip_decode.CRFDecoder(nmax, key[0], key[1], self.device, mode=self.mode, **self.kw)
        with torch.inference_mode(), torch.cuda.stream(self.dec_stream):
            self.dec_stream.wait_event(ready)
            scores.record_stream(self.dec_stream)
            ticket = dec.submit(scores)
        planes = ticket.result_planes()      # [3, n, T] int8: sequence, qstring, moves
        # the decode outputs are on the host, so the encoder forward that produced `scores` (and the 4-byte copy of the
        # engine's timeout flag behind it) has completed: a spin timeout in a persistent kernel means these planes were
        # decoded from invalid scores -> never yield them (reference seam: crf/basecall.py:27-45): run the batch again with
        # nothing else in flight, raise only if that fails too
        lane, ticket = origin if origin is not None else (0, getattr(self.engine(0), "last_ticket", None))
        try:
            self.check_forward(lane, ticket)
        except _lib.HipEngineError:
            if dev_batch is None:
                raise
            planes = self._retry_serially(dev_batch, lane)
            self.engine(lane).ack(ticket)        # repaired: the engine-wide flag (poll / check) must not report this forward later
        if self.mode == "viterbi":
            path = planes[1]                 # plane 1 carries the path for the Viterbi decoder
            planes[0] = hip_decode.path_to_sequence(path)
            planes[1] = torch.where(planes[0] != 0, torch.tensor(33 + 20, dtype=torch.int8), torch.tensor(0, dtype=torch.int8))
        return planes


def _resolved_quantize(model):
    """What the engine will run: the value `use_koi` / `use_hip` stored (command line, config.toml or caller), not a flag."""
    q = getattr(model, "_quantize", None)
    if q is None:
        q = ((getattr(model, "config", None) or {}).get("basecaller") or {}).get("quantize")
    return bool(q)


def lstm_widths(model):
    return [m.rnn.hidden_size for m in model.modules() if hasattr(m, "rnn") and hasattr(m.rnn, "hidden_size")]


def q8_covers(h):
    """Hidden sizes the 8-bit recurrent kernel is instantiated for (`bh_k_lstm_q8_units` in csrc/lstm_q8.hip; tests compare)."""
    h = int(h)
    if h <= 0 or h > 512 or h % 16:
        return False
    nk8 = (h + 63) // 64
    return (h % 48 == 0 and nk8 in (2, 3, 5, 6)) or (h % 64 == 0 and nk8 in (1, 2, 4, 8))


def max_lanes(model, quantize=False):
    """Engine replicas whose recurrent kernels can be co-resident. The kernels of 192...1024-wide layers are persistent and hand h
    over between workgroups: every launch sizes its grid to the whole device and spins on peers, so two lanes would each end up
    partly resident and time out (advisor finding, round 2). The 8-bit kernel is of that kind at EVERY width it covers (at 96 and
    128 its rings span two workgroups), so `quantize` is looked at first (advisor finding, round 3: a quantised 96-wide model was
    given the three lanes of the fp16 ring-in-a-workgroup kernel); its instance built for two workgroups per CU (`lstm_q8_variant`
    2, 384 wide) allows two lanes. Otherwise: the ring-in-a-workgroup kernel of the narrow fp16 layers (64 / 96 / 128: no
    inter-workgroup exchange) any number of lanes, models without recurrent layers no limit, everything else one."""
    sizes = lstm_widths(model)
    if not sizes:
        return 1 << 30
    if quantize and any(q8_covers(h) for h in sizes):
        return 2 if all(h == 384 for h in sizes) else 1
    if all(h in (64, 96, 128) for h in sizes):
        return 1 << 30
    return 1


def auto_lanes(model, quantize=False):
    """`lanes = 0`: two lanes where the kernels are built to share the CUs (8-bit path at 384 hidden units), three for the narrow
    models whose ring-in-a-workgroup kernel leaves most CUs idle (bench.py, fast: 1 lane 8.9 ms per batch, 3 lanes x 4 batches per
    call 2.4), one everywhere else."""
    cap = max_lanes(model, quantize)
    if cap >= (1 << 20):
        return 3 if lstm_widths(model) else 1
    return 2 if quantize and cap >= 2 else 1


def batches_per_call(model, batchsize, quantize=False, chunksize=None, lanes=1):
    """`_batches_per_call_encoder` (what the recurrent kernels want), and for the 1024-state models at least calls of 512 chunks: their
    decode is ONE wave per chunk on a CU of its own - a latency chain - so two 256-chunk batches decode in the time of one (round 5,
    bench.py --per-call 2: decode 6.4 -> 4.1 ms per batch of the v5 transformer, 11.3 -> 7.1 of the 1024-wide LSTM; steps 64.1 -> 62.8 and
    101.3 -> 97.5 ms); bounded by 16 GiB of scores per call."""
    n = _batches_per_call_encoder(model, batchsize, quantize, chunksize, lanes)
    seqdist = getattr(model, "seqdist", None)
    if n == 1 and seqdist is not None and int(getattr(seqdist, "state_len", 0)) >= 5 and int(batchsize) <= 256:
        want = max(1, 512 // max(1, int(batchsize)))
        if chunksize:
            stride = max(1, int(getattr(model, "stride", 1) or 1))
            per_chunk = (int(chunksize) // stride + 1) * (4 ** (int(seqdist.state_len) + 1)) * 2
            want = min(want, max(1, (16 << 30) // max(1, per_chunk * int(batchsize))))
        n = max(1, min(2, want))
    return n


def _batches_per_call_encoder(model, batchsize, quantize=False, chunksize=None, lanes=1):
    """How many `batchsize`-chunk batches one ENGINE call should carry. The reference hands koi one batch per forward
    (crf/basecall.py:70-72) and `batchsize` keeps that meaning for the caller: chunks are independent, so results do not depend on
    how they are grouped (tests). For the 192...512-wide fp16 recurrent layers the engine's kernel carries two rings of 16 chunks
    per workgroup once a call holds more than one launch of single rings (32 rings = 512 chunks at 384 hidden units), which takes a
    512-chunk batch from 2.97 to 1.8 ms per layer, and a call of TWO such launches per layer keeps the two-stream pipeline full
    across call boundaries (hac, batches of 512: 14.9 ms per batch in calls of 1024 chunks, 13.96 in calls of 2048 - bench.py
    --per-call 2 / 4 on one box): calls of up to 2048 chunks there, as long as the score tensor of a call stays below 8 GiB; the
    8-bit path with its two lanes: calls of 2048 chunks as well (round 5: the decode stage then has eight chunks per CU, bench.py
    --quantize 12.28 -> 11.63 ms per batch; round 2 had settled on 1024-chunk calls, 14.6 -> 13.7); the narrow (64 / 96 / 128 wide) models with their three lanes: calls of 4096 chunks (round 5: 2.37 -> 2.28 ms per batch; 2048 before); one batch per call everywhere else."""
    sizes = lstm_widths(model)
    q8 = bool(quantize) and any(q8_covers(h) for h in sizes)
    if sizes and all(h in (64, 96, 128) for h in sizes) and not q8:   # ring-in-a-workgroup kernels: a 512-chunk batch fills an eighth of the chip
        return max(1, min(8, 4096 // max(1, int(batchsize)))) if lanes >= 2 else 1      # (4096 chunks = 256 rings = one workgroup per CU)
    if quantize:
        if lanes >= 2 and max_lanes(model, True) == 2:
            return max(1, min(4, 2048 // max(1, int(batchsize))))
        return 1
    if not sizes or not all(192 <= h <= 512 and (h % 48 == 0 or h % 64 == 0) for h in sizes):
        return 1
    wpr = max(1, (max(sizes) // (12 if max(sizes) % 48 == 0 else 16)) // 4)        # workgroups per ring
    rings_per_launch = max(1, 256 // (8 * wpr)) * 8                                   # single rings on 256 CUs
    target = 2 * 2 * rings_per_launch * 16                                            # chunks of two paired launches
    if chunksize:
        seqdist, stride = getattr(model, "seqdist", None), max(1, int(getattr(model, "stride", 1) or 1))
        states = len(getattr(seqdist, "alphabet", "NACGT")) - 1 if seqdist is not None else 4
        n_scores = (states ** int(getattr(seqdist, "state_len", 4))) * states if seqdist is not None else 1024
        per_chunk = (int(chunksize) // stride + 1) * n_scores * 2                     # bytes of fp16 scores per chunk
        target = min(target, max(int(batchsize), (8 << 30) // max(1, per_chunk)))
    return max(1, min(8, target // max(1, int(batchsize))))


def chunk_batches(reads, chunksize, overlap, batchsize, pin=False, nbuf=4):
    """Fused `chunk` + `batchify` + fp32->fp16 for the product path: yields exactly the (keys, batch) pairs of
    ``batchify(((read, 0, T), chunk(signal, chunksize, overlap)) for read in reads)`` (tests compare the two), but
    every chunk row is converted and written ONCE, straight from the read's signal into a reusable (optionally
    pinned) fp16 batch buffer. The generic pair costs three passes over the samples (window copy, concat, cast) and
    bounded the pipeline at ~1.5e8 samples/s on one host thread.

    A yielded batch stays valid until `nbuf - 1` further batches have been yielded (buffers are recycled)."""
    def new_buf():
        b = torch.empty((batchsize, 1, chunksize), dtype=torch.float16)
        return b.pin_memory() if pin else b

    bufs = [new_buf() for _ in range(nbuf)]
    buf_ptrs = [b.data_ptr() for b in bufs]
    gather = _lib.lib().bh_host_chunk_rows
    cur, pos, keys = 0, 0, []
    for read in reads:
        sig = read.signal
        if (isinstance(sig, np.ndarray) and sig.ndim == 1 and sig.dtype == np.float32 and sig.flags.c_contiguous
                and 0 <= overlap < chunksize <= sig.shape[0]):
            # the common case, in the library (one call per run of rows; ctypes drops the interpreter lock meanwhile):
            # same rows as util.chunk, same round-to-nearest-even cast as .to(float16)
            T = sig.shape[0]
            key = (read, 0, T)
            step = chunksize - overlap
            stub = (T - overlap) % step
            n_total = (T - stub - chunksize) // step + 1 + (1 if stub > 0 else 0)
            src, lo = sig.ctypes.data, 0
            while lo < n_total:
                take = min(n_total - lo, batchsize - pos)
                if gather(src, T, chunksize, overlap, lo, take, buf_ptrs[cur] + pos * chunksize * 2) != take:
                    raise RuntimeError("bh_host_chunk_rows failed")
                if keys and keys[-1][0] is key and keys[-1][1][1] == pos:
                    keys[-1] = (key, (keys[-1][1][0], pos + take))
                else:
                    keys.append((key, (pos, pos + take)))
                pos += take
                lo += take
                if pos == batchsize:
                    yield tuple(keys), bufs[cur]
                    cur, pos, keys = (cur + 1) % nbuf, 0, []
            continue
        sig = torch.from_numpy(sig) if isinstance(sig, np.ndarray) else sig
        T = sig.shape[-1]
        key = (read, 0, T)
        if sig.ndim != 1 or chunksize == 0 or T < chunksize:
            rows = chunk(sig, chunksize, overlap)                       # rare shapes: generic code, then copy in
            rows = rows.reshape(rows.shape[0], -1)
            if rows.shape[-1] != chunksize:
                raise ValueError("chunk_batches needs fixed-size chunks (chunksize=%d, got %d)" % (chunksize, rows.shape[-1]))
        else:
            step = chunksize - overlap
            stub = (T - overlap) % step
            rows = sig[stub:].unfold(0, chunksize, step)               # [n, chunksize] overlapping view, no copy
            if stub > 0:
                rows = (sig[:chunksize].unsqueeze(0), rows)
        pieces = rows if isinstance(rows, tuple) else (rows,)
        n_total = sum(p.shape[0] for p in pieces)
        done = 0
        for piece in pieces:
            lo = 0
            while lo < piece.shape[0]:
                take = min(piece.shape[0] - lo, batchsize - pos)
                bufs[cur][pos:pos + take, 0].copy_(piece[lo:lo + take])  # strided gather + cast in one pass
                # key ranges follow batchify: one (key, (lo, hi)) entry per contiguous run of a read inside a batch
                if keys and keys[-1][0] is key and keys[-1][1][1] == pos:
                    keys[-1] = (key, (keys[-1][1][0], pos + take))
                else:
                    keys.append((key, (pos, pos + take)))
                pos += take
                lo += take
                done += take
                if pos == batchsize:
                    yield tuple(keys), bufs[cur]
                    cur, pos, keys = (cur + 1) % nbuf, 0, []
        assert done == n_total
    if pos:
        yield tuple(keys), bufs[cur][:pos]


def basecall(model, reads, chunksize=4000, overlap=100, batchsize=32, reverse=False, rna=False, decoder="beam", lanes=0,
             per_call=0):
    """Basecalls a set of reads: yields (read, {sequence, qstring, moves, stride}). `lanes`: batches in flight in the encoder
    (engine replicas; 0 = automatic: two for the 8-bit recurrent path at 384 hidden units, one otherwise); `per_call`: batches of `batchsize` chunks per engine call (0 = `batches_per_call`); results are
    identical for any value of either."""
    pipe = _Pipeline(model, decoder=decoder, reverse=reverse, lanes=lanes)
    if not per_call:
        per_call = batches_per_call(model, batchsize, _resolved_quantize(model), chunksize, pipe.lanes)
    batchsize = int(batchsize) * max(1, int(per_call))
    # up to 4 batches are in flight behind the generator (three single-slot queues + the consumer): recycle after 8
    batches = thread_iter(chunk_batches(reads, chunksize, overlap, batchsize, pin=torch.cuda.is_available(), nbuf=8))
    encoded = thread_iter((keys, pipe.encode(batch)) for keys, batch in batches)
    scores = thread_iter((keys, pipe.decode(*enc)) for keys, enc in encoded)
    # Stitching and formatting run in the consumer's thread (the CLI's Writer thread): they cost ~40 us per read now, and
    # every additional Python-heavy thread slows the others through the interpreter lock more than it adds (host-only
    # ceiling of this pipeline, device stages stubbed: 1.6e8 samples/s with two more threads, 2.0e8 without).
    return (
        (read, fmt_planes(model.stride, stitch_planes(sc, end - start, chunksize, overlap, model.stride, reverse), rna))
        for ((read, start, end), sc) in unbatchify(scores, dim=1)
    )


_MODES = {"fastq": 0, "fasta": 1, "sam": 2}


def records_from_planes(batches, chunksize, overlap, stride, mode, min_qscore=0.0, reverse=False, rna=False):
    """(keys, planes [3, n, T] int8) batches -> the (text, summary_row, log) triples `io.format_record` produces for the same reads
    (tests compare the bytes), one library call per read (`bh_host_format_read`: stitch + to_str + record text) instead of
    unbatchify -> stitch_planes -> fmt_planes -> format_record, which cost ~1 ms of interpreter and small-tensor time per read and
    capped a rank at ~1e8 samples/s on one host thread. A read's chunks are passed as runs of rows of the batches they sit in
    (nothing is concatenated); the batches must arrive in order, like for `unbatchify`."""
    import ctypes as C
    from bonito_amd.io import signal_samples, summary_row
    fn = _lib.lib().bh_host_format_read
    m = _MODES[mode]
    cap = 1 << 20
    out = C.create_string_buffer(cap)
    seq_len, mean_q = C.c_long(0), C.c_double(0.0)
    K = 64
    base, pstride, lo_a, rows_a = (C.c_void_p * K)(), (C.c_long * K)(), (C.c_long * K)(), (C.c_long * K)()

    def emit(key, pieces):
        nonlocal cap, out, K, base, pstride, lo_a, rows_a
        read, start, end = key
        n = len(pieces)
        if n > K:            # a very long read / very small engine calls: as many pieces as it takes (the library has no limit either)
            K = 2 * n
            base, pstride, lo_a, rows_a = (C.c_void_p * K)(), (C.c_long * K)(), (C.c_long * K)(), (C.c_long * K)()
        for i, (arr, lo, hi) in enumerate(pieces):
            base[i], pstride[i], lo_a[i], rows_a[i] = arr.ctypes.data, arr.strides[0], lo, hi - lo
        T = pieces[0][0].shape[2]
        rid = read.read_id.encode()
        run = (getattr(read, "run_id", None) or "").encode()
        while True:
            got = fn(base, pstride, lo_a, rows_a, n, T, end - start, chunksize, overlap, stride, int(bool(reverse)), int(bool(rna)), m,
                     float(min_qscore), rid, run, int(getattr(read, "num_samples", end - start)),
                     int(getattr(read, "trimmed_samples", 0)), out, cap, C.byref(seq_len), C.byref(mean_q))
            if got >= -1:
                break
            cap = max(2 * cap, -got)
            out = C.create_string_buffer(cap)
        if got < 0:
            raise RuntimeError("bh_host_format_read failed")
        log = (read.read_id, signal_samples(read))
        if got == 0:
            return None, None, log
        return C.string_at(out, got).decode("utf-8"), summary_row(read, seq_len.value, mean_q.value), log

    cur, pieces = None, []
    for keys, planes in batches:
        arr = planes.numpy() if isinstance(planes, torch.Tensor) else np.asarray(planes)
        if arr.dtype != np.int8 or arr.strides[2] != 1 or arr.strides[1] != arr.shape[2]:
            arr = np.ascontiguousarray(arr, dtype=np.int8)
        for key, (lo, hi) in keys:
            if cur is not None and key[0] is not cur[0]:
                yield emit(cur, pieces)
                pieces = []
            cur = key
            pieces.append((arr, lo, hi))
    if cur is not None:
        yield emit(cur, pieces)


def basecall_records(model, reads, mode, chunksize=4000, overlap=100, batchsize=32, reverse=False, rna=False, decoder="beam",
                     lanes=0, per_call=0, min_qscore=0.0, raw=None):
    """`basecall` (or, with `raw` = the keyword arguments of the device-side ingest, `basecall_raw`) + `io.format_record` for the
    writers of the product path: yields the (text, summary_row, log) triples of the reads in order - the same bytes (tests), with
    the per-read host work in the library (`records_from_planes`)."""
    pipe = _Pipeline(model, decoder=decoder, reverse=reverse, lanes=lanes)
    if not per_call:
        per_call = batches_per_call(model, batchsize, _resolved_quantize(model), chunksize, pipe.lanes)
    batchsize = int(batchsize) * max(1, int(per_call))
    if raw is not None:
        batches = thread_iter(raw_chunk_batches(reads, chunksize, overlap, batchsize, next(model.parameters()).device, **raw))
    else:
        batches = thread_iter(chunk_batches(reads, chunksize, overlap, batchsize, pin=torch.cuda.is_available(), nbuf=8))
    encoded = thread_iter((keys, pipe.encode(batch)) for keys, batch in batches)
    scores = thread_iter((keys, pipe.decode(*enc)) for keys, enc in encoded)
    return records_from_planes(scores, chunksize, overlap, model.stride, mode, min_qscore, reverse, rna)


def raw_chunk_batches(reads, chunksize, overlap, batchsize, device, group_samples=1 << 26, scaling_strategy=None,
                      norm_params=None, do_trim=True):
    """Device-side ingest for raw reads (objects with int16 `.raw`, `.scaling`, `.offset`): groups of reads are shipped as
    int16 once, normalised / trimmed / chunked on the GPU (bonito_amd.signal) and handed to the encoder as fp16 device
    batches. Yields the same (keys, batch) stream as `chunk_batches` over `reader.Read` objects of the same reads, and
    fills in `shift`, `scale`, `trimmed_samples` and `signal_len` on every read (what the writers report)."""
    from bonito_amd.signal import RawBatch

    def groups():
        cur, total = [], 0
        for read in reads:
            cur.append(read)
            total += len(read.raw)
            if total >= group_samples:
                yield cur
                cur, total = [], 0
        if cur:
            yield cur

    pend_keys, pend_parts, pos = [], [], 0
    for group in groups():
        rb = RawBatch([r.raw for r in group], [float(r.scaling) for r in group], [float(r.offset) for r in group], device=device)
        shift, scale, trim = rb.normalise(scaling_strategy, norm_params, do_trim)
        for r, sh, sc, tr in zip(group, shift, scale, trim):
            r.shift, r.scale, r.trimmed_samples = float(sh), float(sc), int(tr)
            r.signal_len = len(r.raw) - int(tr)
        table = rb.chunk_table(chunksize, overlap, trim)
        n = len(table[0])
        # contiguous runs of one read inside the chunk list
        lo = 0
        while lo < n:
            take = min(n - lo, batchsize - pos)
            part = rb.chunks(table, chunksize, lo, lo + take)
            ridx = table[0][lo:lo + take]
            start = 0
            while start < take:
                end = start
                while end < take and ridx[end] == ridx[start]:
                    end += 1
                read = group[int(ridx[start])]
                key = (read, 0, read.signal_len)
                if pend_keys and pend_keys[-1][0][0] is read and pend_keys[-1][1][1] == pos + start:
                    pend_keys[-1] = (key, (pend_keys[-1][1][0], pos + end))
                else:
                    pend_keys.append((key, (pos + start, pos + end)))
                start = end
            pend_parts.append(part)
            pos += take
            lo += take
            if pos == batchsize:
                yield tuple(pend_keys), (pend_parts[0] if len(pend_parts) == 1 else torch.cat(pend_parts))
                pend_keys, pend_parts, pos = [], [], 0
    if pos:
        yield tuple(pend_keys), (pend_parts[0] if len(pend_parts) == 1 else torch.cat(pend_parts))


def basecall_raw(model, reads, chunksize=4000, overlap=100, batchsize=32, reverse=False, rna=False, decoder="beam",
                 scaling_strategy=None, norm_params=None, do_trim=True, lanes=0, per_call=0):
    """`basecall` for raw int16 reads (`.raw`, `.scaling`, `.offset`): the signal pre-processing of reader.Read runs on the
    device. Same results as ``basecall(model, [reader.Read(...) ...])`` on the same reads (tests compare them)."""
    pipe = _Pipeline(model, decoder=decoder, reverse=reverse, lanes=lanes)
    if not per_call:
        per_call = batches_per_call(model, batchsize, _resolved_quantize(model), chunksize, pipe.lanes)
    batchsize = int(batchsize) * max(1, int(per_call))
    device = next(model.parameters()).device
    batches = thread_iter(raw_chunk_batches(reads, chunksize, overlap, batchsize, device, scaling_strategy=scaling_strategy,
                                            norm_params=norm_params, do_trim=do_trim))
    encoded = thread_iter((keys, pipe.encode(batch)) for keys, batch in batches)
    scores = thread_iter((keys, pipe.decode(*enc)) for keys, enc in encoded)
    # Stitching and formatting run in the consumer's thread (the CLI's Writer thread): they cost ~40 us per read now, and
    # every additional Python-heavy thread slows the others through the interpreter lock more than it adds (host-only
    # ceiling of this pipeline, device stages stubbed: 1.6e8 samples/s with two more threads, 2.0e8 without).
    return (
        (read, fmt_planes(model.stride, stitch_planes(sc, end - start, chunksize, overlap, model.stride, reverse), rna))
        for ((read, start, end), sc) in unbatchify(scores, dim=1)
    )
