"""
CPU ORACLE -- TEST INFRASTRUCTURE ONLY (imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline / parity leg).

END-TO-END IDENTITY of the hot path: (HIP encoder -> HIP decoders) against (fp32 oracle encoder on the fp16-rounded weights -> scores
rounded to fp16 -> oracle/crf_oracle.c decoders) on the SAME chunks. This is the composite the per-stage tests cannot give: the decoders
are bit-exact on identical scores, but the fp16 engine's scores differ from the fp32 oracle's by a few 1e-3, so somewhere a near-tie of
the Viterbi recursion flips. The figures here count where (north_star: "Outputs match the reference CPU path's basecalls"; the reference
seam is `compute_scores`, /root/reference bonito/crf/basecall.py:27-45: `model(batch)` -> `beam_search` -> sequence / qstring / moves,
and CTC_CRF.viterbi, bonito/crf/model.py:98-108).

    oracle_outputs(model, x, ...)  -> dict of numpy arrays: what the CPU path calls for chunks x [n,1,L]
    compare(hip, ora)              -> the `parity` object of the bench line / the figures the -m gpu tests put floors under

Sequence identity = matches / alignment columns of an optimal global alignment under unit costs (Needleman-Wunsch) - the usual "read
accuracy" definition -, per chunk, then pooled over the chunks (sum of matches / sum of columns).
"""
import numpy as np


def alignment_identity(a, b):
    """(matches, columns) of an optimal global alignment of two byte strings / int8 arrays under unit costs (mismatch = insertion =
    deletion = 1; Needleman-Wunsch, rows vectorised with numpy, diagonal preferred in the traceback)."""
    a = np.frombuffer(a.encode() if isinstance(a, str) else bytes(a), np.uint8) if isinstance(a, (bytes, str)) else np.asarray(a).astype(np.uint8)
    b = np.frombuffer(b.encode() if isinstance(b, str) else bytes(b), np.uint8) if isinstance(b, (bytes, str)) else np.asarray(b).astype(np.uint8)
    la, lb = len(a), len(b)
    if la == lb and np.array_equal(a, b):
        return la, la
    if la == 0 or lb == 0:
        return 0, max(la, lb)
    ar = np.arange(lb + 1, dtype=np.int32)
    D = np.empty((la + 1, lb + 1), np.int32)
    D[0] = ar
    for i in range(1, la + 1):
        prev = D[i - 1]
        tmp = np.empty(lb + 1, np.int32)
        tmp[0] = i
        np.minimum(prev[1:] + 1, prev[:-1] + (b != a[i - 1]), out=tmp[1:])
        # D[i][j] = min over k <= j of tmp[k] + (j - k): the chain of insertions resolved by a running minimum
        D[i] = np.minimum.accumulate(tmp - ar) + ar
    i, j, matches, cols = la, lb, 0, 0
    while i > 0 or j > 0:
        cols += 1
        if i > 0 and j > 0 and D[i, j] == D[i - 1, j - 1] + (a[i - 1] != b[j - 1]):
            matches += int(a[i - 1] == b[j - 1])
            i, j = i - 1, j - 1
        elif i > 0 and D[i, j] == D[i - 1, j] + 1:
            i -= 1
        else:
            j -= 1
    return matches, cols


def _pooled_identity(rows_a, rows_b):
    m = c = 0
    for a, b in zip(rows_a, rows_b):
        mm, cc = alignment_identity(a, b)
        m += mm
        c += cc
    return (m / c) if c else 1.0, c


def _bases(plane):
    """[n,T] int8 with zeros where nothing is emitted -> list of the non-zero bytes per row."""
    return [row[row != 0] for row in np.asarray(plane)]


def _sliced(fn, sc, threads):
    """fn(scores slice) -> tuple of arrays with the chunk axis first, over `threads` slices of the chunks in parallel (the C oracle is
    single-threaded per call and ctypes drops the interpreter lock), concatenated back in order."""
    n = sc.shape[0]
    threads = max(1, min(int(threads), n))
    if threads == 1:
        return fn(sc)
    from concurrent.futures import ThreadPoolExecutor
    cuts = np.linspace(0, n, threads + 1).astype(int)
    with ThreadPoolExecutor(threads) as ex:
        parts = list(ex.map(lambda i: fn(sc[cuts[i]:cuts[i + 1]]), range(threads)))
    return tuple(np.concatenate([p[k] for p in parts]) for k in range(len(parts[0])))


def oracle_outputs(model, x, blank=2.0, decoders=("viterbi", "beam"), timers=None, fp16=False, threads=1):
    """The CPU path on chunks x [n,1,L] (torch CPU, values already fp16-representable): nn_ref.forward in fp32 on `model` (whose
    parameters the caller has rounded to fp16 with nn_ref.round_params_to_half_), scores rounded to fp16 in the engine's [n,T,4S]
    layout, then the C decoders (`threads` slices of the chunks in parallel). `timers`: optional dict that receives the seconds of
    each stage. `fp16`: nn_ref's fp16-STORAGE mode (every value rounded where the engine stores fp16) instead of the fp32 CPU path -
    the second oracle of the end-to-end comparison (what is left against it is summation order + the hardware exponential)."""
    import time

    import torch
    from oracle import crf_ref, nn_ref
    sl = model.seqdist.state_len
    t0 = time.perf_counter()
    with torch.no_grad():
        y = nn_ref.forward(model.encoder, x.float(), expand_blanks=False, fp16=fp16)
    sc = y.permute(1, 0, 2).contiguous().half().numpy()
    t1 = time.perf_counter()
    out = {"scores": sc, "state_len": np.int32(sl)}
    t2 = t1
    if "viterbi" in decoders:
        mv, path, best = _sliced(lambda z: crf_ref.viterbi(z, sl, blank=blank), sc, threads)
        out.update(vit_moves=mv, vit_path=path, vit_score=best)
        t2 = time.perf_counter()
    t3 = t2
    if "beam" in decoders:
        seq, qs, bmv, qf = _sliced(lambda z: crf_ref.beam_search(z, sl, blank=blank), sc, threads)
        out.update(beam_seq=seq, beam_qs=qs, beam_moves=bmv, beam_qf=qf)
        t3 = time.perf_counter()
    if timers is not None:
        timers["forward"] = timers.get("forward", 0.0) + (t1 - t0)
        timers["viterbi"] = timers.get("viterbi", 0.0) + (t2 - t1)
        timers["beam"] = timers.get("beam", 0.0) + (t3 - t2)
    return out


def bs2_vs_bs1(scores, state_len, bs2_seq, blank=2.0, threads=1):
    """QUALITY GUARD of the product decoder's definition (review, round 5: BS-1 -> BS-2 was a redefinition made for kernel speed, and
    bit-exactness against an oracle that moves with the kernel says nothing about what the change cost). On every chunk of koi-layout
    fp16 `scores` [n,T,4S]: the BS-1 answer (crf_ref.beam_search_bs1, the decoder of rounds 1-4) and the given BS-2 answer `bs2_seq`
    (int8 plane [n,T]) are BOTH scored with the model's exact sequence log-probability in fp64 (crf_ref.seq_logprob: sum over every
    alignment and start state, independent of any decoder). Returns a JSON-ready dict; `dlogp` = ln P(BS-2 sequence) - ln P(BS-1
    sequence) per chunk: >= 0 means the redefinition found an equally or more probable sequence."""
    from oracle import crf_ref
    sc = np.asarray(scores)
    s1, _ = _sliced(lambda z: crf_ref.beam_search_bs1(z, state_len, blank=blank), sc, threads)

    def lp(args):
        i, plane = args
        a, lz = crf_ref.seq_logprob(sc[i], state_len, plane, blank=blank)
        return a - lz

    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max(1, int(threads))) as ex:
        l2 = np.array(list(ex.map(lp, [(i, np.asarray(bs2_seq)[i]) for i in range(sc.shape[0])])))
        l1 = np.array(list(ex.map(lp, [(i, s1[i]) for i in range(sc.shape[0])])))
    d = l2 - l1
    same = int(sum(np.array_equal(a[a != 0], b[b != 0]) for a, b in zip(np.asarray(bs2_seq), s1)))
    return {"chunks": int(sc.shape[0]), "steps_per_chunk": int(sc.shape[1]), "sequences_identical": same,
            "dlogp_mean": float(d.mean()), "dlogp_min": float(d.min()), "dlogp_max": float(d.max()),
            "bs2_better": int((d > 1e-9).sum()), "bs2_worse": int((d < -1e-9).sum()),
            "logp_bs2_mean": float(l2.mean()), "logp_bs1_mean": float(l1.mean()),
            "definition": "ln P(seq | scores) in fp64 by the exact path sum (oracle_seq_logprob_f64), BS-2's sequence minus BS-1's, per chunk"}


def compare(hip, ora):
    """hip / ora: dicts with (any of) scores [n,T,C], vit_path, vit_moves, beam_seq, beam_moves, beam_qf, beam_qs (numpy). Returns the
    identity figures as plain floats (JSON-ready). Every fraction is 1.0 for identical outputs."""
    res = {"chunks": int(np.asarray(ora["scores"]).shape[0]), "steps_per_chunk": int(np.asarray(ora["scores"]).shape[1])}
    if "scores" in hip:
        d = np.abs(np.asarray(hip["scores"], np.float32) - np.asarray(ora["scores"], np.float32))
        res["scores_max_abs"] = float(d.max())
        res["scores_mean_abs"] = float(d.mean())
    if "vit_path" in hip and "vit_path" in ora:
        hp, op = np.asarray(hip["vit_path"]), np.asarray(ora["vit_path"])
        res["viterbi_path_identity"] = float((hp == op).mean())
        res["viterbi_moves_identity"] = float((np.asarray(hip["vit_moves"]) == np.asarray(ora["vit_moves"])).mean())
        res["viterbi_seq_identity"], res["viterbi_alignment_columns"] = _pooled_identity(_bases(hp), _bases(op))
        res["viterbi_chunks_bit_identical"] = int(sum(np.array_equal(a, b) for a, b in zip(hp, op)))
    if "beam_seq" in hip and "beam_seq" in ora:
        hs, os_ = np.asarray(hip["beam_seq"]), np.asarray(ora["beam_seq"])
        hm, om = np.asarray(hip["beam_moves"]), np.asarray(ora["beam_moves"])
        res["beam_seq_identity"], res["beam_alignment_columns"] = _pooled_identity(_bases(hs), _bases(os_))
        res["moves_identity"] = float((hm == om).mean())
        res["beam_chunks_bit_identical"] = int(sum(np.array_equal(a, b) and np.array_equal(c, d)
                                                   for a, b, c, d in zip(hs, os_, hm, om)))
        same = (hs == os_) & (hs != 0)                 # the same base emitted at the same step on both sides
        res["bases_matching_in_place"] = float(same.sum() / max(1, (os_ != 0).sum()))
        if "beam_qf" in hip and "beam_qf" in ora and same.any():
            dq = np.abs(np.asarray(hip["beam_qf"], np.float32) - np.asarray(ora["beam_qf"], np.float32))[same]
            res["q_max_abs_on_matching_bases"] = float(dq.max())
            res["q_mean_abs_on_matching_bases"] = float(dq.mean())
            res["q_frac_within_1e-3"] = float((dq <= 1e-3).mean())
        if "beam_qs" in hip and "beam_qs" in ora and same.any():
            res["qstring_identity_on_matching_bases"] = float((np.asarray(hip["beam_qs"])[same] == np.asarray(ora["beam_qs"])[same]).mean())
    return res


def hip_outputs(gmodel, x_dev, n, blank=2.0, decoder=None):
    """The PRODUCT path on the first n chunks of x_dev [N,1,L] (cuda fp16; N may be a whole engine call so that the timed kernel
    geometry is the one compared): gmodel(x_dev) -> bonito_amd.decode (bh_crf_viterbi, bh_beam_search through the C ABI).
    `decoder`: an optional bonito_amd.decode.CRFDecoder built for this call shape (the bench's own context) used for the beam planes."""
    from bonito_amd import decode
    scores = gmodel(x_dev)
    gmodel._hip.check()
    sub = scores[:n].contiguous()
    vm, vp = decode.viterbi(sub, blank_score=blank)
    seq, qs, mv, qf = decode.beam_search(sub, blank_score=blank, return_qfloat=True)
    out = {"scores": sub.cpu().numpy(), "vit_moves": vm.numpy(), "vit_path": vp.numpy(), "beam_seq": seq.numpy(),
           "beam_qs": qs.numpy(), "beam_moves": mv.numpy(), "beam_qf": qf.numpy()}
    if decoder is not None:                # the pre-allocated context must say the same as the one-shot call
        s2, q2, m2 = decoder.submit(scores).result()
        out["ctx_equal"] = bool(np.array_equal(s2.numpy()[:n], out["beam_seq"]) and np.array_equal(m2.numpy()[:n], out["beam_moves"])
                                and np.array_equal(q2.numpy()[:n], out["beam_qs"]))
    return out
