"""GPU CRF Viterbi vs the C oracle: bit-exact paths, move tables and path scores (-m gpu)."""
import numpy as np
import pytest
import torch

from bonito_amd import decode
from conftest import assert_qstrings_agree
from oracle import crf_ref

pytestmark = pytest.mark.gpu


def _scores(rng, N, T, C, kind):
    x = rng.standard_normal((N, T, C)) * 2.0
    if kind == "tanh":
        x = np.tanh(x) * 5.0
    elif kind == "ties":           # heavy quantisation -> many exact ties exercise the tie-break rule
        x = np.round(x)
    return np.clip(x, -5, 5).astype(np.float16)


@pytest.mark.parametrize("state_len", [1, 2, 3, 4, 5])
@pytest.mark.parametrize("kind", ["normal", "tanh", "ties"])
def test_viterbi_koi_layout_exact(state_len, kind):
    rng = np.random.default_rng(state_len * 7 + len(kind))
    N, T = (5, 77) if state_len == 5 else (19, 203)
    sc = _scores(rng, N, T, 4 ** (state_len + 1), kind)
    moves, path, best = decode.viterbi(torch.from_numpy(sc).cuda(), blank_score=2.0, return_score=True)
    om, op, ob = crf_ref.viterbi(sc, state_len, layout_5s=False, blank=2.0)
    assert np.array_equal(path.numpy(), op)
    assert np.array_equal(moves.numpy(), om)
    assert np.array_equal(best.numpy(), ob)        # same fp32 left-fold -> identical bits


@pytest.mark.parametrize("state_len", [2, 3])
def test_viterbi_expanded_layout_exact(state_len):
    rng = np.random.default_rng(3)
    T, N = 150, 6
    sc = np.clip(rng.standard_normal((T, N, 5 * 4 ** state_len)) * 2, -5, 5).astype(np.float16)
    moves, path = decode.viterbi_5s(torch.from_numpy(sc).cuda(), state_len)
    om, op, _ = crf_ref.viterbi(sc, state_len, layout_5s=True, time_major=True)
    assert np.array_equal(path.numpy(), op) and np.array_equal(moves.numpy(), om)


@pytest.mark.parametrize("state_len", [3, 4, 5])
def test_viterbi_quad_kernel_equals_the_round_1_kernel(state_len):
    """Round 5: `crf_viterbi_quad_kernel` (four states per thread, one wave per 256-state chunk, 16-bit back-pointer words) against the
    round-1 kernel (`viterbi_quad` 0: one thread per state) and the C oracle - moves, path and best score, with tie-heavy scores, a chunk
    count that leaves the last workgroup ragged (16 / 4 / 1 chunks per workgroup), a step count that is no multiple of the prefetch
    depth or of the traceback block; then through the C ABI with padded rows (row stride 2 C: still the new kernel) and with a base
    pointer 8 bytes off a 16-byte boundary (falls back to the round-1 kernel): the same bytes."""
    import ctypes as C
    from bonito_amd import _lib
    S = 4 ** state_len
    Cc = 4 * S
    N, T = {3: 37, 4: 13, 5: 3}[state_len], 131
    rng = np.random.default_rng(300 + state_len)
    sc = _scores(rng, N, T, Cc, "ties")
    dev = torch.from_numpy(sc).cuda()
    m1, p1, b1 = decode.viterbi(dev, return_score=True)
    decode.set_option("viterbi_quad", 0)
    try:
        m0, p0, b0 = decode.viterbi(dev, return_score=True)
    finally:
        decode.set_option("viterbi_quad", 1)
    assert torch.equal(m1, m0) and torch.equal(p1, p0) and torch.equal(b1, b0)
    om, op, ob = crf_ref.viterbi(sc, state_len)
    assert np.array_equal(p1.numpy(), op) and np.array_equal(m1.numpy(), om) and np.array_equal(b1.numpy(), ob)

    lib = _lib.lib()

    def run(ptr, s_n, s_t):
        ws = torch.empty(lib.bh_crf_viterbi_workspace(N, T, state_len), dtype=torch.uint8, device="cuda")
        mo = torch.empty((N, T), dtype=torch.int8, device="cuda")
        pa = torch.empty((N, T), dtype=torch.int8, device="cuda")
        be = torch.empty((N,), dtype=torch.float32, device="cuda")
        _lib.check(lib.bh_crf_viterbi(ptr, N, T, state_len, 0, 2.0, s_n, s_t, _lib.ptr(ws), _lib.ptr(mo), _lib.ptr(pa), _lib.ptr(be),
                                      _lib.stream_ptr("cuda:0")), "bh_crf_viterbi")
        torch.cuda.synchronize()
        return mo.cpu(), pa.cpu(), be.cpu()

    padded = torch.full((N, T, 2 * Cc), 9.0, dtype=torch.float16, device="cuda")
    padded[:, :, :Cc] = dev
    mp, pp, bpad = run(_lib.ptr(padded), T * 2 * Cc, 2 * Cc)
    assert torch.equal(mp, m1) and torch.equal(pp, p1) and torch.equal(bpad, b1)
    flat = torch.zeros(N * T * Cc + 8, dtype=torch.float16, device="cuda")
    flat[4:4 + N * T * Cc] = dev.reshape(-1)
    mu, pu, bu = run(C.c_void_p(flat.data_ptr() + 8), T * Cc, Cc)
    assert torch.equal(mu, m1) and torch.equal(pu, p1) and torch.equal(bu, b1)


def test_viterbi_edge_shapes():
    rng = np.random.default_rng(4)
    for N, T in [(1, 1), (1, 7), (3, 8), (2, 9), (1, 513)]:
        sc = _scores(rng, N, T, 256, "normal")
        moves, path = decode.viterbi(torch.from_numpy(sc).cuda())
        om, op, _ = crf_ref.viterbi(sc, 3)
        assert np.array_equal(path.numpy(), op) and np.array_equal(moves.numpy(), om)


def test_viterbi_full_size_properties():
    """BASELINE shape (hac: N=512, T=1667, C=1024): checks size-independent properties and an exact
    comparison on a random subset of chunks (the C oracle does ~1 chunk/10 ms)."""
    g = torch.Generator(device="cuda").manual_seed(0)
    sc = (torch.randn(512, 1667, 1024, generator=g, device="cuda") * 2).clamp(-5, 5).half()
    moves, path, best = decode.viterbi(sc, return_score=True)
    p, m = path.numpy(), moves.numpy()
    assert ((p > 0) == (m == 1)).all() and p.max() <= 4 and p.min() >= 0
    idx = [0, 17, 255, 511]
    om, op, ob = crf_ref.viterbi(sc[idx].cpu().numpy(), 4)
    assert np.array_equal(p[idx], op) and np.array_equal(best.numpy()[idx], ob)
    # decoding is per-chunk: permuting the batch permutes the result
    perm = torch.randperm(512)
    m2, p2 = decode.viterbi(sc[perm.cuda()].contiguous())
    assert np.array_equal(p2.numpy(), p[perm.numpy()])


def test_decode_rejects_wrong_dtype_and_device():
    with pytest.raises(TypeError):
        decode.viterbi(torch.zeros(1, 4, 64, device="cuda"))
    with pytest.raises(Exception):
        decode.viterbi(torch.zeros(1, 4, 64, dtype=torch.float16))


# ---- beam search (BS-1) vs the C oracle ---------------------------------------------------------
def _peaky_scores(rng, N, T, state_len, sharp=3.0):
    """Scores with a planted path so the decoder emits realistic base runs."""
    S = 4 ** state_len
    x = rng.standard_normal((N, T, 4 * S)).astype(np.float32)
    for n in range(N):
        st = int(rng.integers(S))
        for t in range(T):
            if rng.random() < 0.35:
                b = int(rng.integers(4))
                new = ((st << 2) | b) & (S - 1)
                x[n, t, new * 4 + (st >> (2 * (state_len - 1)))] += 2.0 * sharp
                st = new
            else:
                x[n, t] -= sharp * 0.5
    return np.clip(x, -5, 5).astype(np.float16)


@pytest.mark.parametrize("state_len", [1, 2, 3, 4, 5])
def test_beam_search_matches_oracle(state_len):
    rng = np.random.default_rng(40 + state_len)
    N, T = (3, 60) if state_len == 5 else (7, 150)
    sc = _peaky_scores(rng, N, T, state_len)
    seq, qs, mv, qf = decode.beam_search(torch.from_numpy(sc).cuda(), return_qfloat=True)
    oseq, oqs, omv, oqf = crf_ref.beam_search(sc, state_len)
    assert np.array_equal(mv.numpy(), omv)
    assert np.array_equal(seq.numpy(), oseq)
    assert np.abs(qf.numpy() - oqf).max() < 1e-3
    assert_qstrings_agree(qs.numpy(), oqs, oqf)       # a byte may differ only at a rounding boundary of q, by one, rarely
    assert set(np.unique(seq.numpy())) <= {0, 65, 67, 71, 84}


@pytest.mark.parametrize("state_len", [1, 2, 3, 4, 5])
def test_beam_search_edge_shapes_match_oracle(state_len):
    """Ragged ends of the launch geometry: one and two time steps (shorter than a staging block), one chunk, and chunk counts on both sides
    of what a wave / a workgroup packs (64 / 16 / 4 chunks per wave below 256 states, four chunks per workgroup above) - bit-exact sequence
    and moves, q within tolerance, both decoders."""
    rng = np.random.default_rng(900 + state_len)
    S = 4 ** state_len
    shapes = [(1, 1), (1, 2), (2, 3), (5, 5), (17, 9), (65, 4)] if state_len < 5 else [(1, 1), (1, 2), (3, 5), (5, 9)]
    for N, T in shapes:
        sc = np.clip(rng.standard_normal((N, T, 4 * S)) * 2.0, -5, 5).astype(np.float16)
        seq, qs, mv, qf = decode.beam_search(torch.from_numpy(sc).cuda(), return_qfloat=True)
        oseq, oqs, omv, oqf = crf_ref.beam_search(sc, state_len)
        assert np.array_equal(mv.numpy(), omv) and np.array_equal(seq.numpy(), oseq), (N, T)
        assert np.abs(qf.numpy() - oqf).max() < 1e-3, (N, T)
        m, pth = decode.viterbi(torch.from_numpy(sc).cuda())
        om, op, _ = crf_ref.viterbi(sc, state_len)
        assert np.array_equal(pth.numpy(), op) and np.array_equal(m.numpy(), om), (N, T)


@pytest.mark.parametrize("kind", ["normal", "ties"])
def test_beam_search_random_scores_and_params(kind):
    rng = np.random.default_rng(77)
    sc = _scores(rng, 5, 120, 256, kind)
    for bw, cut, blank in [(32, 100.0, 2.0), (8, 10.0, 0.5), (1, 1.0, 2.0), (16, 1e6, -1.0)]:
        seq, qs, mv, qf = decode.beam_search(torch.from_numpy(sc).cuda(), beam_width=bw, beam_cut=cut,
                                             blank_score=blank, scale=1.05, offset=0.2, return_qfloat=True)
        oseq, oqs, omv, oqf = crf_ref.beam_search(sc, 3, beam_width=bw, beam_cut=cut, blank=blank, scale=1.05, offset=0.2)
        assert np.array_equal(mv.numpy(), omv), (bw, cut, blank)
        assert np.array_equal(seq.numpy(), oseq)
        assert np.abs(qf.numpy() - oqf).max() < 1e-3


@pytest.mark.parametrize("state_len", [2, 4, 5])
def test_beam_search_out_of_range_scores_stay_bit_exact(state_len):
    """BS-2's guide lives in the linear domain: scores far outside the +-5 a trained head emits - up to the fp16 maximum, and a blank
    score of +-30 - must neither overflow nor diverge from the oracle (the exponential clamps its argument to +-40 on both sides, every
    recurrence sum carries a 2^-60 floor, the rescaling is exact): sequence and moves stay bit-identical. The q-scores come from an fp32
    scan: with weights spread over e^+-40 a posterior mass of 2e-5 carries the fp32 resolution of the total (1e-7 absolute), i.e. the
    ERROR PROBABILITIES agree to a few 1e-6 absolute and q to 0.05 up there (within +-5, the range of a trained head, q agrees to 1e-3: the tests
    above)."""
    rng = np.random.default_rng(900 + state_len)
    N, T = (3, 40) if state_len == 5 else (6, 90)
    S = 4 ** state_len
    sc = (rng.normal(0, 30, (N, T, 4 * S))).astype(np.float16)
    sc[rng.random(sc.shape) < 0.01] = np.float16(65504)
    sc[rng.random(sc.shape) < 0.01] = np.float16(-65504)
    sc[0, :, :] = np.float16(-60000)                     # a chunk in which every move is (numerically) impossible
    sc[1, : T // 2, :] = np.float16(60000)               # ... and one in which every move saturates
    for blank in (2.0, -30.0, 30.0):
        seq, qs, mv, qf = decode.beam_search(torch.from_numpy(sc).cuda(), blank_score=blank, return_qfloat=True)
        oseq, oqs, omv, oqf = crf_ref.beam_search(sc, state_len, blank=blank)
        assert np.array_equal(mv.numpy(), omv) and np.array_equal(seq.numpy(), oseq), (state_len, blank)
        assert np.isfinite(qf.numpy()).all() and np.abs(qf.numpy() - oqf).max() < 0.05
        assert np.abs(10.0 ** (-qf.numpy() / 10.0) - 10.0 ** (-oqf / 10.0))[omv != 0].max() < 5e-6


def test_beam_search_close_to_viterbi_on_confident_scores():
    rng = np.random.default_rng(5)
    sc = _peaky_scores(rng, 6, 300, 3, sharp=4.0)
    seq, qs, mv = decode.beam_search(torch.from_numpy(sc).cuda())
    m2, p2 = decode.viterbi(torch.from_numpy(sc).cuda())
    assert (mv.numpy() == m2.numpy()).mean() > 0.97
    q = qs.numpy()[mv.numpy() == 1]
    assert q.min() >= 33 + 1 and q.max() <= 33 + 50 and np.median(q) > 33 + 10


def test_beam_search_full_size():
    """hac BASELINE shape: properties + exact agreement with the oracle on a few chunks."""
    g = torch.Generator(device="cuda").manual_seed(1)
    sc = (torch.randn(512, 1667, 1024, generator=g, device="cuda") * 2.5).clamp(-5, 5).half()
    seq, qs, mv = decode.beam_search(sc)
    s, m, q = seq.numpy(), mv.numpy(), qs.numpy()
    assert ((s != 0) == (m == 1)).all() and ((q != 0) == (m == 1)).all()
    idx = [3, 200, 511]
    oseq, oqs, omv, oqf = crf_ref.beam_search(sc[idx].cpu().numpy(), 4)
    assert np.array_equal(s[idx], oseq) and np.array_equal(m[idx], omv)
    assert_qstrings_agree(q[idx], oqs, oqf)


# ---- reverse_complement / logZ ----------------------------------------------------------------------
def test_reverse_complement_kernel_matches_reference_fixture_and_oracle():
    import os
    from conftest import GOLDEN
    z = np.load(os.path.join(GOLDEN, "crf_rc.npz"))
    for sl in (1, 2, 3):
        x = torch.from_numpy(z["x%d" % sl]).half()
        got = decode.reverse_complement_5s(x.cuda(), sl).cpu().float().numpy()
        assert np.array_equal(got, z["y%d" % sl]), sl                 # reference output (pure permutation -> exact)
    rng = np.random.default_rng(2)
    sc = _scores(rng, 3, 50, 1024, "normal")
    got = decode.reverse_complement(torch.from_numpy(sc).cuda()).cpu().numpy()
    assert np.array_equal(got, crf_ref.reverse_complement(sc, 4, layout_5s=False))


def test_logz_matches_oracle():
    rng = np.random.default_rng(6)
    for sl in (2, 3, 4):
        sc = _scores(rng, 4, 90, 4 ** (sl + 1), "normal")
        got = decode.logz(torch.from_numpy(sc).cuda()).numpy()
        _, _, lz = crf_ref.backward(sc, sl)
        assert np.allclose(got, lz, rtol=0, atol=1e-6 * np.abs(lz).max() + 1e-4)
        assert np.allclose(got, crf_ref.logz(sc, sl), rtol=0, atol=5e-3)


@pytest.mark.parametrize("state_len", [1, 3, 4, 5])
def test_posterior_viterbi_matches_oracle(state_len):
    # decode_batch's decoder (crf/model.py:196-199). The kernel's alpha/beta are fp32 table-LSE scans, the oracle
    # is fp64: near-ties of the edge posteriors may resolve differently, so require >= 99.5 % identical path entries
    # on random scores and identical decoded sequences on confident (model-like) scores.
    rng = np.random.default_rng(60 + state_len)
    N, T = (3, 90) if state_len == 5 else (8, 250)
    sc = _scores(rng, N, T, 4 ** (state_len + 1), "normal")
    moves, path = decode.posterior_viterbi(torch.from_numpy(sc).cuda())
    om, op = crf_ref.posterior_viterbi(sc, state_len)
    agree = (path.numpy() == op).mean()
    assert agree >= 0.995, agree
    assert np.array_equal(moves.numpy(), (path.numpy() != 0).astype(np.int8))


def test_posterior_viterbi_confident_scores_exact_and_model_api():
    from bonito_amd.synthetic import make_model
    rng = np.random.default_rng(9)
    model = make_model("fast")
    state_len, N, T = model.seqdist.state_len, 6, 400
    S = 4 ** state_len
    sc = (rng.standard_normal((N, T, 4 * S)) * 0.5 - 3.0)
    for n in range(N):
        st = int(rng.integers(S))
        for t in range(T):
            if rng.random() < 0.45:
                new = (st * 4 + int(rng.integers(4))) % S
                sc[n, t, new * 4 + st // (S // 4)] = 4.5
                st = new
    sc = np.clip(sc, -5, 5).astype(np.float16)
    moves, path = decode.posterior_viterbi(torch.from_numpy(sc).cuda())
    om, op = crf_ref.posterior_viterbi(sc, state_len)
    assert np.array_equal(path.numpy(), op)
    assert np.array_equal(moves.numpy(), om)
    seqs = model.decode_batch(torch.from_numpy(sc).cuda())
    assert seqs == [model.seqdist.path_to_str(p) for p in op]


def test_beam_fork_option_does_not_change_results():
    # the posterior scan may run next to the beam kernel on a helper stream (bh_set_option "beam_fork")
    rng = np.random.default_rng(77)
    sc = torch.from_numpy(_scores(rng, 9, 300, 256, "normal")).cuda()
    outs = []
    try:
        for v in (0, 1, -1):
            decode.set_option("beam_fork", v)
            outs.append([x.clone() for x in decode.beam_search(sc)])
    finally:
        decode.set_option("beam_fork", -1)
    for o in outs[1:]:
        for a, b in zip(outs[0], o):
            assert torch.equal(a, b)
    with pytest.raises(Exception):
        decode.set_option("no_such_option", 1)


@pytest.mark.parametrize("state_len", [3, 4, 5])
def test_non_temporal_staging_option_does_not_change_results(state_len):
    """bh_set_option("decode_nt", 1): scores and guide rows staged with the non-temporal cache policy (LDS-DMA `nt`, non-temporal guide
    stores) in the backward scan, the beam kernel and the stand-alone posterior scan - the same bytes."""
    rng = np.random.default_rng(500 + state_len)
    N, T = (3, 70) if state_len == 5 else (9, 160)
    sc = torch.from_numpy(_peaky_scores(rng, N, T, state_len)).cuda()
    try:
        base = [x.clone() for x in decode.beam_search(sc, return_qfloat=True)]
        decode.set_option("decode_nt", 1)
        nt = [x.clone() for x in decode.beam_search(sc, return_qfloat=True)]
    finally:
        decode.set_option("decode_nt", 0)
    for a, b in zip(base, nt):
        assert torch.equal(a, b)


@pytest.mark.parametrize("kind", ["normal", "ties", "zeros"])
def test_beam_selection_variants_agree_with_oracle(kind):
    # top-W selection by histogram + exact boundary ranking (default) or by radix search (bh_set_option "beam_select"):
    # same beams, including when many keys tie (crowded boundary bin -> radix fallback inside the histogram path)
    rng = np.random.default_rng(91)
    sc = np.zeros((6, 160, 256), np.float16) if kind == "zeros" else _scores(rng, 6, 160, 256, kind)
    oseq, oqs, omv, oqf = crf_ref.beam_search(sc, 3)
    try:
        for v in (0, 1):
            decode.set_option("beam_select", v)
            seq, qs, mv = decode.beam_search(torch.from_numpy(sc).cuda())
            assert np.array_equal(mv.numpy(), omv), (kind, v)
            assert np.array_equal(seq.numpy(), oseq), (kind, v)
    finally:
        decode.set_option("beam_select", 0)


def test_hip_decoders_match_reference_ctc_crf_fixture():
    """HIP Viterbi / logZ / posterior decoding vs tests/golden/crf_decode.npz = the reference's own CTC_CRF.viterbi / logZ /
    decode_batch lines executed with a torch scan in place of koi's kernels (tests/golden/make_golden.py)."""
    import json
    import os
    from conftest import GOLDEN
    from bonito_amd.crf.model import CTC_CRF
    z = np.load(os.path.join(GOLDEN, "crf_decode.npz"))
    for sl in (1, 2, 3, 4):
        x = torch.from_numpy(z["x%d" % sl]).cuda()
        moves, path = decode.viterbi(x)
        assert np.array_equal(path.numpy(), z["viterbi%d" % sl]), sl
        sd = CTC_CRF(sl, ["N", "A", "C", "G", "T"])
        assert [sd.path_to_str(p) for p in path.numpy()] == json.loads(str(z["str%d" % sl]))
        assert np.abs(decode.logz(x).numpy() - z["logz%d" % sl]).max() < 2e-3
        pm, pp = decode.posterior_viterbi(x)
        assert [sd.path_to_str(p) for p in pp.numpy()] == json.loads(str(z["post_str%d" % sl]))


@pytest.mark.parametrize("state_len", [1, 2, 3, 4, 5])
def test_fused_and_separate_posterior_scan_agree(state_len):
    """bh_set_option("beam_fuse" / "beam_cpw"): the forward / posterior scan as a second wave of the beam kernel's workgroups (default for
    <= 256 states) with one, two or four chunks per workgroup (256 states: the automatic choice depends on the call's size), or the
    round-1 arrangement (own kernel) -- same sequences and moves bit for bit, q-scores to summation order. T = 203 / 61 / 5 / 1 exercise the
    prologue and the last block."""
    rng = np.random.default_rng(300 + state_len)
    for N, T in ([(5, 61), (2, 5)] if state_len == 5 else [(11, 203), (3, 4), (3, 5), (2, 1), (4, 8)]):
        sc = torch.from_numpy(_peaky_scores(rng, N, T, state_len)).cuda()
        outs = []
        try:
            for fuse, cpw in ((1, 4), (1, 2), (1, 1), (0, 0)):
                decode.set_option("beam_fuse", fuse)
                decode.set_option("beam_cpw", cpw)
                outs.append(decode.beam_search(sc, return_qfloat=True))
        finally:
            decode.set_option("beam_fuse", -1)
            decode.set_option("beam_cpw", 0)
        for o in outs[:3]:
            assert torch.equal(o[0], outs[3][0]) and torch.equal(o[2], outs[3][2]), (N, T)
            assert (o[3] - outs[3][3]).abs().max().item() < 1e-4
            assert (o[1] != outs[3][1]).float().mean().item() < 1e-3
