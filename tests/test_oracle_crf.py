"""C oracle (oracle/crf_oracle.c) pinned against exhaustive enumeration and the autograd restatement of
the reference formulation (bonito/crf/model.py:98-103). CPU only."""
import numpy as np
import pytest

from oracle import crf_ref


def _scores(rng, T, N, C, kind="normal"):
    x = rng.standard_normal((T, N, C)) * 2.0
    if kind == "tanh":
        x = np.tanh(x) * 5.0
    return np.clip(x, -5, 5).astype(np.float16)


@pytest.mark.parametrize("state_len,T", [(1, 1), (1, 5), (2, 3)])
def test_viterbi_vs_bruteforce(state_len, T):
    rng = np.random.default_rng(state_len * 10 + T)
    S = 4 ** state_len
    sc = _scores(rng, T, 2, 5 * S)
    _, path, best = crf_ref.viterbi(sc, state_len, layout_5s=True, time_major=True)
    bb, pb = crf_ref.viterbi_bruteforce(sc.astype(np.float64), state_len)
    assert np.allclose(bb, best, atol=1e-5)
    assert (path.T == pb).all()


@pytest.mark.parametrize("state_len", [1, 2, 3, 4])
@pytest.mark.parametrize("kind", ["normal", "tanh"])
def test_viterbi_vs_autograd_reference_formulation(state_len, kind):
    rng = np.random.default_rng(100 + state_len)
    S = 4 ** state_len
    sc = _scores(rng, 40, 3, 5 * S, kind)
    moves, path, _ = crf_ref.viterbi(sc, state_len, layout_5s=True, time_major=True)
    pa = crf_ref.viterbi_autograd(sc.astype(np.float64), state_len)
    assert (path.T == pa).all()
    assert ((path != 0) == (moves != 0)).all()


@pytest.mark.parametrize("state_len", [2, 3])
def test_koi_layout_equals_expanded_layout(state_len):
    """4S scores + scalar blank == the same scores with the blank column materialised (nn.py:291-297)."""
    rng = np.random.default_rng(5)
    S = 4 ** state_len
    sc4 = _scores(rng, 3, 50, 4 * S).transpose(0, 1, 2)  # [N,T,4S]
    m4, p4, b4 = crf_ref.viterbi(sc4, state_len, layout_5s=False, blank=2.0)
    sc5 = crf_ref.expand_blanks(sc4, np.float16(2.0))
    m5, p5, b5 = crf_ref.viterbi(sc5, state_len, layout_5s=True)
    assert (p4 == p5).all() and (m4 == m5).all() and np.array_equal(b4, b5)


def test_path_is_state_consistent():
    """The emitted bases are the low digits of a consistent k-mer walk."""
    rng = np.random.default_rng(9)
    sl, S = 3, 64
    sc = _scores(rng, 2, 120, 4 * S)
    moves, path, _ = crf_ref.viterbi(sc, sl, blank=2.0)
    assert set(np.unique(path)) <= {0, 1, 2, 3, 4}
    assert ((path > 0) == (moves == 1)).all()


def test_logz_upper_bounds_viterbi():
    rng = np.random.default_rng(11)
    sc = _scores(rng, 3, 30, 4 * 16)
    _, _, best = crf_ref.viterbi(sc, 2, blank=2.0)
    lz = crf_ref.logz(sc, 2, blank=2.0)
    assert (lz >= best - 1e-4).all()
    assert (lz <= best + 30 * np.log(5) + np.log(16) + 1e-3).all()


# ---- beam search oracle self-consistency (CPU) ----------------------------------------------------
def test_table_lse2_accuracy_and_symmetry():
    rng = np.random.default_rng(0)
    for a, b in rng.uniform(-20, 20, (200, 2)):
        v = crf_ref.lse2(float(a), float(b))
        assert abs(v - np.logaddexp(np.float32(a), np.float32(b))) < 2e-6 * max(1.0, abs(v)) + 1e-6
        assert v == crf_ref.lse2(float(b), float(a))
    assert crf_ref.lse2(1.0, -np.inf) == 1.0 and crf_ref.lse2(-np.inf, -np.inf) == -np.inf


@pytest.mark.parametrize("state_len", [1, 2, 3])
def test_backward_forward_normalisation(state_len):
    """logZ from the normalised backward scan equals the plain fp64 scan; class posteriors sum to 1."""
    rng = np.random.default_rng(state_len)
    sc = _scores(rng, 90, 3, 4 ** (state_len + 1)).transpose(1, 0, 2).copy()      # [N,T,4S]
    beta, B, lz = crf_ref.backward(sc, state_len)
    assert np.allclose(lz, crf_ref.logz(sc, state_len), rtol=0, atol=2e-3)
    P = crf_ref.forward_post(sc, state_len, beta, B, lz)
    assert np.abs(P.sum(-1) - 1.0).max() < 1e-4 and P.min() >= 0
    assert np.abs(beta[:, :, 0]).max() == 0.0                                        # normalised by state 0


def test_beam_width_one_is_greedy_guided_and_outputs_are_consistent():
    rng = np.random.default_rng(3)
    sc = _scores(rng, 100, 2, 64).transpose(1, 0, 2).copy()
    for bw in (1, 4, 32):
        seq, qs, mv, qf = crf_ref.beam_search(sc, 2, beam_width=bw)
        assert ((seq != 0) == (mv == 1)).all() and ((qs != 0) == (mv == 1)).all()
        assert set(np.unique(seq)) <= {0, 65, 67, 71, 84}
        assert (qf[mv == 1] >= 1).all() and (qf[mv == 1] <= 50).all()


def test_wide_beam_recovers_planted_sequence():
    """With sharply peaked scores every decoder must read the planted path back."""
    rng = np.random.default_rng(8)
    sl, S, T = 2, 16, 80
    x = np.full((1, T, 4 * S), -4.0, np.float32)
    st, want = 5, []
    for t in range(T):
        if t % 3 == 0:
            b = int(rng.integers(4))
            new = ((st << 2) | b) & (S - 1)
            x[0, t, new * 4 + (st >> 2)] = 5.0
            st = new
            want.append("ACGT"[b])
        else:
            x[0, t] = -5.0          # every move is very unlikely -> stay (blank 2.0)
    seq, qs, mv, qf = crf_ref.beam_search(x.astype(np.float16), sl)
    got = "".join(chr(c) for c in seq[0] if c)
    assert got == "".join(want)
    _, path, _ = crf_ref.viterbi(x.astype(np.float16), sl)
    assert "".join("NACGT"[p] for p in path[0] if p) == got
    assert np.median(qf[mv == 1]) > 20


def _beam_vs_exact_map(cases, make):
    misses, worst = 0, 0.0
    for sl, T, seed in cases:
        x = make(sl, T, seed)
        total = crf_ref.map_sequence_bruteforce(x[0], sl)
        best = max(total, key=total.get)
        seq, _, mv, _ = crf_ref.beam_search(x, sl)
        got = "".join(chr(c) for c in seq[0] if c)
        assert got in total and len(got) == int(mv.sum())
        if got != best:
            misses += 1
            worst = max(worst, total[best] - total[got])
    return misses, worst


def test_beam_search_finds_the_exact_map_sequence_on_peaked_scores_and_stays_close_on_flat_ones():
    """The beam search is the builder's own definition (koi is a closed wheel), so it is pinned to what ANY beam search approximates: the
    sequence with the largest total path probability, computed exactly by an unpruned prefix search in fp64 (tiny T only). Scores with
    a planted path (what a trained head emits): the beam's sequence IS the exact MAP sequence in every case. Flat random scores (the
    hardest input: thousands of sequences within a nat of each other, 5461-87381 distinct sequences against a beam of 32): the MAP in
    29 of 36 cases (asserted: at least 75 %) and never more than one nat below it (measured worst: 0.78)."""
    def peaked(sl, T, seed):
        rng = np.random.default_rng(5000 * sl + 10 * T + seed)
        S = 4 ** sl
        x = np.clip(rng.standard_normal((1, T, 4 * S)) * 1.5, -5, 5)
        st = int(rng.integers(S))
        for t in range(T):
            if rng.random() < 0.5:
                ns = ((st << 2) | int(rng.integers(4))) & (S - 1)
                x[0, t, ns * 4 + st // (S // 4)] += 6.0
                st = ns
            else:
                x[0, t] -= 3.0
        return np.clip(x, -5, 5).astype(np.float16)

    def flat(sl, T, seed):
        rng = np.random.default_rng(1000 * sl + 10 * T + seed)
        return np.clip(rng.standard_normal((1, T, 4 * 4 ** sl)) * 2.5, -5, 5).astype(np.float16)

    misses, _ = _beam_vs_exact_map([(sl, T, seed) for sl, T in ((1, 8), (2, 7)) for seed in range(12)], peaked)
    assert misses == 0
    cases = [(sl, T, seed) for sl, T in ((1, 6), (1, 8), (2, 6)) for seed in range(12)]
    misses, worst = _beam_vs_exact_map(cases, flat)
    assert misses <= 0.25 * len(cases) and worst < 1.0, (misses, worst)


def test_reverse_complement_matches_reference_fixture():
    import os
    from conftest import GOLDEN
    z = np.load(os.path.join(GOLDEN, "crf_rc.npz"))
    for sl in (1, 2, 3):
        got = crf_ref.reverse_complement(z["x%d" % sl], sl, layout_5s=True)
        assert np.array_equal(got, z["y%d" % sl]), sl
        # involution
        assert np.array_equal(crf_ref.reverse_complement(got, sl), z["x%d" % sl])


def test_reverse_complement_decodes_to_revcomp_sequence():
    """Paths map one-to-one with equal total score, so the best score is preserved and the Viterbi basecall
    of the permuted scores is the reverse complement of the original (up to the k-mer context at the ends)."""
    rng = np.random.default_rng(21)
    sl = 3
    sc = _scores(rng, 80, 2, 4 * 64).transpose(1, 0, 2).copy()          # koi layout [N,T,4S]
    _, p_fwd, b_fwd = crf_ref.viterbi(sc, sl)
    rc = crf_ref.reverse_complement(sc, sl, layout_5s=False)
    _, p_rev, b_rev = crf_ref.viterbi(rc, sl)
    assert np.allclose(b_fwd, b_rev, rtol=0, atol=1e-3)
    comp = {1: 4, 2: 3, 3: 2, 4: 1}
    for n in range(2):
        fwd = "".join("NACGT"[comp[b]] for b in reversed([b for b in p_fwd[n] if b]))
        rev = "".join("NACGT"[b] for b in p_rev[n] if b)
        core = fwd[sl:-sl]
        assert len(core) > 10 and core in rev, (fwd, rev)
    assert np.array_equal(crf_ref.reverse_complement(rc, sl, layout_5s=False), sc)      # involution


@pytest.mark.parametrize("state_len", [1, 2, 3])
def test_posterior_viterbi_matches_autograd_reference_formulation(state_len):
    # decode_batch (crf/model.py:196-199): viterbi(log(posteriors + 1e-8)); the C oracle uses explicit fp64
    # forward/backward, the restatement uses autograd posteriors as the reference does.
    rng = np.random.default_rng(40 + state_len)
    sc = np.clip(rng.standard_normal((3, 60, 4 ** (state_len + 1))) * 2.5, -5, 5).astype(np.float16)
    moves, path = crf_ref.posterior_viterbi(sc, state_len)
    ref = crf_ref.posterior_viterbi_autograd(sc, state_len)
    assert np.array_equal(path, ref)
    assert np.array_equal(moves, (path != 0).astype(np.int8))


def test_posterior_viterbi_equals_viterbi_on_peaked_scores():
    # one dominant path -> posterior decoding and MAP decoding agree
    rng = np.random.default_rng(5)
    state_len, T = 3, 120
    S = 4 ** state_len
    sc = np.full((1, T, 4 * S), -5.0, np.float16)
    st = 0
    for t in range(T):
        if rng.random() < 0.5:
            b = int(rng.integers(4)); new = (st * 4 + b) % S
            sc[0, t, new * 4 + (st // (S // 4))] = 5.0
            st = new
    _, p1 = crf_ref.posterior_viterbi(sc, state_len)
    _, p2, _ = crf_ref.viterbi(sc, state_len, layout_5s=False, blank=2.0)
    assert np.array_equal(p1, p2)


def test_oracle_matches_reference_ctc_crf_executed_with_stubbed_koi():
    """tests/golden/crf_decode.npz holds the outputs of the REFERENCE's own CTC_CRF.viterbi / logZ / path_to_str and
    SeqdistModel.decode_batch lines (bonito/crf/model.py:47-52,98-108,196-199), run by tests/golden/make_golden.py with a torch
    scan standing in for the four koi.ctc names. The C oracle must reproduce them: paths bit for bit, logZ to fp32 accuracy,
    decoded strings exactly."""
    import json
    import os
    from conftest import GOLDEN
    z = np.load(os.path.join(GOLDEN, "crf_decode.npz"))
    alphabet = np.frombuffer(b"NACGT", dtype="u1")
    for sl in (1, 2, 3, 4):
        x = z["x%d" % sl]
        mv, path, best = crf_ref.viterbi(x, sl, blank=2.0)
        assert np.array_equal(path, z["viterbi%d" % sl]), sl
        assert np.array_equal(mv, (z["viterbi%d" % sl] != 0).astype(np.int8))
        assert [alphabet[p[p != 0]].tobytes().decode() for p in path] == json.loads(str(z["str%d" % sl]))
        assert np.abs(crf_ref.logz(x, sl) - z["logz%d" % sl]).max() < 1e-4
        pm, pp = crf_ref.posterior_viterbi(x, sl)
        assert [alphabet[p[p != 0]].tobytes().decode() for p in pp] == json.loads(str(z["post_str%d" % sl]))


def test_bs2_deterministic_exp_and_log_accuracy_and_range():
    """The two elementary functions of the linear-domain guide (include/bh_bs2.h): accuracy against libm, the clamp, and that the
    exponential never leaves the normal fp32 range (the bit-exact contract with the GPU rests on normal operands only)."""
    xs = np.concatenate([np.linspace(-45, 45, 4001), np.float16(np.linspace(-5, 5, 2001)).astype(np.float64), [65504.0, -65504.0]])
    e = np.array([crf_ref.bs2_exp(x) for x in xs], np.float64)
    ref = np.exp(np.clip(xs, -40.0, 40.0))
    assert np.abs(e / ref - 1.0).max() < 3e-6 and np.abs(e / ref - 1.0)[np.abs(xs) <= 5].max() < 1e-6     # (log2e in fp32 times 58 at the clamp)
    assert e.min() > 1e-18 and e.max() < 3e17 and (e >= 2.0 ** -126).all()
    assert crf_ref.bs2_exp(0.0) == np.float32(1.0000001192092896)               # the polynomial's constant term, not exactly 1
    vs = np.exp(np.random.default_rng(0).uniform(-85, 2, 4000)).astype(np.float32)
    lg = np.array([crf_ref.bs2_log(v) for v in vs], np.float64)
    assert np.abs(lg - np.log(vs.astype(np.float64))).max() < 2e-5
    assert all(crf_ref.bs2_log(a) <= crf_ref.bs2_log(b) for a, b in zip(np.sort(vs)[:-1:50], np.sort(vs)[1::50]))   # monotone where it ranks


@pytest.mark.parametrize("state_len", [1, 2, 3, 4])
def test_bs2_guide_equals_the_log_semiring_guide_and_rows_are_normalised(state_len):
    """oracle_bs2_backward (BS-2, linear domain) against the table-lse2 scan of rounds 1-4: ln b_t[s] - ln b_t[0] == beta~_t[s] to 1e-5;
    every row's maximum lies in [1, 2) (an exact power-of-two scaling), nothing underflows; and the fp64 posteriors agree with the
    fp32 table scan's to 1e-5 and sum to one."""
    rng = np.random.default_rng(10 + state_len)
    S = 4 ** state_len
    sc = rng.normal(0, 2.5, (3, 120, 4 * S)).clip(-5, 5).astype(np.float16)
    b = crf_ref.bs2_backward(sc, state_len)
    beta, B, lz = crf_ref.backward(sc, state_len)
    g = np.log(b.astype(np.float64))
    assert np.abs((g - g[:, :, :1]) - beta).max() < 2e-5
    mx = b.max(axis=2)
    assert (mx >= 1.0).all() and (mx < 2.0).all() and b.min() > 1e-30
    assert (b[:, -1, :] == 1.0).all()
    P = crf_ref.posteriors_f64(sc, state_len)
    assert np.abs(P.sum(-1) - 1.0).max() < 1e-6
    assert np.abs(P - crf_ref.forward_post(sc, state_len, beta, B, lz)).max() < 2e-5


def test_bs2_out_of_range_scores_are_clamped_not_overflowed():
    rng = np.random.default_rng(4)
    sc = rng.normal(0, 40, (2, 50, 64)).astype(np.float16)
    sc[0, 10:20] = np.float16(65504)
    sc[1, :25] = np.float16(-65504)
    for blank in (2.0, 60.0, -60.0):
        b = crf_ref.bs2_backward(sc, 2, blank=blank)
        assert np.isfinite(b).all() and (b > 0).all() and b.min() >= 2.0 ** -126
        seq, qs, mv, qf = crf_ref.beam_search(sc, 2, blank=blank)
        assert np.isfinite(qf).all()
    # the all-saturated block is symmetric: uniform class posteriors
    P = crf_ref.posteriors_f64(sc, 2)
    assert np.abs(P[0, 11:18] - 0.25).max() < 1e-6
