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
