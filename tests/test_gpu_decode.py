"""GPU CRF Viterbi vs the C oracle: bit-exact paths, move tables and path scores (-m gpu)."""
import numpy as np
import pytest
import torch

from bonito_amd import decode
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
