"""The 8-bit recurrent path Q8-1 on the GPU (-m gpu): `lstm_layer_q8_kernel` against its definition, oracle/lstm_q8_ref.py.

* the integer part is checked EXACTLY: the int32 sums the kernel's gate arithmetic starts from (test hook of
  bh_lstm_q8_layer) equal numpy's integer matmuls of the oracle's quantised weights with the quantised input / with the int8
  h the kernel itself published one step earlier; the published int8 h is rint(127 * fp16 h);
* the cell (fast exp / rcp on the GPU, libm on the CPU) is checked on those exact pre-activations within 2e-3;
* end to end, the engine with quantize=True stays within the oracle's stated bound of the fp32 oracle and decodes the same
  Viterbi paths on hac- and fast-shaped synthetic models.
koi's own int8 kernels are closed source: parity with koi is unpinned; Q8-1 is this repository's definition.
"""
import ctypes as C

import numpy as np
import pytest
import torch

from bonito_amd import _lib, decode, synthetic
from bonito_amd.engine import HipEncoder
from oracle import crf_ref, lstm_q8_ref, nn_ref

pytestmark = pytest.mark.gpu


def unscramble(frag, T, N, H):
    """int8 fragment order [T][N/16][ceil(H/64)][64 lanes][16] -> [T][N][H]."""
    nk8 = (H + 63) // 64
    a = frag.reshape(T, N // 16, nk8, 4, 16, 16)              # t, ring, ks, q, c, j
    return a.transpose(0, 1, 4, 2, 3, 5).reshape(T, N, nk8 * 64)[:, :, :H]


def run_layer(x, w_ih, w_hh, bias, bound, reverse, variant=0):
    T, N, H = x.shape
    lib = _lib.lib()
    nk8 = (H + 63) // 64
    xd = x.cuda()
    h16 = torch.zeros(T, N, H, dtype=torch.float16, device="cuda")
    hq = torch.zeros(T * (N // 16) * nk8 * 1024, dtype=torch.int8, device="cuda")
    sums = torch.zeros(T, N, 4 * H, 2, dtype=torch.int32, device="cuda")
    w1, w2, b = (np.ascontiguousarray(a, np.float32) for a in (w_ih, w_hh, bias))
    _lib.check(lib.bh_lstm_q8_layer(_lib.ptr(xd), float(bound), w1.ctypes.data_as(C.c_void_p), w2.ctypes.data_as(C.c_void_p),
                                    b.ctypes.data_as(C.c_void_p), T, N, H, int(reverse), variant, _lib.ptr(h16), _lib.ptr(hq),
                                    _lib.ptr(sums), _lib.stream_ptr()), "bh_lstm_q8_layer")
    return h16.cpu().numpy(), unscramble(hq.cpu().numpy(), T, N, H), sums.cpu().numpy()


def _sigmoid(v):
    return 1.0 / (1.0 + np.exp(-v))


@pytest.mark.parametrize("H,variant,N,reverse,bound", [(384, 0, 32, 0, 1.0), (384, 0, 48, 1, 4.0), (384, 1, 32, 1, 1.0), (384, 2, 32, 0, 1.0), (96, 0, 32, 0, 1.0),
                                                       (128, 0, 16, 1, 3.5), (64, 0, 48, 0, 1.0), (192, 0, 32, 1, 1.0), (256, 0, 16, 0, 1.0),
                                                       (288, 0, 16, 1, 1.0), (512, 0, 16, 0, 1.0)])
def test_q8_layer_integer_sums_exact_and_cell_close(H, variant, N, reverse, bound):
    T = 37
    rng = np.random.default_rng(H + variant)
    w_ih = (rng.standard_normal((4 * H, H)) * rng.uniform(0.02, 0.12, (4 * H, 1))).astype(np.float32)
    w_hh = (rng.standard_normal((4 * H, H)) * rng.uniform(0.02, 0.12, (4 * H, 1))).astype(np.float32)
    bias = (rng.standard_normal(4 * H) * 0.3).astype(np.float32)
    x = torch.from_numpy(np.clip(rng.standard_normal((T, N, H)) * 0.6 * bound, -1.2 * bound, 1.2 * bound).astype(np.float16))
    h16, hq, sums = run_layer(x, w_ih, w_hh, bias, bound, reverse, variant)
    q_ih, s_ih = lstm_q8_ref.quantise_rows(w_ih)
    q_hh, s_hh = lstm_q8_ref.quantise_rows(w_hh)
    xq = lstm_q8_ref.quantise_act(x.float().numpy(), bound).astype(np.int64)
    want_x = xq.reshape(T * N, H) @ q_ih.astype(np.int64).T
    assert np.array_equal(sums[..., 0].reshape(T * N, 4 * H), want_x)                      # exact
    order = list(range(T - 1, -1, -1)) if reverse else list(range(T))
    prev = np.zeros((N, H), np.int64)
    c = np.zeros((N, H), np.float32)
    sxs = s_ih * np.float32(bound / 127.0)
    shs = s_hh / np.float32(127.0)
    worst = 0.0
    for t in order:
        want_h = prev @ q_hh.astype(np.int64).T
        assert np.array_equal(sums[t, :, :, 1], want_h), t                                 # exact, from the kernel's own h_{t-1}
        # the oracle's fp32 pre-activation from the exact sums, then its cell; the kernel's cell state is tracked through ITS h
        g = (sums[t, :, :, 0].astype(np.float32) * sxs + bias) + sums[t, :, :, 1].astype(np.float32) * shs
        i, f, gg, o = np.split(g, 4, axis=-1)
        c = (_sigmoid(f) * c + _sigmoid(i) * np.tanh(gg)).astype(np.float32)
        h = (_sigmoid(o) * np.tanh(c)).astype(np.float32)
        worst = max(worst, float(np.abs(h - h16[t].astype(np.float32)).max()))
        assert np.array_equal(hq[t], np.clip(np.rint(h16[t].astype(np.float32) * np.float32(127.0)), -127, 127).astype(np.int8))
        prev = hq[t].astype(np.int64)
    assert worst < 2e-3, worst        # fp16 rounding of h (4.9e-4) + fast exp / rcp, accumulated through c over 37 steps


def _scores(model, x, quantize):
    enc = HipEncoder(model.encoder, batchsize=x.shape[0], chunksize=x.shape[-1], quantize=quantize)
    out = enc(x.half().cuda())
    enc.check()
    return out, enc


@pytest.mark.parametrize("name,N,L", [("hac", 21, 1800), ("fast", 19, 1800)])
def test_q8_engine_matches_oracle_and_decodes_like_fp32(name, N, L):
    model = synthetic.make_model(name, batchsize=N, chunksize=L)
    nn_ref.round_params_to_half_(model)
    x = torch.randn(N, 1, L, generator=torch.Generator().manual_seed(3)).half()
    got, enc = _scores(model, x, True)
    assert "lstm_layer_q8_kernel" in enc.describe()
    with torch.no_grad():
        ref = nn_ref.forward(model.encoder, x.float(), expand_blanks=False).permute(1, 0, 2)
        q8 = lstm_q8_ref.forward_q8(model.encoder, x.float(), expand_blanks=False).permute(1, 0, 2)
    g = got.cpu().float()
    d_def = (g - q8).abs()              # kernel vs its definition: the integer part is exact, the cell differs by ~1e-3, and a
    d_ref = (g - ref).abs()             # rare flip of a quantisation bucket (1/127) propagates -> a loose bound on max, tight mean
    # measured on MI355X: kernel vs definition max 0.086 / mean 0.010-0.011 (the convolutions ahead of the first recurrent layer
    # run in fp16 on the GPU and in fp32 in the oracle, so ~10 % of its int8 inputs land in the neighbouring bucket)
    assert d_def.mean().item() < 0.03 and d_def.max().item() < 0.3, (d_def.max().item(), d_def.mean().item())      # max: 3.5 x measured
    assert d_ref.max().item() < 0.6 and d_ref.mean().item() < 0.06, (d_ref.max().item(), d_ref.mean().item())
    sl = model.seqdist.state_len
    paths = [crf_ref.viterbi(s.contiguous().numpy().astype(np.float16), sl, blank=2.0)[1] for s in (ref, g)]
    assert (paths[0] == paths[1]).mean() > 0.99
    # the fp16 engine on the same model, for scale
    f16, _ = _scores(model, x, False)
    assert (f16.cpu().float() - ref).abs().max().item() < 3e-2


@pytest.mark.parametrize("H", [64, 96, 128, 192, 256, 288, 320, 384, 448, 480, 512])
def test_quantize_covers_or_falls_back_for_every_width(H):
    """`quantize=True` on any hidden size the ring kernels serve: widths with an 8-bit kernel instance run it, the others keep the
    fp16 kernels (use_hip's contract) - never a failing forward (advisor finding, round 2: the shape predicate accepted widths
    without an instance). Either way the scores stay close to the fp32 oracle."""
    from bonito_amd import nn as bnn
    cfg = synthetic.lstm_crf_encoder_config(H, 3)
    torch.manual_seed(H)
    model = bnn.from_dict(cfg).eval()
    nn_ref.round_params_to_half_(model)
    x = torch.randn(32, 1, 900, generator=torch.Generator().manual_seed(H)).half()
    enc = HipEncoder(model, batchsize=32, chunksize=900, quantize=True)
    got = enc(x.cuda())
    enc.check()
    layout = enc.describe()
    has_q8 = H in (64, 96, 128, 192, 256, 288, 384, 512)          # 320, 448, 480: accepted by the round-2 predicate, no kernel instance
    assert ("lstm_layer_q8_kernel" in layout) == has_q8, layout
    from bonito_amd.crf.basecall import q8_covers
    assert q8_covers(H) == has_q8          # the host-side lane policy (max_lanes) uses the same predicate
    with torch.no_grad():
        ref = nn_ref.forward(model, x.float(), expand_blanks=False).permute(1, 0, 2)
    d = (got.cpu().float() - ref).abs()
    assert d.mean().item() < (0.06 if has_q8 else 8e-3), (H, d.max().item(), d.mean().item())      # fp16 fallback measured: 3.8e-3 at 480


def test_q8_option_off_runs_the_fp16_kernels():
    model = synthetic.make_model("hac", batchsize=16, chunksize=1200)
    x = torch.randn(16, 1, 1200, generator=torch.Generator().manual_seed(1)).half().cuda()
    enc = HipEncoder(model.encoder, batchsize=16, chunksize=1200, quantize=True)
    a = enc(x)
    enc.set_option("lstm_q8", 0)
    b = enc(x)
    enc.check()
    plain = HipEncoder(model.encoder, batchsize=16, chunksize=1200)
    assert torch.equal(b, plain(x)) and not torch.equal(a, b)
    assert "q8" not in plain.describe()


def test_q8_full_size_hac_512x10000():
    """BASELINE size with quantize=True: no exchange timeout, finite scores, four chunks of the batch against the Q8-1 oracle
    and the fp32 oracle, decode of the engine's scores bit-exact vs the C oracle; both kernel geometries agree bit for bit."""
    model = synthetic.make_model("hac", batchsize=512, chunksize=10000)
    nn_ref.round_params_to_half_(model)
    x = torch.randn(512, 1, 10000, generator=torch.Generator().manual_seed(25)).half()
    got, enc = _scores(model, x, True)
    assert got.shape == (512, 1667, 1024) and torch.isfinite(got.float()).all()
    rows = [0, 17, 255, 511]
    with torch.no_grad():
        ref = nn_ref.forward(model.encoder, x[rows].float(), expand_blanks=False).permute(1, 0, 2)
        q8 = lstm_q8_ref.forward_q8(model.encoder, x[rows].float(), expand_blanks=False).permute(1, 0, 2)
    g = got[rows].cpu().float()
    d_def, d_ref = (g - q8).abs(), (g - ref).abs()
    assert d_def.mean().item() < 0.02, (d_def.max().item(), d_def.mean().item())
    assert d_ref.mean().item() < 0.08, (d_ref.max().item(), d_ref.mean().item())
    sub = got[rows].cpu().numpy()
    seq, qs, mv = decode.beam_search(got)
    oseq, oqs, omv, _ = crf_ref.beam_search(sub, 4)
    assert np.array_equal(mv.numpy()[rows], omv) and np.array_equal(seq.numpy()[rows], oseq)
    p_ref = crf_ref.viterbi(ref.contiguous().numpy().astype(np.float16), 4, blank=2.0)[1]
    p_q8 = crf_ref.viterbi(sub, 4, blank=2.0)[1]
    assert (p_ref == p_q8).mean() > 0.98
    enc.close()
    try:
        for variant, tag in ((1, "lstm_layer_q8_kernel<6,1>"), (2, "lstm_layer_q8_kernel<6,3>")):
            decode.set_option("lstm_q8_variant", variant)       # 1: 4 units per wave, 3 workgroups per CU; 2: occupancy 2
            alt, enc2 = _scores(model, x, True)
            assert tag in enc2.describe()
            assert torch.equal(alt, got), variant
            enc2.close()
    finally:
        decode.set_option("lstm_q8_variant", 0)
