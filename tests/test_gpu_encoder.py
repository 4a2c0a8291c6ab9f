"""End-to-end encoder engine (C ABI bh_encoder_*) vs the committed reference outputs (-m gpu)."""
import numpy as np
import pytest
import torch

from conftest import NN_FIXTURES, build_model, load_nn_fixture, ref_scores_to_koi
from bonito_amd.engine import HipEncoder
from oracle import crf_ref, nn_ref

pytestmark = pytest.mark.gpu

# fp16 activations / fp32 accumulation vs the fp32 CPU reference; scores live in [-5, 5] (fp16 ulp 4e-3).
# <= 5x the largest error measured on MI355X over the fixtures and the full-size BASELINE runs (max 4.9e-3, mean 6.0e-4:
# tests/test_gpu_configs.py, profiles/r02_parity.json)
TOL_MAX, TOL_MEAN = 2.4e-2, 3e-3


def _run(name, batch_pad=None):
    cfg, sd, x, y = load_nn_fixture(name)
    model = build_model(cfg, sd)
    enc = HipEncoder(model, batchsize=batch_pad or x.shape[0], chunksize=x.shape[-1])
    got = enc(x.half().cuda())
    enc.check()
    has_blank = any(getattr(m, "blank_score", None) is not None for m in model.modules())
    want = ref_scores_to_koi(y, has_blank)
    return got.cpu().float(), want


@pytest.mark.parametrize("name", NN_FIXTURES)
def test_encoder_matches_reference_fixture(name):
    got, want = _run(name)
    assert got.shape == want.shape
    d = (got - want).abs()
    assert d.max().item() < TOL_MAX and d.mean().item() < TOL_MEAN, (d.max().item(), d.mean().item())


def test_encoder_batch_padding_and_reuse():
    """N=3 is padded to 16 inside the engine; a larger max_batch and repeated calls change nothing."""
    cfg, sd, x, y = load_nn_fixture("lstm32_sl2")
    model = build_model(cfg, sd)
    enc = HipEncoder(model, batchsize=40, chunksize=x.shape[-1])
    xs = x.half().cuda()
    a = enc(xs)
    b = enc(xs)
    big = enc(torch.cat([xs] * 11)[:33])
    enc.check()
    assert torch.equal(a, b)
    assert torch.equal(big[:3], a) and torch.equal(big[30:33], a)


def test_encoder_shorter_chunk_than_max():
    cfg, sd, x, y = load_nn_fixture("lstm32_sl2")
    model = build_model(cfg, sd)
    enc = HipEncoder(model, batchsize=4, chunksize=1200)
    xs = x[:, :, :300].half()
    got = enc(xs.cuda()).cpu().float()
    enc.check()
    with torch.no_grad():
        want = ref_scores_to_koi(nn_ref.forward(model, xs.float()))
    assert got.shape == want.shape == (3, 50, 64)
    assert (got - want).abs().max().item() < TOL_MAX


def test_model_surface_forward_and_decode():
    """bonito_amd.crf.Model: config -> use_koi -> load_state_dict -> half -> cuda -> forward -> HIP Viterbi,
    and the HIP decode of the HIP scores equals the oracle decode of the same scores bit for bit."""
    from bonito_amd import decode
    from bonito_amd.crf.model import Model
    cfg, sd, x, y = load_nn_fixture("lstm64_sl3")
    config = {"model": {"package": "bonito.crf"}, "labels": {"labels": ["N", "A", "C", "G", "T"]},
              "input": {"features": 1}, "global_norm": {"state_len": 3}, "encoder": cfg}
    model = Model(config)
    model.use_koi(batchsize=4, chunksize=1200, quantize=False)
    model.load_state_dict({"encoder." + k: v for k, v in sd.items()})
    model = model.half().eval().to("cuda")
    assert model.stride == 6
    scores = model(x.half().cuda())
    assert scores.shape == (2, 200, 256) and scores.dtype == torch.float16 and scores.is_contiguous()
    moves, path = decode.viterbi(scores)
    om, op, _ = crf_ref.viterbi(scores.cpu().numpy(), 3, blank=2.0)
    assert np.array_equal(path.numpy(), op) and np.array_equal(moves.numpy(), om)
    seq = decode.to_str(decode.path_to_sequence(path[0]))
    assert set(seq) <= set("ACGT") and len(seq) == int((path[0] != 0).sum())


def test_engine_fails_loudly_on_unsupported_layers():
    from bonito_amd import nn as bnn
    from bonito_amd.engine import LoweringError
    with pytest.raises(LoweringError):
        HipEncoder(bnn.Serial([bnn.Reverse(bnn.Linear(8, 8))]), 1, 100)


# ---- transformer encoder ----------------------------------------------------------------------------
from conftest import TF_FIXTURES, build_tf_model, load_tf_fixture  # noqa: E402


@pytest.mark.parametrize("name", TF_FIXTURES)
def test_transformer_encoder_matches_reference_fixture(name):
    cfg, sd, x, y = load_tf_fixture(name)
    model = build_tf_model(cfg, sd)
    model.use_koi(batchsize=x.shape[0], chunksize=x.shape[-1], quantize=False)
    model = model.half().to("cuda")
    got = model(x.half().cuda()).cpu().float()
    model._hip.check()
    want = ref_scores_to_koi(y)
    assert got.shape == want.shape
    d = (got - want).abs()
    # scores are scaled by 5 with no squashing: tolerance relative to their range
    rng = want.abs().max().item()
    assert d.max().item() < 2e-2 * max(rng, 1.0) and d.mean().item() < 3e-3 * max(rng, 1.0), (d.max().item(), d.mean().item(), rng)
    assert model.stride == 6


@pytest.mark.parametrize("name", TF_FIXTURES)
def test_transformer_attention_paths_agree(name):
    """attn_ring = 1 (default: rotary + softmax scale in the Wqkv epilogue, persistent ring-buffer attention kernel) and
    attn_ring = 0 (rotation applied while staging K/Q, one workgroup per query block) are two implementations of
    MultiHeadAttention: both match the reference fixture, and each other to fp16 rounding (the fused rotation rounds once,
    the staged one rounds q/k before and after rotating, as flash-attn's in-place rotary does)."""
    cfg, sd, x, y = load_tf_fixture(name)
    model = build_tf_model(cfg, sd)
    want = ref_scores_to_koi(y)
    rng = max(want.abs().max().item(), 1.0)
    outs = {}
    for ring in (1, 0):
        enc = HipEncoder(model.encoder, batchsize=x.shape[0], chunksize=x.shape[-1])
        enc.set_option("attn_ring", ring)
        outs[ring] = enc(x.half().cuda()).cpu().float()
        enc.check()
        d = (outs[ring] - want).abs()
        assert d.max().item() < 2e-2 * rng and d.mean().item() < 3e-3 * rng, (ring, d.max().item(), d.mean().item())
    assert (outs[1] - outs[0]).abs().max().item() < 1e-2 * rng


@pytest.mark.parametrize("name", TF_FIXTURES)
def test_transformer_residual_fused_into_the_projections_agrees(name):
    """DeepNorm's alpha * x (bonito/transformer/model.py:125-128: `norm(sublayer(x), alpha * x)`) is added in the epilogue of the
    out_proj / fc2 GEMMs (fp32 accumulator + alpha * x, one rounding) and the norm kernel reads one tensor (`norm_fuse` 1; an option - it measured no faster);
    `norm_fuse` 0 (default) is the separate residual read in the norm kernel. Both match the reference fixture, and each other to fp16 rounding."""
    cfg, sd, x, y = load_tf_fixture(name)
    model = build_tf_model(cfg, sd)
    want = ref_scores_to_koi(y)
    rng = max(want.abs().max().item(), 1.0)
    outs = {}
    for fuse in (1, 0):
        enc = HipEncoder(model.encoder, batchsize=x.shape[0], chunksize=x.shape[-1])
        enc.set_option("norm_fuse", fuse)
        outs[fuse] = enc(x.half().cuda()).cpu().float()
        enc.check()
        d = (outs[fuse] - want).abs()
        assert d.max().item() < 2e-2 * rng and d.mean().item() < 3e-3 * rng, (fuse, d.max().item(), d.mean().item())
    assert (outs[1] - outs[0]).abs().max().item() < 1e-2 * rng


def test_transformer_long_chunk_ring_attention_matches_oracle():
    """T = 700 tokens (six query blocks: the ring wraps) and a batch of 3, v5.0-style window (127, 128), against the fp32
    oracle of the reference modules."""
    from bonito_amd import synthetic
    from bonito_amd.transformer import Model
    torch.manual_seed(11)
    cfg = synthetic.transformer_model_config(d_model=128, nhead=2, dim_ff=256, depth=2, window=(127, 128), state_len=3,
                                             batchsize=3, chunksize=8400)
    model = Model(cfg).eval()
    nn_ref.round_params_to_half_(model.encoder)
    x = torch.randn(3, 1, 8400).half()
    outs = {}
    for ring in (1, 0):
        enc = HipEncoder(model.encoder, batchsize=3, chunksize=8400)
        enc.set_option("attn_ring", ring)
        outs[ring] = enc(x.cuda()).cpu().float()
        enc.check()
    with torch.no_grad():
        want = nn_ref.forward(model.encoder, x.float(), expand_blanks=False)
    if want.shape != outs[1].shape:
        want = want.permute(1, 0, 2)
    rng = max(want.abs().max().item(), 1.0)
    for ring in (1, 0):
        assert outs[ring].shape == want.shape
        assert (outs[ring] - want).abs().max().item() < 2e-2 * rng, ring
    # the ring kernel's two geometries (round 5: twelve waves = query blocks of 192 is the automatic choice from 384 tokens on; eight =
    # blocks of 128): a query sees the same key tiles in the same order either way - identical bytes
    from bonito_amd import decode
    geo = {}
    try:
        for waves in (8, 12):
            decode.set_option("attn_waves", waves)
            enc = HipEncoder(model.encoder, batchsize=3, chunksize=8400)
            geo[waves] = enc(x.cuda()).cpu()
            enc.check()
    finally:
        decode.set_option("attn_waves", 0)
    assert torch.equal(geo[8], geo[12]) and torch.equal(geo[12].float(), outs[1])


@pytest.mark.parametrize("tile16", [0, 1])
def test_transformer_layers_on_the_four_wave_gemm_match_oracle(tile16):
    """Every linear layer of a v5-width transformer (d_model 512: Wqkv with the rotary epilogue, out_proj / fc2 with and without the fused
    residual, the SwiGLU fc1, the CRF head with scale) forced onto gemm_w4_kernel ("gemm_path" 5: any legal shape; small calls take the
    eight-wave kernels otherwise) - on its 32x32x16 K-tile stream and on the 16x16x32 one ("gemm_tile16") - against the fp32 oracle of
    the reference modules; 2 x 300 tokens: ragged token tiles, one rotary wrap inside a tile."""
    from bonito_amd import decode, synthetic
    from bonito_amd.transformer import Model
    torch.manual_seed(12)
    cfg = synthetic.transformer_model_config(d_model=512, nhead=8, dim_ff=2048, depth=2, window=(127, 128), state_len=3,
                                             batchsize=2, chunksize=1800)
    model = Model(cfg).eval()
    nn_ref.round_params_to_half_(model.encoder)
    x = torch.randn(2, 1, 1800).half()
    with torch.no_grad():
        want = nn_ref.forward(model.encoder, x.float(), expand_blanks=False)
    outs = {}
    try:
        decode.set_option("gemm_tile16", tile16)
        decode.set_option("gemm_path", 5)
        for fuse in (0, 1):
            enc = HipEncoder(model.encoder, batchsize=2, chunksize=1800)
            enc.set_option("norm_fuse", fuse)
            outs[fuse] = enc(x.cuda()).cpu().float()
            enc.check()
        decode.set_option("gemm_path", 2)
        enc = HipEncoder(model.encoder, batchsize=2, chunksize=1800)
        outs["small"] = enc(x.cuda()).cpu().float()
        enc.check()
    finally:
        decode.set_option("gemm_path", 0)
        decode.set_option("gemm_tile16", 1)              # the library's default
    if want.shape != outs[0].shape:
        want = want.permute(1, 0, 2)
    rng = max(want.abs().max().item(), 1.0)
    for key, got in outs.items():
        assert got.shape == want.shape
        d = (got - want).abs()
        assert d.max().item() < 2e-2 * rng and d.mean().item() < 3e-3 * rng, (key, d.max().item(), d.mean().item(), rng)
    assert (outs[0] - outs["small"]).abs().max().item() < 1e-2 * rng


def test_fused_and_unfused_lstm_paths_agree():
    """The ring-in-a-workgroup kernel (3, narrow layers only), the workgroup-shared fused kernel (2), the per-wave fused kernel (1) and the GEMM + recurrence pair (0) are four
    implementations of the same layer: all must match the reference fixture; 0 differs from the fused ones only by
    the fp16 rounding of the intermediate gate tensor."""
    cfg, sd, x, y = load_nn_fixture("lstm96_sl3")
    model = build_model(cfg, sd)
    want = ref_scores_to_koi(y)
    outs = {}
    for fused in (3, 2, 1, 0):
        enc = HipEncoder(model, batchsize=x.shape[0], chunksize=x.shape[-1])
        enc.set_option("lstm_fused", fused)
        outs[fused] = enc(x.half().cuda()).cpu().float()
        enc.check()
        assert (outs[fused] - want).abs().max().item() < TOL_MAX, fused
    assert (outs[0] - outs[1]).abs().max().item() < 2e-2
    assert torch.equal(outs[2], outs[1])      # same arithmetic in the same order: lstm_cell() pins the contractions
    assert torch.equal(outs[3], outs[2])      # ring-in-a-workgroup kernel (H = 96 here): exchange through LDS only
    for fused in (2, 1):
        enc = HipEncoder(model, batchsize=x.shape[0], chunksize=x.shape[-1])
        enc.set_option("lstm_fused", fused)
        enc.set_option("lstm_force_slow", 1)
        slow = enc(x.half().cuda()).cpu().float()
        enc.check()
        assert torch.equal(slow, outs[fused])               # exchange policy never changes the bytes


@pytest.mark.parametrize("H,sl", [(384, 4), (256, 3), (128, 2), (192, 3), (64, 2)])
def test_workgroup_shared_lstm_kernel_widths(H, sl):
    """Every hidden size the workgroup-shared kernel instantiates (U = 12: 96/192/288/384, U = 16: 64/128/256),
    forward and reverse layers, batch not a multiple of 16, against the fp32 oracle of the reference modules."""
    from bonito_amd import nn as bnn, synthetic
    torch.manual_seed(H)
    cfg = synthetic.lstm_crf_encoder_config(H, sl, n_lstm=3)
    model = bnn.from_dict(cfg)
    synthetic.randomise_batchnorm_(model)
    nn_ref.round_params_to_half_(model)
    x = torch.randn(21, 1, 900).half()
    with torch.no_grad():
        want = nn_ref.forward(model, x.float(), expand_blanks=False).permute(1, 0, 2)
    outs = {}
    for fused in (3, 2, 1, -2, -3):
        enc = HipEncoder(model, batchsize=21, chunksize=900)
        enc.set_option("lstm_fused", 2 if fused < 0 else fused)
        if fused == -2:
            enc.set_option("lstm_prefill", 0)
        if fused <= -2:
            enc.set_option("lstm_exchange", 0)         # hand-off through the sentinel-filled output tensor (lstm_layer_fused_kernel / cta)
        outs[fused] = enc(x.cuda()).cpu().float()
        enc.check()
        assert (outs[fused] - want).abs().max().item() < TOL_MAX, fused
    assert torch.equal(outs[2], outs[1]) and torch.equal(outs[3], outs[1]) and torch.equal(outs[-2], outs[2])
    assert torch.equal(outs[-3], outs[2])              # ring-buffer exchange (default) == exchange through the output tensor


@pytest.mark.parametrize("H,batch", [(768, 3), (1024, 37), (640, 33)])
def test_wide_lstm_models(H, batch):
    """H > 512 (768: old-style r9.4.1 width, 1024: v4.3 sup): W_hh does not fit one 16-chunk ring's registers -> the
    stationary-weight kernel with 32-chunk rings (default) and the weight-streaming kernel (lstm_wide = 0); both vs the
    fp32 oracle of the reference modules, and equal to each other up to the fp32 summation order."""
    from bonito_amd import nn as bnn, synthetic
    torch.manual_seed(3)
    cfg = synthetic.lstm_crf_encoder_config(H, 3, n_lstm=2)
    model = bnn.from_dict(cfg)
    synthetic.randomise_batchnorm_(model)
    nn_ref.round_params_to_half_(model)
    x = torch.randn(batch, 1, 600).half()
    with torch.no_grad():
        want = nn_ref.forward(model, x.float(), expand_blanks=False).permute(1, 0, 2)
    outs = {}
    for wide in (1, 0, -1):
        if wide == 0 and H % 64 != 0:
            continue
        enc = HipEncoder(model, batchsize=batch, chunksize=600)
        enc.set_option("lstm_wide", abs(wide))
        if wide < 0:
            enc.set_option("lstm_exchange", 0)      # round-1 hand-off through the sentinel-filled output tensor
        else:
            assert wide == 0 or "lstm_layer_wide_kernel<%d,true>" % (H // 32) in enc.describe()
        outs[wide] = enc(x.cuda()).cpu().float()
        enc.check()
        assert (outs[wide] - want).abs().max().item() < TOL_MAX, wide
    assert torch.equal(outs[1], outs[-1])           # ring-buffer exchange == exchange through the output tensor
    if 0 in outs:      # the streaming kernel adds G after the recurrent product, the stationary one starts from G: fp32 order differs
        assert (outs[1] - outs[0]).abs().max().item() < 2e-3



def test_full_size_hac_encoder_kernel_variants_bit_identical():
    """BASELINE size (batch 512 x chunk 10000, hac widths): the workgroup-shared LSTM kernel that bench.py measures must give
    the same bytes as the per-wave fused kernel and as the write-through exchange policy (size-independent property: all
    variants share the accumulation order and lstm_cell()); the exchange must never time out."""
    from bonito_amd import synthetic
    model = synthetic.make_model("hac", batchsize=512, chunksize=10000)
    x = torch.randn(512, 1, 10000, generator=torch.Generator().manual_seed(25)).half().cuda()
    outs = []
    for fused, slow, prefill, exch in ((3, 0, 1, 1), (1, 0, 1, 1), (3, 1, 1, 1), (3, 0, 0, 0), (3, 0, 1, 0)):
        enc = HipEncoder(model.encoder, batchsize=512, chunksize=10000)
        enc.set_option("lstm_fused", fused)
        enc.set_option("lstm_force_slow", slow)
        enc.set_option("lstm_prefill", prefill)     # 0: sentinel fill inline instead of beside the previous layer's kernel
        enc.set_option("lstm_exchange", exch)       # 1 (default): ring-buffer hand-off, 0: through the output tensor (per-wave fused kernel)
        for _ in range(2):                             # twice: the side-stream fill must also be ordered across calls
            out = enc(x)
        outs.append(out)
        enc.check()
        enc.close()
    assert outs[0].shape == (512, 1667, 1024)
    assert torch.isfinite(outs[0].float()).all()
    assert torch.equal(outs[0], outs[1])
    assert torch.equal(outs[0], outs[2])
    assert torch.equal(outs[0], outs[3])
    assert torch.equal(outs[0], outs[4])


@pytest.mark.parametrize("quantize", [False, True])
def test_ring_exchange_across_xcds_gives_the_same_bytes(quantize):
    """The hand-off of the recurrent kernels must not depend on workgroup -> XCD placement. "lstm_tune" bit 5 spreads the
    eight workgroups of every ring over all eight XCDs: the XCD agreement then selects the write-through policy and every poll
    crosses XCDs (re-used ring-buffer addresses included). Same bytes as the co-located default, no timeout. hac-shaped model at
    a batch that fills the chip, and the wide (1024) kernel."""
    from bonito_amd import synthetic
    cases = [("hac", 512, 2400)] + ([] if quantize else [("sup_lstm", 256, 1200)])
    for name, N, L in cases:
        model = synthetic.make_model(name, batchsize=N, chunksize=L)
        x = torch.randn(N, 1, L, generator=torch.Generator().manual_seed(7)).half().cuda()
        outs = []
        for tune in (0, 32):
            enc = HipEncoder(model.encoder, batchsize=N, chunksize=L, quantize=quantize)
            enc.set_option("lstm_tune", tune)
            outs.append(enc(x))
            enc.check()
            enc.close()
        assert torch.equal(outs[0], outs[1]), name


def _encode(model, x, **options):
    enc = HipEncoder(model, batchsize=x.shape[0], chunksize=x.shape[-1])
    for name, value in options.items():
        enc.set_option(name, value)
    out = enc(x).clone()
    again = enc(x).clone()
    layout = enc.describe()
    enc.check()
    enc.close()
    assert torch.equal(out, again)          # (a data race in the hand-off would show as run-to-run differences)
    return out, layout


@pytest.mark.parametrize("N", [1024, 528, 1008, 640])
def test_paired_rings_same_bytes_as_two_launches(N):
    """Batches of more than 32 rings at H = 384 (N > 512): `lstm_layer_wgx2_kernel` carries two rings per workgroup on one copy of
    the weights, with the gate arithmetic woven into the MFMA stream by hand. Same arithmetic in the same order: the same bytes as
    the single-ring kernel launched twice (`lstm_pair = 0`), for an even ring count, an odd one (a lone ring in the last workgroups),
    one ring beyond a launch (528 = 33 rings) and with the rings of a pair spread over all XCDs."""
    from bonito_amd import synthetic
    model = synthetic.make_model("hac", batchsize=N, chunksize=2400)
    x = torch.randn(N, 1, 2400, generator=torch.Generator().manual_seed(N)).half().cuda()
    two, layout0 = _encode(model.encoder, x, lstm_pair=0)
    one, layout1 = _encode(model.encoder, x)
    assert "lstm_layer_wgx2_kernel<12,3>" in layout1 and "wgx2" not in layout0
    assert torch.equal(one, two)
    scattered, _ = _encode(model.encoder, x, lstm_tune=32)
    assert torch.equal(scattered, two)
    generic, _ = _encode(model.encoder, x, lstm_tune=64)       # bit 6: no unrolled main loop, the generic section code for every step
    assert torch.equal(generic, two)


@pytest.mark.parametrize("L", [30, 48, 54, 60, 66, 72, 78, 102])
def test_paired_rings_main_loop_boundaries(L):
    """The main loop of the paired kernel is unrolled over four steps and runs for 4 <= step, step + 4 <= T - 2; the steps around
    it run the generic section code. T = 5 ... 17 puts the seams everywhere (no main loop at all, exactly one pass, every length
    of the tail): the same bytes as the single-ring kernel launched twice and as the generic code throughout."""
    from bonito_amd import synthetic
    model = synthetic.make_model("hac", batchsize=1024, chunksize=L)
    x = torch.randn(1024, 1, L, generator=torch.Generator().manual_seed(L)).half().cuda()
    two, _ = _encode(model.encoder, x, lstm_pair=0)
    one, layout = _encode(model.encoder, x)
    assert "lstm_layer_wgx2_kernel<12,3>" in layout and one.shape[1] == L // 6
    assert torch.equal(one, two)
    generic, _ = _encode(model.encoder, x, lstm_tune=64)
    assert torch.equal(generic, two)


@pytest.mark.parametrize("H,sl,N", [(288, 3, 1024), (192, 3, 1500), (256, 3, 1500)])
def test_paired_rings_other_widths(H, sl, N):
    """The widths whose gate arithmetic is not hand-woven (generic path of the paired kernel), against two launches and the oracle."""
    from bonito_amd import nn as bnn, synthetic
    torch.manual_seed(H)
    model = bnn.from_dict(synthetic.lstm_crf_encoder_config(H, sl, n_lstm=3))
    synthetic.randomise_batchnorm_(model)
    nn_ref.round_params_to_half_(model)
    x = torch.randn(N, 1, 900, generator=torch.Generator().manual_seed(H)).half()
    one, layout = _encode(model, x.cuda())
    two, _ = _encode(model, x.cuda(), lstm_pair=0)
    assert "wgx2" in layout and torch.equal(one, two)
    rows = [0, N // 2 + 3, N - 1]
    with torch.no_grad():
        want = nn_ref.forward(model, x[rows].float(), expand_blanks=False).permute(1, 0, 2)
    assert (one[rows].cpu().float() - want).abs().max().item() < TOL_MAX


def test_paired_rings_full_size_1024x10000():
    """Two BASELINE batches in one engine call: same bytes as two launches per layer, four chunks against the fp32 oracle, no
    exchange timeout; and with the spin bound at zero the first incomplete poll round raises the device flag (never a hang)."""
    from bonito_amd import decode, synthetic
    model = synthetic.make_model("hac", batchsize=1024, chunksize=10000)
    nn_ref.round_params_to_half_(model)
    x = torch.randn(1024, 1, 10000, generator=torch.Generator().manual_seed(25)).half()
    one, _ = _encode(model.encoder, x.cuda())
    two, _ = _encode(model.encoder, x.cuda(), lstm_pair=0)
    assert one.shape == (1024, 1667, 1024) and torch.equal(one, two)
    rows = [0, 511, 512, 1023]
    with torch.no_grad():
        want = nn_ref.forward(model.encoder, x[rows].float(), expand_blanks=False).permute(1, 0, 2)
    d = (one[rows].cpu().float() - want).abs()
    assert d.max().item() < 2.4e-2 and d.mean().item() < 3e-3, (d.max().item(), d.mean().item())
    enc = HipEncoder(model.encoder, batchsize=1024, chunksize=10000)
    try:
        decode.set_option("lstm_max_spins", 0)
        enc(x.cuda())
        torch.cuda.synchronize()
    finally:
        decode.set_option("lstm_max_spins", -1)
    from bonito_amd import _lib
    with pytest.raises(_lib.HipEngineError):
        enc.check()
    enc.close()


@pytest.mark.parametrize("L", [6, 12, 18, 30, 66])
def test_paired_rings_very_short_chunks(L):
    """One to eleven time steps: the prologue / epilogue of the paired kernel (x-stream prefetch two steps ahead, first poll round one
    section ahead, re-arm two steps behind) must not run past either end. Against two launches of the single-ring kernel."""
    from bonito_amd import synthetic
    model = synthetic.make_model("hac", batchsize=1024, chunksize=L)
    x = torch.randn(1024, 1, L, generator=torch.Generator().manual_seed(L)).half().cuda()
    one, layout = _encode(model.encoder, x)
    two, _ = _encode(model.encoder, x, lstm_pair=0)
    assert "wgx2" in layout and one.shape[1] == (L - 1) // 6 + 1 and torch.equal(one, two)


@pytest.mark.parametrize("kind,N,L", [("hac", 37, 3000), ("hac", 16, 1531), ("fast", 21, 2400), ("hac", 5, 9996), ("fast", 3, 500)])
def test_fused_conv_front_end_equals_three_kernels(kind, N, L):
    """conv1 -> conv2 -> conv3 of the LSTM models as ONE kernel (`conv_front3_kernel`, the 16-channel intermediates stay in LDS;
    default) against the three separate kernels ("conv_fuse" 0): every output is computed by the same operations in the same order,
    positions outside a layer's output are the zeros of the next layer's padding -> identical bytes at the END of the encoder (any
    difference in the convolutions would pass through five recurrent layers), for chunk lengths that end inside a workgroup's span,
    odd batches, a chunk shorter than one workgroup's span, and the 96-wide stack."""
    from bonito_amd import decode, synthetic
    model = synthetic.make_model(kind, batchsize=N, chunksize=L)
    x = torch.randn(N, 1, L, generator=torch.Generator().manual_seed(N + L)).half().cuda()
    try:
        decode.set_option("conv_fuse", 0)
        three, layout3 = _encode(model.encoder, x)
        decode.set_option("conv_fuse", 2)            # 2: also for the 96-channel stack (default 1 fuses the 384-channel one only)
        one, layout1 = _encode(model.encoder, x)
    finally:
        decode.set_option("conv_fuse", 1)
    assert "conv_front3_kernel" in layout1 and "conv_front3_kernel" not in layout3
    assert torch.equal(one, three)
    assert torch.isfinite(one.float()).all() and one.float().abs().max().item() > 0.1


@pytest.mark.parametrize("H,N,L,tune", [(1024, 512, 600, 0), (1024, 288, 300, 0), (768, 544, 300, 0), (1024, 512, 300, 32), (1024, 512, 12, 0)])
def test_wide_layers_calls_of_several_launches(H, N, L, tune):
    """Wide layers (H = 768 / 1024), calls of more than the 8 rings of 32 chunks that one launch of `lstm_layer_wide_kernel` holds: 16
    rings (two full launches), 9 rings (a full launch + a lone ring), 17 rings at H = 768, the workgroups of a ring spread over all XCDs
    (write-through hand-off), and two time steps only. The launches of a layer share the exchange ring buffer (armed once, each launch
    at its ring offset): ring-buffer hand-off == hand-off through the output tensor byte for byte, identical bytes when run twice
    (`_encode`), close to the fp32 oracle. (Round 4 also ran these geometries through a two-rings-per-workgroup variant of the kernel:
    bit-identical, slower, removed - DESIGN 4.)"""
    from bonito_amd import nn as bnn, synthetic
    torch.manual_seed(H + N)
    cfg = synthetic.lstm_crf_encoder_config(H, 3, n_lstm=2)
    model = bnn.from_dict(cfg)
    synthetic.randomise_batchnorm_(model)
    nn_ref.round_params_to_half_(model)
    x = torch.randn(N, 1, L, generator=torch.Generator().manual_seed(N + L)).half()
    opts = {"lstm_tune": tune} if tune else {}
    one, layout = _encode(model, x.cuda(), **opts)
    two, _ = _encode(model, x.cuda(), lstm_exchange=0, **opts)
    assert "lstm_layer_wide_kernel<%d,true>" % (H // 32) in layout
    assert torch.equal(one, two)
    rows = [0, 31, 32, N // 2 + 5, N - 1]
    with torch.no_grad():
        want = nn_ref.forward(model, x[rows].float(), expand_blanks=False).permute(1, 0, 2)
    assert (one[rows].cpu().float() - want).abs().max().item() < TOL_MAX
