"""GPU parity of the operator-level C ABI against fp32 CPU restatements (run with -m gpu on MI355X)."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from bonito_amd import _lib

pytestmark = pytest.mark.gpu
INF = float("inf")


def dev():
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return torch.device("cuda", 0)


def _linear(x, w, bias=None, act=0, scale=1.0, lo=-INF, hi=INF, gated=0, row=(0, 0, 0, 0), out_rows=None):
    M, K = x.shape
    N = w.shape[0]
    ncol = N // 2 if gated else N
    out = torch.zeros((out_rows or M, ncol), dtype=torch.float16, device=x.device)
    b = None if bias is None else bias.float().contiguous()
    _lib.check(_lib.lib().bh_linear(_lib.ptr(x), _lib.ptr(w), _lib.ptr(b), _lib.ptr(out), M, N, K, K, K, ncol,
                                    act, scale, lo, hi, gated, row[0], row[1], row[2], row[3], _lib.stream_ptr()),
               "bh_linear")
    torch.cuda.synchronize()
    return out


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (300, 384, 96), (1000, 1024, 384), (77, 80, 32), (4096, 1536, 384)])
def test_linear_plain(M, N, K):
    g = torch.Generator().manual_seed(M + N + K)
    x = (torch.randn(M, K, generator=g) * 0.5).half()
    w = (torch.randn(N, K, generator=g) * 0.2).half()
    b = torch.randn(N, generator=g)
    want = x.float() @ w.float().T + b
    got = _linear(x.to(dev()), w.to(dev()), b.to(dev())).cpu().float()
    assert (got - want).abs().max().item() < 2e-2 + 2e-3 * want.abs().max().item()


def test_linear_asymmetric_identity():
    """A = I against an asymmetric W catches transposed / permuted fragment layouts exactly."""
    K = N = 128
    x = torch.eye(K).half()
    w = (torch.arange(N * K).reshape(N, K) % 251 / 16.0).half()
    got = _linear(x.to(dev()), w.to(dev())).cpu()
    assert torch.equal(got, w.T.contiguous())


@pytest.mark.parametrize("act,scale,lo,hi", [(2, 5.0, -INF, INF), (0, 1.0, -5.0, 5.0), (1, 1.0, -0.5, 3.5)])
def test_linear_epilogues(act, scale, lo, hi):
    g = torch.Generator().manual_seed(3)
    x = torch.randn(200, 96, generator=g).half()
    w = (torch.randn(256, 96, generator=g) * 0.3).half()
    z = x.float() @ w.float().T
    z = {0: z, 1: z * torch.sigmoid(z), 2: torch.tanh(z)}[act] * scale
    want = z.clamp(lo, hi)
    got = _linear(x.to(dev()), w.to(dev()), act=act, scale=scale, lo=lo, hi=hi).cpu().float()
    assert (got - want).abs().max().item() < 1.5e-2


def test_linear_row_remap_drops_padding():
    """TNC rows -> NTC output with padded batch rows skipped (engine's final CRF linear)."""
    T, Np, Nv, K, Cc = 7, 16, 11, 64, 64
    g = torch.Generator().manual_seed(5)
    x = torch.randn(T * Np, K, generator=g).half()
    w = (torch.randn(Cc, K, generator=g) * 0.2).half()
    want = (x.float() @ w.float().T).view(T, Np, Cc)[:, :Nv].permute(1, 0, 2).reshape(Nv * T, Cc)
    got = _linear(x.to(dev()), w.to(dev()), row=(Np, 1, T, Nv), out_rows=Nv * T).cpu().float()
    assert (got - want).abs().max().item() < 2e-2


def test_linear_gated_swiglu():
    g = torch.Generator().manual_seed(6)
    x = torch.randn(130, 64, generator=g).half()
    w = (torch.randn(256, 64, generator=g) * 0.3).half()     # rows already interleaved (y0, g0, y1, g1, ...)
    z = x.float() @ w.float().T
    y, gate = z[:, 0::2], z[:, 1::2]
    want = y * gate * torch.sigmoid(gate)
    got = _linear(x.to(dev()), w.to(dev()), gated=1).cpu().float()
    assert (got - want).abs().max().item() < 2e-2


def test_linear_rejects_bad_shapes():
    x = torch.zeros(4, 12, dtype=torch.float16, device=dev())
    w = torch.zeros(4, 12, dtype=torch.float16, device=dev())
    out = torch.zeros(4, 8, dtype=torch.float16, device=dev())
    rc = _lib.lib().bh_linear(_lib.ptr(x), _lib.ptr(w), None, _lib.ptr(out), 4, 4, 12, 12, 12, 8, 0, 1.0, -INF, INF,
                              0, 0, 0, 0, 0, _lib.stream_ptr())
    assert rc != 0 and "multiples of 8" in _lib.last_error()


@pytest.mark.parametrize("Cout,K,stride,act", [(16, 5, 1, 1), (64, 5, 1, 1), (344, 9, 3, 0), (8, 19, 6, 2), (4, 5, 1, 1), (6, 3, 2, 0)])
def test_conv_first(Cout, K, stride, act):
    g = torch.Generator().manual_seed(Cout)
    N, L = 3, 1000
    x = torch.randn(N, L, generator=g).half()
    w = torch.randn(Cout, 1, K, generator=g) * 0.4
    b = torch.randn(Cout, generator=g) * 0.1
    z = F.conv1d(x.float()[:, None], w, b, stride=stride, padding=K // 2)
    want = {0: z, 1: z * torch.sigmoid(z), 2: torch.tanh(z)}[act].permute(0, 2, 1)   # [N][Lout][C]
    Lout = want.shape[1]
    out = torch.zeros((N, Lout, Cout), dtype=torch.float16, device=dev())
    wd, bd = w.reshape(Cout, K).contiguous().to(dev()), b.to(dev())
    xd = x.to(dev())
    _lib.check(_lib.lib().bh_conv1d_first(_lib.ptr(xd), _lib.ptr(wd), _lib.ptr(bd), _lib.ptr(out), N, L, Cout,
                                          K, stride, K // 2, act, -INF, INF, Lout * Cout, Cout, _lib.stream_ptr()),
               "conv1d_first")
    torch.cuda.synchronize()
    assert (out.cpu().float() - want).abs().max().item() < 1e-2


def _pack_conv(w):
    Cout, Cin, K = w.shape
    lib = _lib.lib()
    n = lib.bh_conv1d_packed_halves(Cin, Cout, K)
    pk = np.zeros(n, np.uint16)
    wf = np.ascontiguousarray(w.numpy().astype(np.float32))
    _lib.check(lib.bh_conv1d_pack(wf.ctypes.data_as(C.c_void_p), Cin, Cout, K, pk.ctypes.data_as(C.c_void_p)), "pack")
    return torch.from_numpy(pk.view(np.int16)).to(dev())


@pytest.mark.parametrize("Cin,Cout,K,stride,L,tnc", [
    (16, 16, 5, 1, 1000, False), (16, 96, 19, 6, 1200, True), (16, 384, 19, 6, 3000, True),
    (64, 128, 9, 3, 900, False), (128, 128, 9, 2, 700, False), (128, 512, 5, 2, 333, False), (8, 20, 3, 1, 50, False)])
def test_conv_igemm(Cin, Cout, K, stride, L, tnc):
    g = torch.Generator().manual_seed(Cin * Cout + K)
    N = 3
    x = (torch.randn(N, Cin, L, generator=g) * 0.7).half()
    w = (torch.randn(Cout, Cin, K, generator=g) * (1.0 / (Cin * K) ** 0.5)).half().float()
    b = torch.randn(Cout, generator=g) * 0.1
    z = F.conv1d(x.float(), w, b, stride=stride, padding=K // 2)
    want = (z * torch.sigmoid(z)).clamp(-0.5, 3.5)
    Lout = want.shape[-1]
    xin = x.permute(0, 2, 1).contiguous().to(dev())                       # channel-minor [N][L][Cin]
    if tnc:
        out = torch.zeros((Lout, N, Cout), dtype=torch.float16, device=dev())
        os_n, os_t = Cout, N * Cout
        want = want.permute(2, 0, 1)
    else:
        out = torch.zeros((N, Lout, Cout), dtype=torch.float16, device=dev())
        os_n, os_t = Lout * Cout, Cout
        want = want.permute(0, 2, 1)
    wpk, bd = _pack_conv(w), b.to(dev())      # keep the device buffers alive across the async launch
    _lib.check(_lib.lib().bh_conv1d(_lib.ptr(xin), _lib.ptr(wpk), _lib.ptr(bd), _lib.ptr(out), N, L,
                                    Cin, Cout, K, stride, K // 2, 1, -0.5, 3.5, os_n, os_t, _lib.stream_ptr()), "conv1d")
    torch.cuda.synchronize()
    assert (out.cpu().float() - want).abs().max().item() < 1.5e-2


@pytest.mark.parametrize("tnc,Cout", [(True, 384), (False, 384), (True, 96)])
def test_conv_weight_stationary_kernel_equals_generic(tnc, Cout):
    """conv3 shapes (16 -> 384 / 96 channels, 19 taps, stride 6): the weight-stationary kernel (default) and the generic
    implicit-GEMM kernel (bh_set_option "conv_ws" = 0) accumulate in the same order -> identical bytes; ragged last block."""
    from bonito_amd import decode
    g = torch.Generator().manual_seed(5)
    N, Cin, K, stride, L = 5, 16, 19, 6, 4000
    x = (torch.randn(N, L, Cin, generator=g) * 0.7).half().to(dev())
    w = (torch.randn(Cout, Cin, K, generator=g) * 0.06).half().float()
    wpk, bd = _pack_conv(w), (torch.randn(Cout, generator=g) * 0.1).to(dev())
    Lout = (L + 2 * (K // 2) - K) // stride + 1
    os_n, os_t = (Cout, N * Cout) if tnc else (Lout * Cout, Cout)
    outs = []
    try:
        for ws in (1, 0):
            decode.set_option("conv_ws", ws)
            out = torch.zeros(N * Lout * Cout, dtype=torch.float16, device=dev())
            _lib.check(_lib.lib().bh_conv1d(_lib.ptr(x), _lib.ptr(wpk), _lib.ptr(bd), _lib.ptr(out), N, L, Cin, Cout, K, stride,
                                            K // 2, 1, -0.5, 3.5, os_n, os_t, _lib.stream_ptr()), "conv1d")
            torch.cuda.synchronize()
            outs.append(out.cpu())
    finally:
        decode.set_option("conv_ws", 1)
    assert torch.equal(outs[0], outs[1])
    assert outs[0].float().abs().max().item() > 0.1


@pytest.mark.parametrize("Cin,Cout,K,stride,L", [(64, 64, 5, 1, 1300), (64, 128, 9, 3, 2000), (128, 128, 9, 2, 1111), (128, 512, 5, 2, 900)])
def test_conv_feature_split_instances_equal_position_split(Cin, Cout, K, stride, L):
    """The convolutions of the v5 transformer models (multiples of 64 output channels): conv_igemm_kernel's feature-split instances
    (four waves = four feature tiles, each covering all positions of the workgroup; default) and the position-split ones
    ("conv_fs" 0) accumulate every output in the same order -> identical bytes, for every LDS budget (positions per workgroup 64,
    128, 256) and a ragged last block."""
    from bonito_amd import decode
    g = torch.Generator().manual_seed(Cin + Cout + K)
    N = 3
    x = (torch.randn(N, L, Cin, generator=g) * 0.7).half().to(dev())
    w = (torch.randn(Cout, Cin, K, generator=g) * (1.0 / (Cin * K) ** 0.5)).half().float()
    wpk, bd = _pack_conv(w), (torch.randn(Cout, generator=g) * 0.1).to(dev())
    Lout = (L + 2 * (K // 2) - K) // stride + 1
    outs = []
    try:
        for fs, kb in ((1, 64), (0, 64), (1, 150), (1, 24)):
            decode.set_option("conv_fs", fs)
            decode.set_option("conv_lds_kb", kb)
            out = torch.zeros(N * Lout * Cout, dtype=torch.float16, device=dev())
            _lib.check(_lib.lib().bh_conv1d(_lib.ptr(x), _lib.ptr(wpk), _lib.ptr(bd), _lib.ptr(out), N, L, Cin, Cout, K, stride,
                                            K // 2, 1, -0.5, 3.5, Lout * Cout, Cout, _lib.stream_ptr()), "conv1d")
            torch.cuda.synchronize()
            outs.append(out.cpu())
    finally:
        decode.set_option("conv_fs", 1)
        decode.set_option("conv_lds_kb", 0)
    assert all(torch.equal(outs[0], o) for o in outs[1:])
    z = F.conv1d(x.cpu().float().permute(0, 2, 1), w, bd.cpu(), stride=stride, padding=K // 2)
    want = (z * torch.sigmoid(z)).clamp(-0.5, 3.5).permute(0, 2, 1).reshape(-1)
    assert (outs[0].float() - want).abs().max().item() < 1.5e-2


def _lstm_ref(G, Whh, reverse):
    T, N, H4 = G.shape
    H = H4 // 4
    h = torch.zeros(N, H)
    c = torch.zeros(N, H)
    out = torch.empty(T, N, H)
    for t in (range(T - 1, -1, -1) if reverse else range(T)):
        g = G[t] + h @ Whh.T
        i, f, gg, o = g.chunk(4, -1)
        c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(gg)
        h = torch.sigmoid(o) * torch.tanh(c)
        out[t] = h
        h = h.half().float()        # the engine publishes h in fp16
    return out


@pytest.mark.parametrize("H,N,T,reverse,flags", [(32, 16, 40, 0, 0), (96, 48, 60, 1, 0), (384, 64, 50, 0, 0),
                                                 (384, 512, 30, 1, 0), (128, 16, 33, 1, 0), (512, 32, 20, 0, 0),
                                                 (96, 48, 60, 0, 1), (384, 512, 30, 0, 1),
                                                 (1024, 64, 12, 1, 0), (768, 32, 10, 0, 0), (128, 48, 20, 0, 2), (1024, 512, 6, 0, 0)])
def test_lstm_layer(H, N, T, reverse, flags):
    g = torch.Generator().manual_seed(H + N)
    G = (torch.randn(T, N, 4 * H, generator=g) * 1.5).half()
    Whh = (torch.randn(4 * H, H, generator=g) * (1.0 / H ** 0.5)).half().float()
    want = _lstm_ref(G.float(), Whh, reverse)
    lib = _lib.lib()
    pk = np.zeros(4 * H * H, np.uint16)
    wf = np.ascontiguousarray(Whh.numpy())
    _lib.check(lib.bh_lstm_pack_whh(wf.ctypes.data_as(C.c_void_p), H, pk.ctypes.data_as(C.c_void_p)), "pack_whh")
    pkd = torch.from_numpy(pk.view(np.int16)).to(dev())
    h = torch.zeros((T, N, H), dtype=torch.float16, device=dev())
    err = torch.zeros(1, dtype=torch.int32, device=dev())
    Gd = G.to(dev())
    ws = torch.zeros(lib.bh_lstm_workspace(N, H), dtype=torch.uint8, device=dev())
    _lib.check(lib.bh_lstm_layer(_lib.ptr(Gd), _lib.ptr(pkd), _lib.ptr(h), T, N, H, reverse, _lib.ptr(ws), _lib.ptr(err),
                                 flags, _lib.stream_ptr()), "lstm_layer")
    torch.cuda.synchronize()
    assert err.item() == 0, "persistent LSTM kernel hit its spin bound"
    d = (h.cpu().float() - want).abs()
    assert d.max().item() < 6e-3, d.max().item()


def test_lstm_layer_is_deterministic_and_restartable():
    """Same call twice (buffer re-poisoned by the ABI) -> identical bytes."""
    H, N, T = 96, 32, 25
    g = torch.Generator().manual_seed(1)
    G = torch.randn(T, N, 4 * H, generator=g).half().to(dev())
    Whh = (torch.randn(4 * H, H, generator=g) * 0.1).numpy()
    lib = _lib.lib()
    pk = np.zeros(4 * H * H, np.uint16)
    lib.bh_lstm_pack_whh(np.ascontiguousarray(Whh).ctypes.data_as(C.c_void_p), H, pk.ctypes.data_as(C.c_void_p))
    pkd = torch.from_numpy(pk.view(np.int16)).to(dev())
    outs = []
    err = torch.zeros(1, dtype=torch.int32, device=dev())
    ws = torch.zeros(lib.bh_lstm_workspace(N, H), dtype=torch.uint8, device=dev())
    for flags in (0, 1):      # fast (same-XCD) and placement-independent exchange policies agree bit for bit
        h = torch.zeros((T, N, H), dtype=torch.float16, device=dev())
        _lib.check(lib.bh_lstm_layer(_lib.ptr(G), _lib.ptr(pkd), _lib.ptr(h), T, N, H, 0, _lib.ptr(ws), _lib.ptr(err),
                                     flags, _lib.stream_ptr()))
        torch.cuda.synchronize()
        outs.append(h.cpu())
    assert err.item() == 0 and torch.equal(outs[0], outs[1])


# ---- transformer operators -------------------------------------------------------------------------
def _rot_table(T, dim=64):
    lib = _lib.lib()
    tab = np.zeros((T, dim // 2, 2), np.float32)
    _lib.check(lib.bh_rotary_table(T, dim, tab.ctypes.data_as(C.c_void_p)), "rotary_table")
    return torch.from_numpy(tab)


@pytest.mark.parametrize("T,H,win", [(100, 2, (31, 32)), (1000, 8, (127, 128)), (333, 1, (5, 0)), (130, 2, (0, 7)),
                                     (64, 1, (200, 200))])
def test_attention_matches_sdpa_restatement(T, H, win):
    from oracle import nn_ref
    g = torch.Generator().manual_seed(T + H)
    N, D = 2, 64 * H
    qkv = (torch.randn(N, T, 3, H, 64, generator=g) * 0.8).half()
    want_qkv = nn_ref.rotary(qkv.float())
    q, k, v = (want_qkv[:, :, i].permute(0, 2, 1, 3) for i in range(3))
    s = (q @ k.transpose(-1, -2)) / 8.0
    s = s.masked_fill(~nn_ref.window_mask(T, win), float("-inf"))
    want = (torch.softmax(s, -1) @ v).permute(0, 2, 1, 3).reshape(N, T, D)
    tab = _rot_table(T).to(dev())
    qd = qkv.reshape(N * T, 3 * D).contiguous().to(dev())
    out = torch.zeros((N * T, D), dtype=torch.float16, device=dev())
    _lib.check(_lib.lib().bh_attention(_lib.ptr(qd), _lib.ptr(out), _lib.ptr(tab), N, T, H, 64, win[0], win[1],
                                       _lib.stream_ptr()), "attention")
    torch.cuda.synchronize()
    d = (out.cpu().float().view(N, T, D) - want).abs()
    assert d.max().item() < 8e-3, d.max().item()


@pytest.mark.parametrize("N,T,H,win", [(2, 1000, 8, (127, 128)), (70, 333, 8, (127, 128)), (45, 200, 2, (40, 17)), (3, 17, 1, (127, 128)),
                                       (33, 1000, 8, (128, 128)), (130, 400, 8, (0, 0)), (9, 1667, 8, (127, 128)), (65, 96, 4, (127, 128))])
def test_ring_attention_on_prerotated_qkv_matches_a_device_restatement(N, T, H, win):
    """bh_attention_prerotated = the persistent ring kernel the engine runs (round 6: ONE stream per workgroup - with more chunks than the
    device has workgroup columns (256 CUs / heads) a workgroup walks several chunks back to back through one ring, tiles are classified
    empty / full / partial per wave, V^T rows are permuted). Every output row against softmax(base-2 scores in the window) V computed by
    torch on the device; both kernel versions (`attn_version` 1 = rounds 2-5, 2 = round 6) and both wave counts. The shapes put chunk
    boundaries inside blocks of 192 / 128 queries (T = 333, 200, 96, 17), use other windows than the fast path's, and more chunks than
    columns (N = 70, 130 at 8 heads -> 3 and 5 chunks per workgroup)."""
    from bonito_amd import decode
    D = 64 * H
    g = torch.Generator(device=dev()).manual_seed(N * T + H)
    qkv = (torch.randn(N * T, 3 * D, generator=g, device=dev()) * 0.7).half()
    x = qkv.float().view(N, T, 3, H, 64)
    q, k, v = (x[:, :, i].permute(0, 2, 1, 3) for i in range(3))
    i = torch.arange(T, device=dev())[:, None]
    j = torch.arange(T, device=dev())[None, :]
    mask = (j >= i - win[0]) & (j <= i + win[1])
    want = torch.empty((N, T, D), device=dev())
    for lo in range(0, N, 16):                            # in slabs: the [N, H, T, T] score tensor of the big shapes would not fit
        sc = (q[lo:lo + 16] @ k[lo:lo + 16].transpose(-1, -2)) * 0.6931471805599453          # q carries log2(e) / sqrt(d)
        sc = sc.masked_fill(~mask, float("-inf"))
        want[lo:lo + 16] = (torch.softmax(sc, -1) @ v[lo:lo + 16]).permute(0, 2, 1, 3).reshape(-1, T, D)
    try:
        for version, waves in ((2, 0), (2, 8), (2, 12), (1, 0)):
            decode.set_option("attn_version", version)
            decode.set_option("attn_waves", waves)
            out = torch.full((N * T, D), float("nan"), dtype=torch.float16, device=dev())
            _lib.check(_lib.lib().bh_attention_prerotated(_lib.ptr(qkv), _lib.ptr(out), N, T, H, 64, win[0], win[1], _lib.stream_ptr()),
                       "attention_prerotated")
            torch.cuda.synchronize()
            d = (out.float().view(N, T, D) - want).abs()
            assert not torch.isnan(d).any() and d.max().item() < 4e-3, (version, waves, d.max().item())
    finally:
        decode.set_option("attn_version", 2)
        decode.set_option("attn_waves", 0)


def test_rotary_table_matches_flash_attn_convention():
    tab = _rot_table(50)
    inv = 1.0 / (10000 ** (torch.arange(0, 64, 2).float() / 64))
    ang = torch.outer(torch.arange(50, dtype=torch.float32), inv)
    assert (tab[..., 0] - ang.cos()).abs().max() < 1e-5 and (tab[..., 1] - ang.sin()).abs().max() < 1e-5


@pytest.mark.parametrize("M,D", [(1000, 512), (7, 128), (300, 1024), (5, 64)])
def test_rmsnorm_residual(M, D):
    g = torch.Generator().manual_seed(M)
    a = torch.randn(M, D, generator=g).half()
    x = torch.randn(M, D, generator=g).half()
    w = 1 + 0.1 * torch.randn(D, generator=g)
    alpha = 2.4494897
    z = a.float() + alpha * x.float()
    want = z * torch.rsqrt(z.pow(2).mean(-1, keepdim=True) + 1e-5) * w
    ad, xd, wd = a.to(dev()), x.to(dev()), w.to(dev())
    out = torch.zeros((M, D), dtype=torch.float16, device=dev())
    _lib.check(_lib.lib().bh_rmsnorm_residual(_lib.ptr(ad), _lib.ptr(xd), _lib.ptr(wd), _lib.ptr(out), M, D, alpha, 1e-5,
                                              _lib.stream_ptr()), "rmsnorm")
    torch.cuda.synchronize()
    assert (out.cpu().float() - want).abs().max().item() < 4e-3


@pytest.mark.parametrize("M,N,K,gated", [(16384 + 37, 2048, 128, 0), (40000, 1024, 384, 0), (33000, 1024, 64, 1), (70000, 512, 512, 0),
                                         (36000, 1024, 192, 1), (131072 + 5, 256, 2048, 0)])
def test_linear_persistent_big_tile_kernel(M, N, K, gated):
    """Problems with >= 512 tiles of 256 x 256 and K % 64 == 0 run on the persistent 256x256x64 kernel: same K order and
    epilogue as the 128-tile kernels, so the bytes must be identical (and right) - also with the staggered start ("gemm_stagger");
    short and long K-tile streams, ragged last token tile."""
    from bonito_amd import decode
    g = torch.Generator().manual_seed(M)
    x = (torch.randn(M, K, generator=g) * 0.5).half().to(dev())
    w = (torch.randn(N, K, generator=g) * 0.2).half().to(dev())
    b = torch.randn(N, generator=g).to(dev())
    try:
        decode.set_option("gemm_path", 3)            # (0 = automatic would pick the four-wave kernel where it applies: its own test below)
        big = _linear(x, w, b, act=0 if gated else 1, gated=gated)
        decode.set_option("gemm_path", 2)
        small = _linear(x, w, b, act=0 if gated else 1, gated=gated)
        decode.set_option("gemm_path", 3)
        decode.set_option("gemm_stagger", 3)
        again = _linear(x, w, b, act=0 if gated else 1, gated=gated)
    finally:
        decode.set_option("gemm_path", 0)
        decode.set_option("gemm_stagger", 0)
    assert torch.equal(big, small)
    assert torch.equal(again, big)
    z = x[-3000:].float() @ w.float().T + b
    want = (z[:, 0::2] * z[:, 1::2] * torch.sigmoid(z[:, 1::2])) if gated else z * torch.sigmoid(z)
    assert (big[-3000:].float() - want).abs().max().item() < 3e-2 + 3e-3 * want.abs().max().item()


TILE16_DEFAULT = 1        # the library's default MFMA shape of the four-wave GEMM ("gemm_tile16")


def _linear_ref(x, w, b, act, gated, scale=1.0, lo=-INF, hi=INF):
    z = x.float() @ w.float().T
    if b is not None:
        z = z + b.float()
    if gated:
        return z[:, 0::2] * z[:, 1::2] * torch.sigmoid(z[:, 1::2])
    z = {0: z, 1: z * torch.sigmoid(z), 2: torch.tanh(z), 3: torch.relu(z)}[act] * scale
    return z.clamp(lo, hi)


@pytest.mark.parametrize("tile16", [0, 1])
@pytest.mark.parametrize("M,N,K,gated,act", [(300, 256, 640, 0, 0), (1000 + 7, 512, 384, 0, 1), (2048, 256, 512, 1, 0), (777, 768, 1024, 0, 2),
                                             (40000 + 13, 1024, 384, 0, 1), (70000, 512, 512, 1, 0), (131072 + 5, 256, 2048, 0, 0)])
def test_linear_four_wave_kernel(M, N, K, gated, act, tile16):
    """gemm_w4_kernel (256 x 256 x 64 tile on four waves, 32x32x16 or 16x16x32 MFMAs, one generated instruction stream per K-tile, outputs
    through an LDS scratch into full-line stores) serves K % 128 == 0, N % 256 == 0 problems of >= 512 tiles ("gemm_path" 5: any
    legal shape). EVERY output row against the fp32 restatement on the device (it accumulates K in another order than the
    16x16x32 kernels, so the bar is the tolerance of test_linear_plain, not bit equality with them), closeness to the 128-tile
    kernel, and identical bytes when run twice; shortest (4) and long (32) K-tile streams, one workgroup walking several output
    tiles, ragged last token tile, SwiGLU / swish / tanh epilogues with bias. `tile16`: the same kernel around the 16x16x32 K-tile
    stream ("gemm_tile16", the default: 64 float4 accumulators, W rows staged in natural order, bias / activation / SwiGLU / scale-clamp
    applied in the accumulator layout and fp16 through the LDS scratch)."""
    from bonito_amd import decode
    g = torch.Generator().manual_seed(M + K)
    x = (torch.randn(M, K, generator=g) * 0.5).half().to(dev())
    w = (torch.randn(N, K, generator=g) * 0.2).half().to(dev())
    b = torch.randn(N, generator=g).to(dev())
    try:
        decode.set_option("gemm_tile16", tile16)
        decode.set_option("gemm_path", 5)
        got = _linear(x, w, b, act=act, gated=gated)
        again = _linear(x, w, b, act=act, gated=gated)
        decode.set_option("gemm_path", 2)
        small = _linear(x, w, b, act=act, gated=gated)
    finally:
        decode.set_option("gemm_path", 0)
        decode.set_option("gemm_tile16", TILE16_DEFAULT)
    assert torch.equal(got, again)
    worst = 0.0
    for lo_ in range(0, M, 16384):                     # fp32 restatement on the device, every row
        want = _linear_ref(x[lo_:lo_ + 16384], w, b, act, gated)
        d = (got[lo_:lo_ + 16384].float() - want).abs().max().item()
        worst = max(worst, d / (2e-2 + 2e-3 * want.abs().max().item()))
    assert worst < 1.0, worst
    assert (got.float() - small.float()).abs().max().item() < 2e-2 + 2e-3 * small.float().abs().max().item()


def test_linear_four_wave_kernel_work_order_does_not_change_bytes():
    """"gemm_order" (feature groups or token blocks fastest inside an XCD's share) and "gemm_gf" (feature tiles per block) only change which
    workgroup computes which output tile when: identical bytes, ragged token edge and a feature-tile count (6) that 4 does not divide included."""
    from bonito_amd import decode
    g = torch.Generator().manual_seed(77)
    M, N, K = 20000 + 9, 1536, 512
    x = (torch.randn(M, K, generator=g) * 0.5).half().to(dev())
    w = (torch.randn(N, K, generator=g) * 0.2).half().to(dev())
    outs = []
    try:
        decode.set_option("gemm_path", 5)
        for order, gf in ((1, 0), (0, 0), (1, 1), (0, 8), (1, 2)):
            decode.set_option("gemm_order", order)
            decode.set_option("gemm_gf", gf)
            outs.append(_linear(x, w))
    finally:
        decode.set_option("gemm_path", 0)
        decode.set_option("gemm_order", 1)
        decode.set_option("gemm_gf", 0)
    for o in outs[1:]:
        assert torch.equal(o, outs[0])
    want = _linear_ref(x[:4096], w, None, 0, 0)
    assert (outs[0][:4096].float() - want).abs().max().item() < 2e-2 + 2e-3 * want.abs().max().item()


@pytest.mark.parametrize("tile16", [0, 1])
def test_linear_four_wave_kernel_layouts_exactly(tile16):
    """X = I-like (one 1.0 per row) against an asymmetric W: every output is one W element exactly, so a permuted fragment row, a
    wrong swizzle or a transposed store shows up as a wrong value, not as noise. K = 384 (the shortest K the kernel takes), several feature and token tiles."""
    from bonito_amd import decode
    M, N, K = 1024, 512, 384
    x = torch.zeros(M, K)
    x[torch.arange(M), (torch.arange(M) * 7) % K] = 1.0
    w = ((torch.arange(N * K).reshape(N, K) * 37) % 2039 / 16.0).half()
    want = w.float()[:, (torch.arange(M) * 7) % K].T.contiguous().half()
    try:
        decode.set_option("gemm_tile16", tile16)
        decode.set_option("gemm_path", 5)
        got = _linear(x.half().to(dev()), w.to(dev())).cpu()
    finally:
        decode.set_option("gemm_path", 0)
        decode.set_option("gemm_tile16", TILE16_DEFAULT)
    assert torch.equal(got, want)


@pytest.mark.parametrize("tile16", [0, 1])
def test_linear_four_wave_kernel_row_remap_scale_clamp(tile16):
    """The CRF head's call shape on the four-wave kernel: (t, n)-major rows -> [N][T][C] with the padded batch rows dropped, tanh,
    scale 5, clamp."""
    from bonito_amd import decode
    T, Np, Nv, K, Cc = 70, 32, 27, 384, 1024
    g = torch.Generator().manual_seed(8)
    x = torch.randn(T * Np, K, generator=g).half()
    w = (torch.randn(Cc, K, generator=g) * 0.1).half()
    z = torch.tanh(x.float() @ w.float().T) * 5.0
    want = z.clamp(-4.5, 4.5).view(T, Np, Cc)[:, :Nv].permute(1, 0, 2).reshape(Nv * T, Cc)
    try:
        decode.set_option("gemm_tile16", tile16)
        decode.set_option("gemm_path", 5)
        got = _linear(x.to(dev()), w.to(dev()), act=2, scale=5.0, lo=-4.5, hi=4.5, row=(Np, 1, T, Nv), out_rows=Nv * T).cpu().float()
    finally:
        decode.set_option("gemm_path", 0)
        decode.set_option("gemm_tile16", TILE16_DEFAULT)
    assert (got - want).abs().max().item() < 2e-2
