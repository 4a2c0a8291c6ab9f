"""
CPU ORACLE -- TEST INFRASTRUCTURE ONLY.  Plain PyTorch-CPU fp32 restatement of the encoder forward of
``bonito.nn`` / ``bonito.crf`` / ``bonito.transformer`` module trees (/root/reference bonito/nn.py:
Convolution 222-241, BatchNorm 192-198, LSTM/RNNWrapper 353-415, LinearCRFEncoder 269-298, Clamp 60-67,
Permute 331-338, LinearUpsample 140-159, Serial 77-89; bonito/transformer/model.py:42-79,116-128).

It walks either a reference module tree or a ``bonito_amd.nn`` parameter-container tree (they share
attribute names) and evaluates it with explicit tensor algebra.  Pinned against the reference's own
``bonito/nn.py`` executed by PyTorch-CPU: tests/golden/make_golden.py runs both on seeded weights and
commits the vectors; tests/test_oracle_nn.py re-checks this file against them on every run.

Two modes. ``forward(m, x)`` is the fp32 CPU path (the parity target). ``forward(m, x, fp16=True)`` is the SAME graph with
every value rounded to fp16 exactly where the HIP engine stores fp16 (BatchNorm folded in fp32 and the folded
weights of the MFMA convolutions rounded once; every convolution output; every published h_t of a recurrent layer; packed qkv after
the rotary epilogue, the attention probabilities that feed the P.V product and its output, every linear-layer output; the scores) -
accumulation stays fp32 in PyTorch's order with libm transcendentals. Comparing the engine with BOTH separates what fp16 STORAGE
costs (fp32 mode minus fp16 mode) from what is left - summation order inside the MFMA tiles and the hardware exp / rcp
(engine minus fp16 mode). It is a diagnostic restatement of the engine's storage points, not a second definition of the reference.
"""
import math

import torch
import torch.nn.functional as F


def _name(m):
    return getattr(m, "name", type(m).__name__.lower())


def _act(m, x):
    if m is None:
        return x
    n = _name(m)
    if n in ("swish", "silu"):
        return x * torch.sigmoid(x)
    if n == "tanh":
        return torch.tanh(x)
    if n == "relu":
        return torch.relu(x)
    raise NotImplementedError(n)


def _h(x):
    """Round to the nearest fp16 value, keep fp32 storage."""
    return x.half().float()


def conv_forward(m, x, fp16=False):
    c = m.conv
    if fp16:
        # the engine's arithmetic: BatchNorm folded into weight and bias in fp32 (bonito_amd.nn.Convolution.folded, reference
        # nn.py:447-454), the folded weights of the MFMA convolutions (Cin > 1) rounded to fp16 once, the bias kept in fp32, fp32
        # accumulation, the activation in fp32, ONE rounding of the output
        w = c.weight.float()
        b = torch.zeros(w.shape[0]) if c.bias is None else c.bias.float()
        if m.norm is not None:
            bn = m.norm.bn
            sc = torch.rsqrt(bn.running_var.float() + bn.eps)
            if bn.affine:
                sc = sc * bn.weight.float()
            w = w * sc[:, None, None]
            b = (b - bn.running_mean.float()) * sc + (bn.bias.float() if bn.affine else 0.0)
        if c.in_channels > 1:
            w = _h(w)
        h = F.conv1d(x, w, b, stride=c.stride, padding=c.padding, dilation=c.dilation, groups=c.groups)
        return _h(_act(m.activation, h))
    h = F.conv1d(x, c.weight.float(), None if c.bias is None else c.bias.float(), stride=c.stride,
                 padding=c.padding, dilation=c.dilation, groups=c.groups)
    if m.norm is not None:
        bn = m.norm.bn
        h = (h - bn.running_mean.float()[None, :, None]) * torch.rsqrt(bn.running_var.float()[None, :, None] + bn.eps)
        if bn.affine:
            h = h * bn.weight.float()[None, :, None] + bn.bias.float()[None, :, None]
    return _act(m.activation, h)


def lstm_forward(m, x, fp16=False):
    """x [T,N,I] -> [T,N,H]; gates i,f,g,o; h0=c0=0; `reverse` runs time backwards. fp16: h_t is rounded to fp16 when it is
    published (it feeds the next step AND the next layer as fp16); the cell state and the pre-activations stay fp32."""
    r = m.rnn
    W_ih, W_hh = r.weight_ih_l0.float(), r.weight_hh_l0.float()
    b = 0
    if r.bias:
        b = r.bias_ih_l0.float() + r.bias_hh_l0.float()
    T, N, _ = x.shape
    H = r.hidden_size
    h = torch.zeros(N, H)
    c = torch.zeros(N, H)
    out = torch.empty(T, N, H)
    steps = range(T - 1, -1, -1) if m.reverse else range(T)
    gx = x @ W_ih.T + b
    if fp16 and H > 512:
        gx = _h(gx)          # layers wider than 512: the engine's input projection is a GEMM of its own whose output (bias included) is stored in fp16
    for t in steps:
        g = gx[t] + h @ W_hh.T
        i, f, gg, o = g.chunk(4, dim=-1)
        c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(gg)
        h = torch.sigmoid(o) * torch.tanh(c)
        if fp16:
            h = _h(h)
        out[t] = h
    return out


def crf_encoder_forward(m, x, expand_blanks=None, fp16=False):
    """fp16: the scores are stored by the caller's final rounding (a following Clamp acts on the fp32 value in the engine's epilogue,
    so nothing is rounded here unless this layer is the last one - see `forward`)."""
    if m.permute is not None:
        x = x.permute(*m.permute)
    s = x @ m.linear.weight.float().T
    if m.linear.bias is not None:
        s = s + m.linear.bias.float()
    s = _act(m.activation, s)
    if m.scale is not None:
        s = s * m.scale
    expand = m.expand_blanks if expand_blanks is None else expand_blanks
    if m.blank_score is not None and expand:
        T, N, C = s.shape
        s = F.pad(s.view(T, N, C // m.n_base, m.n_base), (1, 0), value=m.blank_score).view(T, N, -1)
    return s


def rotary(qkv):
    """RotaryEmbedding(head_dim, interleaved=False), base 10000, positions from 0, on q and k of a packed
    [N,T,3,h,d] tensor ([EXT] flash_attn.layers.rotary)."""
    N, T, _, h, d = qkv.shape
    half = d // 2
    inv = 1.0 / (10000.0 ** (torch.arange(0, d, 2, dtype=torch.float32) / d))
    ang = torch.arange(T, dtype=torch.float32)[:, None] * inv[None, :]
    cos, sin = ang.cos()[None, :, None, :], ang.sin()[None, :, None, :]
    out = qkv.clone()
    for i in (0, 1):
        x1, x2 = qkv[:, :, i, :, :half], qkv[:, :, i, :, half:]
        out[:, :, i, :, :half] = x1 * cos - x2 * sin
        out[:, :, i, :, half:] = x1 * sin + x2 * cos
    return out


def window_mask(T, window):
    i = torch.arange(T)[:, None]
    j = torch.arange(T)[None, :]
    return (j >= i - window[0]) & (j <= i + window[1])


def transformer_layer_forward(m, x, fp16=False):
    """x [N,T,D]; reference transformer/model.py:68-79,125-128 with the SDPA formulation (:62-65). fp16: rounded where the engine
    stores fp16 - qkv after the rotary epilogue (the engine scales q by log2(e)/sqrt(d) BEFORE that rounding, here the scale follows it:
    a relative 2^-11 either way, not emulated), the un-normalised probabilities exp(s - max) that feed the P.V MFMAs, the attention output, every GEMM / norm
    output."""
    r16 = _h if fp16 else (lambda t: t)
    att = m.self_attn
    N, T, D = x.shape
    h, d = att.nhead, att.head_dim
    qkv = (x @ att.Wqkv.weight.float().T).view(N, T, 3, h, d)
    if att.Wqkv.bias is not None:
        qkv = qkv + att.Wqkv.bias.float().view(3, h, d)
    qkv = r16(rotary(qkv))
    q, k, v = (qkv[:, :, i].permute(0, 2, 1, 3) for i in range(3))
    s = (q @ k.transpose(-1, -2)) / math.sqrt(d)
    win = tuple(att.attn_window)
    if win != (-1, -1):
        s = s.masked_fill(~window_mask(T, win), float("-inf"))
    if fp16:
        p = r16(torch.exp(s - s.amax(-1, keepdim=True)))
        o = r16((p @ v) / p.sum(-1, keepdim=True)).permute(0, 2, 1, 3).reshape(N, T, D)
    else:
        o = (torch.softmax(s, dim=-1) @ v).permute(0, 2, 1, 3).reshape(N, T, D)
    o = o @ att.out_proj.weight.float().T
    if att.out_proj.bias is not None:
        o = o + att.out_proj.bias.float()
    o = r16(o)
    alpha = float(m.deepnorm_alpha)

    def rms(z, w, eps=1e-5):
        return r16(z * torch.rsqrt(z.pow(2).mean(-1, keepdim=True) + eps) * w.float())

    x = rms(o + alpha * x, m.norm1.weight)
    hmid = x @ m.ff.fc1.weight.float().T
    y, gate = hmid.chunk(2, dim=-1)
    f = r16(r16(y * (gate * torch.sigmoid(gate))) @ m.ff.fc2.weight.float().T)
    return rms(f + alpha * x, m.norm2.weight)


def _bn_eval(bn, h):
    h = (h - bn.running_mean.float()[None, :, None]) * torch.rsqrt(bn.running_var.float()[None, :, None] + bn.eps)
    if bn.affine:
        h = h * bn.weight.float()[None, :, None] + bn.bias.float()[None, :, None]
    return h


def _conv1d(c, x):
    return F.conv1d(x, c.weight.float(), None if c.bias is None else c.bias.float(), stride=c.stride,
                    padding=c.padding, dilation=c.dilation, groups=c.groups)


def _tcs(m, x):
    """TCSConv1d (bonito/ctc/model.py:90-121): depthwise + pointwise, or a plain convolution."""
    if m.separable:
        return _conv1d(m.pointwise, _conv1d(m.depthwise, x))
    return _conv1d(m.conv, x)


def _quartznet_layer(layer, x):
    n = type(layer).__name__
    if n == "TCSConv1d":
        return _tcs(layer, x)
    if n == "BatchNorm1d":
        return _bn_eval(layer, x)
    if n == "Dropout":
        return x
    return _act(layer, x)      # Swish / SiLU / ReLU / Tanh marker


def ctc_forward(model, x):
    """bonito.ctc Model.forward (ctc/model.py:35-37): Encoder blocks (:177-192) then Decoder (:206-207).
    Returns log-probabilities [T, N, n_labels]."""
    for block in model.encoder.encoder:
        h = x
        for layer in block.conv:
            h = _quartznet_layer(layer, h)
        if block.use_res:
            r = x
            for layer in block.residual:
                r = _quartznet_layer(layer, r)
            h = h + r
        for layer in block.activation:
            h = _quartznet_layer(layer, h)
        x = h
    logits = _conv1d(model.decoder.layers[0], x).permute(2, 0, 1)
    return torch.log_softmax(logits, dim=-1)


def forward(m, x, expand_blanks=None, fp16=False):
    """Evaluate module tree `m` on fp32 CPU tensor x (reference layouts: NCL in, TNC scores out). `fp16`: see the module docstring."""
    n = _name(m)
    if n in ("serial", "namedserial", "stack", "sequential"):
        for child in m.children():
            x = forward(child, x, expand_blanks, fp16)
        return x
    if n == "convolution":
        return conv_forward(m, x, fp16)
    if n == "permute":
        return x.permute(*m.dims)
    if n == "makecontiguous":
        return x.contiguous()
    if n == "lstm":
        return lstm_forward(m, x, fp16)
    if n == "linearcrfencoder":
        return crf_encoder_forward(m, x, expand_blanks, fp16)
    if n == "clamp":
        return torch.clamp(x, m.min, m.max)
    if n == "linear":                     # bonito/nn.py:27-38
        y = x @ m.linear.weight.float().T
        y = y if m.linear.bias is None else y + m.linear.bias.float()
        return _h(y) if fp16 else y
    if n == "linearupsample":
        if not m.batch_first:
            x = x.permute(1, 0, 2)
        N, L, E = x.shape
        hh = (x @ m.linear.weight.float().T + m.linear.bias.float()).reshape(N, m.scale_factor * L, E)
        if fp16:
            hh = _h(hh)
        return hh if m.batch_first else hh.permute(1, 0, 2)
    if n == "transformerencoderlayer":
        return transformer_layer_forward(m, x, fp16)
    if n == "seqdistmodel" or hasattr(m, "encoder"):
        return forward(m.encoder, x, expand_blanks, fp16)
    raise NotImplementedError("oracle has no restatement of layer %r" % n)


def round_params_to_half_(m):
    """Round every floating parameter/buffer to fp16 precision in place (kept as fp32): the oracle then
    sees exactly the weights the fp16 engine sees."""
    with torch.no_grad():
        for p in list(m.parameters()) + list(m.buffers()):
            if p.dtype.is_floating_point:
                p.copy_(p.half().float())
    return m
