"""
TEST INFRASTRUCTURE - not imported by the product path.

Definition "Q8-1" of an 8-bit recurrent path (bonito's `--quantize`, cli/basecaller.py:186-189, hands the
LSTM stack to koi's int8 kernels; those are closed source, so there is nothing to pin parity against: **parity unpinned**).
This file states what an MI355X int8 kernel would have to compute bit for bit, so that the kernel can be checked against
it the way the fp16 kernels are checked against oracle/nn_ref.py, and it measures what the scheme costs in accuracy.

    W_ih, W_hh : int8 per output row, scale s_r = max|W[r, :]| / 127, q = clip(rint(W / s_r), -127, 127)
    x_t, h_t   : int8 with a static scale: h in (-1, 1) -> rint(127 h); x of the first recurrent layer -> rint(127 x / bound)
                 (bound = the clamp / tanh bound of the layer before, or 4.0 after an unbounded swish; larger values saturate)
    pre-activation[r] = s_ih[r] * (bound / 127) * sum_k q_ih[r, k] xq[k]  +  s_hh[r] / 127 * sum_k q_hh[r, k] hq[k]  +  b[r]
                 (the two int32 sums are exact: v_mfma_i32_16x16x64_i8), everything after it in fp32 as in the fp16 path:
    c' = sigmoid(f) c + sigmoid(i) tanh(g);  h' = sigmoid(o) tanh(c');  the layer publishes fp16(h') and quantises THAT value.
rint is round-half-to-even (v_rndne_f32).
"""
import numpy as np
import torch

from oracle import nn_ref

SWISH_BOUND = 4.0


def quantise_rows(W):
    """[R, K] fp32 -> (int8 [R, K], fp32 scale [R])."""
    W = np.asarray(W, np.float32)
    s = np.abs(W).max(axis=1) / np.float32(127.0)
    s = np.where(s > 0, s, np.float32(1.0)).astype(np.float32)
    q = np.clip(np.rint(W / s[:, None]), -127, 127).astype(np.int8)
    return q, s


def quantise_act(x, bound):
    return np.clip(np.rint(np.asarray(x, np.float32) * np.float32(127.0 / bound)), -127, 127).astype(np.int8)


def _sigmoid(v):
    return 1.0 / (1.0 + np.exp(-v))


def lstm_q8_forward(m, x, bound=1.0):
    """x [T, N, I] fp32 tensor -> [T, N, H] fp32 tensor (values on the fp16 grid, as the engine publishes them)."""
    r = m.rnn
    q_ih, s_ih = quantise_rows(r.weight_ih_l0.detach().float().numpy())
    q_hh, s_hh = quantise_rows(r.weight_hh_l0.detach().float().numpy())
    b = np.zeros(q_ih.shape[0], np.float32)
    if r.bias:
        b = (r.bias_ih_l0.detach().float() + r.bias_hh_l0.detach().float()).numpy()
    x = x.detach().float().numpy()
    T, N, _ = x.shape
    H = r.hidden_size
    xq = quantise_act(x, bound).astype(np.int32)
    gx = (xq.reshape(T * N, -1) @ q_ih.astype(np.int32).T).reshape(T, N, 4 * H).astype(np.float32) * (s_ih * np.float32(bound / 127.0)) + b
    w_hh = q_hh.astype(np.int32).T
    s_h = s_hh / np.float32(127.0)
    h16 = np.zeros((N, H), np.float32)
    c = np.zeros((N, H), np.float32)
    out = np.empty((T, N, H), np.float32)
    for t in (range(T - 1, -1, -1) if m.reverse else range(T)):
        hq = quantise_act(h16, 1.0).astype(np.int32)
        g = gx[t] + (hq @ w_hh).astype(np.float32) * s_h
        i, f, gg, o = np.split(g, 4, axis=-1)
        c = (_sigmoid(f) * c + _sigmoid(i) * np.tanh(gg)).astype(np.float32)
        h = (_sigmoid(o) * np.tanh(c)).astype(np.float32)
        h16 = h.astype(np.float16).astype(np.float32)
        out[t] = h16
    return torch.from_numpy(out)


def forward_q8(m, x, expand_blanks=None, _state=None):
    """oracle/nn_ref.forward with every LSTM layer replaced by its Q8-1 version; tracks the bound of the tensor that
    enters the first recurrent layer."""
    st = _state if _state is not None else {"bound": SWISH_BOUND}
    n = nn_ref._name(m)
    if n in ("serial", "namedserial", "stack", "sequential"):
        for child in m.children():
            x = forward_q8(child, x, expand_blanks, st)
        return x
    if n == "seqdistmodel" or (hasattr(m, "encoder") and n not in ("lstm",)):
        return forward_q8(m.encoder, x, expand_blanks, st)
    if n == "lstm":
        y = lstm_q8_forward(m, x, st["bound"])
        st["bound"] = 1.0
        return y
    y = nn_ref.forward(m, x, expand_blanks)
    if n == "convolution":
        act = nn_ref._name(m.activation) if getattr(m, "activation", None) is not None else ""
        st["bound"] = 1.0 if act == "tanh" else SWISH_BOUND
    elif n == "clamp":
        st["bound"] = float(max(abs(m.min), abs(m.max)))
    return y
