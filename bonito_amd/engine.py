"""
Lowering of a ``bonito_amd.nn`` module tree to the C-ABI encoder engine (``bh_encoder_*`` in
include/bonito_hip.h) and a small Python handle around it.

This is the MI355X counterpart of ``koi.lstm.update_graph(encoder, batchsize, chunksize, quantize)``
(/root/reference bonito/crf/model.py:240-246) and of the transformer ``use_koi`` rewrite
(bonito/transformer/model.py:136-146): the returned object is callable like the swapped-in encoder and
yields **NTC fp16 scores with expand_blanks=False** ([N, T, n_base^(state_len+1)], contiguous) -- the
layout koi.decode.beam_search consumes (bonito/crf/basecall.py:36-40).

torch is used for device memory and the current stream only.
"""
import ctypes as C

import torch

from bonito_amd import _lib
from bonito_amd import nn as bnn


class LoweringError(RuntimeError):
    pass


def _f32(t):
    return t.detach().to(device="cpu", dtype=torch.float32).contiguous()


class _Lowered:
    """bh_layer_t array + the host tensors it points into (kept alive until the engine is created)."""

    def __init__(self):
        self.descs = []
        self.keep = []

    def add(self, **fields):
        d = _lib.bh_layer_t()
        for k, v in fields.items():
            if isinstance(v, torch.Tensor):
                self.keep.append(v)
                v = v.data_ptr()
            setattr(d, k, v)
        self.descs.append(d)
        return d

    def array(self):
        arr = (_lib.bh_layer_t * len(self.descs))()
        for i, d in enumerate(self.descs):
            arr[i] = d
        return arr


def _act_id(module):
    if module is None:
        return 0
    name = getattr(module, "name", type(module).__name__.lower())
    if name == "silu":
        name = "swish"
    if name not in _lib.BH_ACT:
        raise LoweringError("activation %r is not supported by the HIP engine" % name)
    return _lib.BH_ACT[name]


def _flatten(module):
    """Depth-first list of leaf layers of Serial / NamedSerial / Stack containers."""
    if isinstance(module, (bnn.Serial, bnn.NamedSerial, torch.nn.Sequential)):
        out = []
        for child in module.children():
            out.extend(_flatten(child))
        return out
    return [module]


def lower(encoder, quantize=False):
    """Module tree -> _Lowered chain. Raises LoweringError on anything the engine cannot run.
    `quantize`: mark every LSTM layer for the 8-bit recurrent path Q8-1 (koi's `quantize`, reference crf/model.py:245); the
    engine uses it where its int8 kernel covers the layer shape and keeps the fp16 kernels elsewhere (HipEncoder.describe())."""
    low = _Lowered()
    time_major = False   # reference activations are NCL until Permute([2,0,1]) makes them TNC
    for m in _flatten(encoder):
        if isinstance(m, bnn.Convolution):
            w, b = m.folded()
            c = m.conv
            if c.groups != 1 or c.dilation[0] != 1:
                raise LoweringError("grouped/dilated convolutions are lowered by the ctc path only")
            low.add(kind=_lib.BH_LAYER_CONV, in_size=c.in_channels, out_size=c.out_channels,
                    winlen=c.kernel_size[0], stride=c.stride[0], padding=c.padding[0],
                    activation=_act_id(m.activation), groups=1, w0=w, b0=b)
        elif isinstance(m, bnn.Permute):
            dims = list(m.dims)
            if dims == [2, 0, 1]:      # NCL -> TNC: folded into the producing convolution's store
                time_major = True
            elif dims == [0, 2, 1]:    # NCL -> NLC: our activations are channel-minor already
                pass
            elif dims == [1, 0, 2]:    # TNC <-> NTC after the final linear: the engine always emits NTC
                pass
            else:
                raise LoweringError("permute %s cannot be folded by the HIP engine" % (dims,))
        elif isinstance(m, bnn.MakeContiguous):
            pass
        elif isinstance(m, bnn.LSTM):
            r = m.rnn
            if r.num_layers != 1 or r.bidirectional:
                raise LoweringError("only single-layer unidirectional LSTMs are supported")
            if not time_major:
                raise LoweringError("lstm must follow permute [2,0,1]")
            low.add(kind=_lib.BH_LAYER_LSTM, in_size=r.input_size, out_size=r.hidden_size,
                    reverse=int(bool(m.reverse)), quantize=int(bool(quantize)), w0=_f32(r.weight_ih_l0), w1=_f32(r.weight_hh_l0),
                    b0=_f32(r.bias_ih_l0) if r.bias else 0, b1=_f32(r.bias_hh_l0) if r.bias else 0)
        elif isinstance(m, bnn.LinearCRFEncoder):
            lin = m.linear
            low.add(kind=_lib.BH_LAYER_LINEAR_CRF, in_size=lin.in_features, out_size=lin.out_features,
                    activation=_act_id(m.activation), scale=float(m.scale) if m.scale is not None else 0.0,
                    blank_score=float(m.blank_score) if m.blank_score is not None else 0.0,
                    w0=_f32(lin.weight), b0=_f32(lin.bias) if lin.bias is not None else 0)
        elif type(m).__name__ == "TransformerEncoderLayer":
            att, kw = m.self_attn, m.kwargs
            win = tuple(att.attn_window)
            if win[0] < 0 or win[1] < 0:
                raise LoweringError("the HIP attention kernel needs a finite attn_window")
            low.add(kind=_lib.BH_LAYER_TRANSFORMER, in_size=kw["d_model"], nhead=kw["nhead"], dim_ff=kw["dim_feedforward"],
                    win_left=int(win[0]), win_right=int(win[1]), alpha=float(m.deepnorm_alpha), eps=float(m.norm1.eps),
                    w0=_f32(att.Wqkv.weight), b0=_f32(att.Wqkv.bias) if att.Wqkv.bias is not None else 0,
                    w1=_f32(att.out_proj.weight), b1=_f32(att.out_proj.bias) if att.out_proj.bias is not None else 0,
                    w2=_f32(m.ff.fc1.weight), w3=_f32(m.ff.fc2.weight), w4=_f32(m.norm1.weight), w5=_f32(m.norm2.weight))
        elif isinstance(m, bnn.LinearUpsample):
            if not m.batch_first:
                raise LoweringError("linearupsample with batch_first=False is not lowered")
            low.add(kind=_lib.BH_LAYER_UPSAMPLE, in_size=m.d_model, scale_factor=m.scale_factor,
                    w0=_f32(m.linear.weight), b0=_f32(m.linear.bias))
        elif isinstance(m, bnn.Linear):
            lin = m.linear
            low.add(kind=_lib.BH_LAYER_LINEAR, in_size=lin.in_features, out_size=lin.out_features,
                    w0=_f32(lin.weight), b0=_f32(lin.bias) if lin.bias is not None else 0)
        elif isinstance(m, bnn.Clamp):
            low.add(kind=_lib.BH_LAYER_CLAMP, clamp_lo=float(m.min), clamp_hi=float(m.max))
        else:
            raise LoweringError("layer %s has no HIP lowering" % type(m).__name__)
    return low


def _fold(conv, bn):
    """fp32 (weight, bias) of `conv` followed by eval-mode BatchNorm `bn`."""
    w = _f32(conv.weight)
    b = _f32(conv.bias) if conv.bias is not None else torch.zeros(w.shape[0])
    s = _f32(bn.weight) * torch.rsqrt(_f32(bn.running_var) + bn.eps)
    return (w * s[:, None, None]).contiguous(), ((b - _f32(bn.running_mean)) * s + _f32(bn.bias)).contiguous()


def lower_ctc(model):
    """bonito_amd.ctc.Model (QuartzNet) -> primitive engine layers:
    Block = [residual projection] + repeat x ([depthwise conv] + pointwise/plain conv with folded BatchNorm,
    activation fused; the last one adds the residual before the activation) ; Decoder = 1x1 conv + log_softmax."""
    low = _Lowered()
    for block in model.encoder.encoder:
        mods = list(block.conv)
        pairs = [(mods[i], mods[i + 1]) for i in range(0, len(mods), 4)]      # (TCSConv1d, BatchNorm1d) every 4
        act = _act_id(block.activation[0])
        first = pairs[0][0]
        cin = (first.depthwise if first.separable else first.conv).in_channels
        if block.use_res:
            rconv, rbn = block.residual[0].conv, block.residual[1]
            w, b = _fold(rconv, rbn)
            low.add(kind=_lib.BH_LAYER_RESIDUAL_PROJ, in_size=cin, out_size=rconv.out_channels,
                    w0=w.reshape(w.shape[0], -1).contiguous(), b0=b)
        for r, (tcs, bn) in enumerate(pairs):
            last = r == len(pairs) - 1
            if tcs.separable:
                dw, pw = tcs.depthwise, tcs.pointwise
                if dw.dilation[0] != 1:
                    raise LoweringError("dilated depthwise convolutions are not lowered")
                low.add(kind=_lib.BH_LAYER_DWCONV, in_size=dw.in_channels, out_size=dw.in_channels,
                        winlen=dw.kernel_size[0], stride=dw.stride[0], padding=dw.padding[0],
                        w0=_f32(dw.weight).reshape(dw.in_channels, -1).contiguous())
                w, b = _fold(pw, bn)
                low.add(kind=_lib.BH_LAYER_CONV, in_size=pw.in_channels, out_size=pw.out_channels, winlen=1, stride=1,
                        padding=0, activation=act, groups=1, add_residual=int(last and block.use_res), w0=w, b0=b)
            else:
                c = tcs.conv
                if c.dilation[0] != 1:
                    raise LoweringError("dilated convolutions are not lowered")
                w, b = _fold(c, bn)
                low.add(kind=_lib.BH_LAYER_CONV, in_size=c.in_channels, out_size=c.out_channels, winlen=c.kernel_size[0],
                        stride=c.stride[0], padding=c.padding[0], activation=act, groups=1,
                        add_residual=int(last and block.use_res), w0=w, b0=b)
    head = model.decoder.layers[0]
    low.add(kind=_lib.BH_LAYER_CTC_DECODER, in_size=head.in_channels, out_size=head.out_channels,
            w0=_f32(head.weight).reshape(head.out_channels, -1).contiguous(), b0=_f32(head.bias))
    return low


class HipEncoder:
    """Callable engine handle: ``scores = enc(signal)`` with signal fp16 cuda [N,1,L] or [N,L]."""

    def __init__(self, encoder, batchsize, chunksize, device=None, lowering=None, quantize=False):
        self._handle = None
        self.last_ticket = None
        lib = _lib.lib()
        if not torch.cuda.is_available():
            raise _lib.HipEngineError("no HIP device visible: the MI355X engine has no CPU fallback")
        dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        if dev.type != "cuda":
            raise _lib.HipEngineError("the encoder engine needs a GPU device, got %s" % (dev,))
        self.device = dev
        self.max_batch, self.max_chunk = int(batchsize), int(chunksize)
        low = lowering(encoder) if lowering is not None else lower(encoder, quantize=quantize)
        self.quantize = bool(quantize)
        handle = C.c_void_p()
        _lib.check(lib.bh_encoder_create(low.array(), len(low.descs), dev.index or 0, self.max_batch,
                                         self.max_chunk, C.byref(handle)), "bh_encoder_create")
        self._handle = handle

    def output_shape(self, L):
        T, Cc, s = C.c_int(), C.c_int(), C.c_int()
        _lib.check(_lib.lib().bh_encoder_output_shape(self._handle, int(L), C.byref(T), C.byref(Cc), C.byref(s)),
                   "bh_encoder_output_shape")
        return T.value, Cc.value, s.value

    def __call__(self, x):
        if x.dim() == 3:
            if x.shape[1] != 1:
                raise ValueError("expected [N,1,L] signal, got %s" % (tuple(x.shape),))
            x = x[:, 0]
        if x.dtype != torch.float16:
            raise TypeError("signal must be float16 (as in bonito/crf/basecall.py:33), got %s" % x.dtype)
        if x.device != self.device:
            raise ValueError("signal is on %s but the engine lives on %s" % (x.device, self.device))
        x = x.contiguous()
        N, L = x.shape
        T, Cc, _ = self.output_shape(L)
        scores = torch.empty((N, T, Cc), dtype=torch.float16, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().bh_encoder_forward(self._handle, _lib.ptr(x), N, L, _lib.ptr(scores),
                                                     _lib.stream_ptr(self.device)), "bh_encoder_forward")
        # the number of this forward: its timeout flag is kept apart from those of the forwards around it (poll_ticket)
        self.last_ticket = int(_lib.lib().bh_encoder_last_ticket(self._handle))
        return scores

    # "attention" = the projections + norm around the attention kernel, "mlp" = fc2 + norm; "mlp_fc1" and "attention_core" hold ONE kernel each
    PROF_CLASSES = ("conv", "lstm_gemm", "fill", "lstm_rec", "crf_linear", "attention", "mlp", "other", "mlp_fc1", "attention_core")

    def profile(self, enable=True):
        _lib.check(_lib.lib().bh_encoder_profile(self._handle, int(bool(enable))), "bh_encoder_profile")

    def profile_read(self):
        """-> {class: (milliseconds, spans)} accumulated since the last read (HIP events on the forward stream)."""
        ms = (C.c_float * len(self.PROF_CLASSES))()
        n = (C.c_int * len(self.PROF_CLASSES))()
        _lib.check(_lib.lib().bh_encoder_profile_read(self._handle, ms, n), "bh_encoder_profile_read")
        return {k: (float(ms[i]), int(n[i])) for i, k in enumerate(self.PROF_CLASSES)}

    def set_option(self, name, value):
        _lib.check(_lib.lib().bh_encoder_set_option(self._handle, name.encode(), int(value)), "bh_encoder_set_option")

    def check(self):
        """Synchronise and raise if a persistent kernel hit its spin bound."""
        with torch.cuda.device(self.device):
            rc = _lib.lib().bh_encoder_check(self._handle, _lib.stream_ptr(self.device))
        if rc:
            raise _lib.HipEngineError("bh_encoder_check: %s" % _lib.last_error())

    def clear_error(self):
        """Synchronise and reset the (sticky) timeout flag without raising; returns whether it was set."""
        with torch.cuda.device(self.device):
            return bool(_lib.lib().bh_encoder_check(self._handle, _lib.stream_ptr(self.device)))

    def describe(self):
        """One line per layer: which kernels the engine launches for it."""
        buf = C.create_string_buffer(1 << 14)
        _lib.check(_lib.lib().bh_encoder_describe(self._handle, buf, len(buf)), "bh_encoder_describe")
        return buf.value.decode()

    def poll(self):
        """Raise if a forward whose completion the caller has already observed (event / decoded outputs / synchronise)
        hit the spin bound of a persistent kernel: its scores are invalid. No device round trip (bh_encoder_error_flag)."""
        if self._handle is not None and _lib.lib().bh_encoder_error_flag(self._handle):
            raise _lib.HipEngineError("bh_encoder_error_flag: %s" % _lib.last_error())

    def poll_ticket(self, ticket):
        """`poll` for ONE forward (`last_ticket` read right after the call that issued it): raises iff that forward timed out.
        Flags of other forwards in flight on this engine are left alone (bh_encoder_error_flag_at)."""
        if self._handle is None or ticket is None:
            return
        rc = _lib.lib().bh_encoder_error_flag_at(self._handle, int(ticket))
        if rc < 0:          # the slot has been recycled: more than 64 forwards were issued before this one was queried - a caller bug,
            raise RuntimeError("bh_encoder_error_flag_at: %s" % _lib.last_error())      # not a timeout (nothing to retry)
        if rc:
            raise _lib.HipEngineError("bh_encoder_error_flag_at: %s" % _lib.last_error())

    def ack(self, ticket):
        """The timeout of forward `ticket` has been handled (the batch was re-run): drop its flag, so that `poll()` / `check()` of this
        engine do not report a batch that was already repaired (bh_encoder_ack)."""
        if self._handle is not None and ticket is not None:
            _lib.check(_lib.lib().bh_encoder_ack(self._handle, int(ticket)), "bh_encoder_ack")

    def close(self):
        if self._handle is not None:
            _lib.lib().bh_encoder_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
