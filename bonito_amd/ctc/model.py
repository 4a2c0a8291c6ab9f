"""
QuartzNet CTC model surface (legacy dna_r9.4.1 `bonito.ctc`) on the MI355X engine: mirrors
/root/reference bonito/ctc/model.py -- ``Model`` (14-57), ``Encoder`` (59-87), ``TCSConv1d`` (90-121),
``Block`` (124-192), ``Decoder`` (195-207) -- as parameter containers with identical module nesting, hence
identical ``state_dict()`` keys (``encoder.encoder.<i>.conv.<j>.{depthwise,pointwise,conv}.weight``,
``...residual.0.conv.weight``, ``decoder.layers.0.{weight,bias}``) and identical seeded initialisation.

``model(x)`` runs the HIP engine (depthwise conv kernel, MFMA pointwise GEMMs with folded BatchNorm and the
residual add fused, 1x1 decoder + log_softmax) and returns log-probabilities in the reference's TNC layout.
The CPU Rust decoders of fast_ctc_decode (:11,39-46) are replaced by bonito_amd.ctc.decode (HIP).
"""
import torch
from torch.nn import BatchNorm1d, Conv1d, Dropout, Module, ModuleList, Sequential

from bonito_amd.engine import HipEncoder, lower_ctc
from bonito_amd.nn import NoTorchCompute, Permute, _no_forward, layers


class TCSConv1d(Module):
    """Time-channel separable 1-D convolution: depthwise (groups=C) + pointwise, or a plain convolution."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias=False,
                 separable=False):
        super().__init__()
        self.separable = separable
        if separable:
            self.depthwise = Conv1d(in_channels, in_channels, kernel_size=kernel_size, stride=stride, padding=padding,
                                    dilation=dilation, bias=bias, groups=in_channels)
            self.pointwise = Conv1d(in_channels, out_channels, kernel_size=1, stride=1, dilation=dilation, bias=bias,
                                    padding=0)
        else:
            self.conv = Conv1d(in_channels, out_channels, kernel_size=kernel_size, stride=stride, padding=padding,
                               dilation=dilation, bias=bias)

    forward = _no_forward


class Block(Module):
    """`repeat` x (TCSConv1d, BatchNorm(eps=1e-3)) with activation/dropout in between, optional pointwise
    residual branch added before the final activation."""

    def __init__(self, in_channels, out_channels, activation, repeat=5, kernel_size=1, stride=1, dilation=1,
                 dropout=0.0, residual=False, separable=False):
        super().__init__()
        self.use_res = residual
        self.conv = ModuleList()
        if stride[0] > 1 and dilation[0] > 1:
            raise ValueError("Dilation and stride can not both be greater than 1")
        padding = (kernel_size[0] // 2) * dilation[0]
        cin = in_channels
        for _ in range(repeat - 1):
            self.conv.extend(self._tcs(cin, out_channels, kernel_size, stride, dilation, padding, separable))
            self.conv.extend((activation, Dropout(p=dropout)))
            cin = out_channels
        self.conv.extend(self._tcs(cin, out_channels, kernel_size, stride, dilation, padding, separable))
        if self.use_res:
            self.residual = Sequential(*self._tcs(in_channels, out_channels))
        self.activation = Sequential(activation, Dropout(p=dropout))

    @staticmethod
    def _tcs(cin, cout, kernel_size=1, stride=1, dilation=1, padding=0, separable=False):
        return [TCSConv1d(cin, cout, kernel_size, stride=stride, dilation=dilation, padding=padding, bias=False,
                          separable=separable),
                BatchNorm1d(cout, eps=1e-3, momentum=0.1)]

    forward = _no_forward


class Encoder(Module):
    def __init__(self, config):
        super().__init__()
        self.config = config
        features = config["input"]["features"]
        activation = layers[config["encoder"]["activation"]]()
        blocks = []
        for layer in config["block"]:
            blocks.append(Block(features, layer["filters"], activation, repeat=layer["repeat"],
                                kernel_size=layer["kernel"], stride=layer["stride"], dilation=layer["dilation"],
                                dropout=layer["dropout"], residual=layer["residual"], separable=layer["separable"]))
            features = layer["filters"]
        self.encoder = Sequential(*blocks)

    forward = _no_forward


class Decoder(Module):
    def __init__(self, features, classes):
        super().__init__()
        self.layers = Sequential(Conv1d(features, classes, kernel_size=1, bias=True), Permute([2, 0, 1]))

    forward = _no_forward


class Model(Module):
    """QuartzNet-style CTC model (https://arxiv.org/abs/1910.10261) from a ``[[block]]`` config."""

    def __init__(self, config):
        super().__init__()
        qs = config.get("qscore")
        self.qbias = qs["bias"] if qs else 0.0
        self.qscale = qs["scale"] if qs else 1.0
        self.config = config
        self.stride = config["block"][0]["stride"][0]
        self.alphabet = config["labels"]["labels"]
        self.features = config["block"][-1]["filters"]
        self.encoder = Encoder(config)
        self.decoder = Decoder(self.features, len(self.alphabet))
        self._hip = None
        self._hip_args = None

    def use_hip(self, batchsize=None, chunksize=None, quantize=None, **_):
        self._hip_args = (batchsize, chunksize)
        self._hip = None
        return self

    use_koi = use_hip     # the reference CLI calls use_koi unconditionally (cli/basecaller.py:59, util.py:292-296)

    # the engine snapshots the weights at its first forward: parameter changes afterwards drop it (see crf/model.py)
    def _drop_engine(self):
        eng = self.__dict__.get("_hip")
        if eng is not None:
            eng.close()
        self.__dict__["_hip"] = None

    def load_state_dict(self, *args, **kwargs):
        self._drop_engine()
        return super().load_state_dict(*args, **kwargs)

    def _apply(self, fn, *args, **kwargs):
        self._drop_engine()
        return super()._apply(fn, *args, **kwargs)

    def forward(self, x):
        """x: cuda fp16 [N,1,L] -> log-probabilities fp16 [T, N, n_labels] (reference layout, a view)."""
        if not x.is_cuda:
            raise NoTorchCompute("bonito_amd models only run on a HIP device; got a %s tensor" % x.device)
        N, L = x.shape[0], x.shape[-1]
        if self._hip is not None and (N > self._hip.max_batch or L > self._hip.max_chunk or x.device != self._hip.device):
            self._hip.close()
            self._hip = None
        if self._hip is None:
            bs, cs = self._hip_args if self._hip_args is not None else (None, None)
            self._hip = HipEncoder(self, max(int(bs or 0), N), max(int(cs or 0), L), device=x.device, lowering=lower_ctc)
        return self._hip(x).permute(1, 0, 2)

    def decode(self, x, beamsize=5, threshold=1e-3, qscores=False, return_path=False):
        """x: log-probabilities [T, n_labels] of ONE read (any device). Greedy when beamsize == 1 or
        qscores (returns seq+qstring concatenated, like fast_ctc_decode.viterbi_search), else prefix beam."""
        from bonito_amd.ctc import decode as ctc_decode
        if beamsize == 1 or qscores:
            seq, path = ctc_decode.viterbi_search(x, self.alphabet, qscores, self.qscale, self.qbias)
        else:
            seq, path = ctc_decode.beam_search(x, self.alphabet, beamsize, threshold)
        return (seq, path) if return_path else seq
