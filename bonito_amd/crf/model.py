"""
CTC-CRF model surface of the MI355X engine: mirrors ``bonito.crf.model`` (/root/reference
bonito/crf/model.py) -- ``get_stride``, ``CTC_CRF``, ``rnn_encoder``, ``SeqdistModel``, ``Model`` -- with
every koi / cuDNN call replaced by the HIP engine:

* ``Model.use_koi(batchsize, chunksize, quantize)`` (crf/model.py:240-246) == ``use_hip``: the encoder is
  lowered to ``bh_encoder_*`` (built lazily on first forward, i.e. after ``load_state_dict`` / ``half`` /
  ``to`` / ``fuse_bn_`` which the reference loader runs *after* use_koi, util.py:292-310);
* ``CTC_CRF.viterbi / logZ / posteriors`` (crf/model.py:47-67,98-103) call the HIP decode kernels.
"""
import numpy as np
import torch

from bonito_amd import decode as hip_decode
from bonito_amd.engine import HipEncoder
from bonito_amd.nn import (Module, Convolution, LinearCRFEncoder, Serial, Permute, layers, to_dict, from_dict,
                           register, NoTorchCompute)


def get_stride(m, stride=1):
    """Total down-sampling of a module tree (reference crf/model.py:15-27)."""
    if hasattr(m, "output_stride"):
        return m.output_stride(stride)
    if hasattr(m, "stride"):
        s = m.stride
        if isinstance(s, tuple):
            assert len(s) == 1
            s = s[0]
        return stride * s
    for child in m.children():
        stride = get_stride(child, stride)
    return stride


class CTC_CRF:
    """k-mer CTC-CRF sequence distribution (reference crf/model.py:30-108).

    State j is a base-4 k-mer (oldest base most significant). ``idx[j, 0] = j`` (stay) and
    ``idx[j, 1+r] = r*S/4 + j//4`` (move into j having dropped base r)."""

    def __init__(self, state_len, alphabet):
        self.alphabet = alphabet
        self.state_len = state_len
        self.n_base = len(alphabet[1:])
        S = self.n_base ** self.state_len
        states = torch.arange(S)
        self.idx = torch.cat([
            states[:, None],
            states.repeat_interleave(self.n_base).reshape(self.n_base, -1).T,
        ], dim=1).to(torch.int32)

    def n_score(self):
        return len(self.alphabet) * self.n_base ** self.state_len

    def viterbi(self, scores):
        """Best path on expand_blanks-layout scores [T, N, 5*S] (cuda fp16/fp32) -> [T, N] in {0..4}
        (0 = no emission), as reference crf/model.py:98-103."""
        moves, path = hip_decode.viterbi_5s(scores.to(torch.float16), self.state_len)
        return path.T.to(torch.int64)

    def path_to_str(self, path):
        alphabet = np.frombuffer("".join(self.alphabet).encode(), dtype="u1")
        seq = alphabet[path[path != 0]]
        return seq.tobytes().decode()

    def reverse_complement(self, scores):
        """Permute scores so that decoding yields the reverse-complement strand (reference crf/model.py:84-96).
        Accepts the engine's koi layout [N, T, 4S] or the reference layout [T, N, 5S] (cuda fp16)."""
        S = self.n_base ** self.state_len
        if scores.shape[-1] == 4 * S:
            return hip_decode.reverse_complement(scores.contiguous())
        return hip_decode.reverse_complement_5s(scores, self.state_len)

    def logZ(self, scores, blank_score=2.0):
        """Log partition function per chunk of koi-layout scores [N, T, 4S] (reference crf/model.py:47-52)."""
        return hip_decode.logz(scores.contiguous(), blank_score)


def conv(c_in, c_out, ks, stride=1, bias=False, activation=None, norm=None):
    return Convolution(c_in, c_out, ks, stride=stride, padding=ks // 2, bias=bias, activation=activation, norm=norm)


def rnn_encoder(n_base, state_len, insize=1, first_conv_size=4, stride=5, winlen=19, activation="swish",
                rnn_type="lstm", features=768, scale=5.0, blank_score=None, expand_blanks=True, num_layers=5,
                norm=None):
    """Old-style ([encoder] without `type`) conv x3 -> alternating LSTMs -> tanh*scale CRF head
    (reference crf/model.py:150-162)."""
    rnn = layers[rnn_type]
    return Serial([
        conv(insize, first_conv_size, ks=5, bias=True, activation=activation, norm=norm),
        conv(first_conv_size, 16, ks=5, bias=True, activation=activation, norm=norm),
        conv(16, features, ks=winlen, stride=stride, bias=True, activation=activation, norm=norm),
        Permute([2, 0, 1]),
        *(rnn(features, features, reverse=(num_layers - i) % 2) for i in range(num_layers)),
        LinearCRFEncoder(features, n_base, state_len, activation="tanh", scale=scale, blank_score=blank_score,
                         expand_blanks=expand_blanks),
    ])


@register
class SeqdistModel(Module):
    def __init__(self, encoder, seqdist, n_pre_post_context_bases=None, target_projection=None):
        super().__init__()
        self.seqdist = seqdist
        self.encoder = encoder
        self.stride = get_stride(encoder)
        self.alphabet = seqdist.alphabet
        if n_pre_post_context_bases is None:
            self.n_pre_context_bases = self.seqdist.state_len - 1
            self.n_post_context_bases = 1
        else:
            self.n_pre_context_bases, self.n_post_context_bases = n_pre_post_context_bases
        if target_projection is None:
            self.target_projection = None
        else:
            self.register_buffer("target_projection", torch.tensor([0] + target_projection), persistent=False)
        self._hip = None          # HipEncoder (plain object, not a submodule), built lazily
        self._hip_args = None     # (batchsize, chunksize) requested through use_koi / use_hip

    @classmethod
    def from_dict(cls, model_dict, layer_types=None):
        kwargs = dict(model_dict, encoder=from_dict(model_dict["encoder"], layer_types),
                      seqdist=CTC_CRF(**model_dict["seqdist"]))
        return cls(**kwargs)

    # ---- accelerator swap point ---------------------------------------------------------------
    def use_hip(self, batchsize=None, chunksize=None, quantize=None, **_):
        """Route ``forward`` through the HIP engine. Same keywords as the reference's ``use_koi``.
        ``quantize=True`` (cli/basecaller.py:186-189, crf/model.py:245) selects the 8-bit recurrent path Q8-1 for every LSTM
        layer the int8 kernel covers (hidden sizes that are multiples of 48 or 64 up to 512, input size == hidden size);
        other layers keep the fp16 kernels -- ``model._hip.describe()`` lists what runs where. The default stays fp16."""
        self._hip_args = (batchsize, chunksize)
        self._quantize = bool(quantize)
        self._drop_engine()
        return self

    use_koi = use_hip

    # The engine snapshots the weights when it is built (first forward). Anything that changes parameters afterwards --
    # load_state_dict, .half() / .to() / .float() (Module._apply), apply(fuse_bn_) -- drops it, so the next forward lowers
    # the current weights again instead of silently running stale ones.
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

    def _engine(self, x):
        N, L = x.shape[0], x.shape[-1]
        if self._hip is not None and (N > self._hip.max_batch or L > self._hip.max_chunk
                                      or x.device != self._hip.device):
            self._hip.close()
            self._hip = None
        if self._hip is None:
            bs, cs = self._hip_args if self._hip_args is not None else (None, None)
            self._hip = HipEncoder(self.encoder, max(int(bs or 0), N), max(int(cs or 0), L), device=x.device,
                                   quantize=getattr(self, "_quantize", False))
        return self._hip

    def engine_replica(self, x):
        """A second, independent engine over the same weights (own workspace): the basecaller keeps `lanes` batches in flight,
        one engine each (crf/basecall.py). Built for the geometry of the model's own engine."""
        eng = self._engine(x)
        return HipEncoder(self.encoder, eng.max_batch, eng.max_chunk, device=eng.device, quantize=getattr(self, "_quantize", False))

    def forward(self, x, *args):
        """x: cuda fp16 [N,1,L] -> scores fp16 [N, T, 4^(state_len+1)] (koi layout, contiguous)."""
        if not x.is_cuda:
            raise NoTorchCompute("bonito_amd models only run on a HIP device; got a %s tensor" % x.device)
        return self._engine(x)(x)

    def decode_batch(self, x, blank_score=2.0):
        """Posterior decoding of engine scores [N, T, 4S] (cuda fp16) -> list of strings
        (reference crf/model.py:196-199: viterbi over log(posteriors + 1e-8))."""
        _, paths = hip_decode.posterior_viterbi(x.contiguous(), blank_score)
        return [self.seqdist.path_to_str(p) for p in paths.numpy()]

    def decode(self, x):
        return self.decode_batch(x.unsqueeze(0))[0]

    def to_dict(self, include_weights=False):
        if include_weights:
            raise NotImplementedError
        res = {
            "encoder": to_dict(self.encoder),
            "seqdist": {"state_len": self.seqdist.state_len, "alphabet": self.seqdist.alphabet},
            "n_pre_post_context_bases": (self.n_pre_context_bases, self.n_post_context_bases),
        }
        if self.target_projection is not None:
            res["target_projection"] = self.target_projection.tolist()[1:]
        return res


class Model(SeqdistModel):
    """``config.toml`` -> model (reference crf/model.py:225-238)."""

    def __init__(self, config):
        seqdist = CTC_CRF(state_len=config["global_norm"]["state_len"], alphabet=config["labels"]["labels"])
        if "type" in config["encoder"]:   # new-style config
            encoder = from_dict(config["encoder"])
        else:                             # old-style
            encoder = rnn_encoder(seqdist.n_base, seqdist.state_len, insize=config["input"]["features"],
                                  **config["encoder"])
        super().__init__(encoder, seqdist,
                         n_pre_post_context_bases=config["input"].get("n_pre_post_context_bases"))
        self.config = config
