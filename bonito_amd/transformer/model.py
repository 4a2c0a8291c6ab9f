"""
Transformer (v5 `sup`) model surface on the MI355X engine: mirrors ``bonito.transformer.model``
(/root/reference bonito/transformer/model.py) -- ``deepnorm_params``, ``MultiHeadAttention``,
``TransformerEncoderLayer`` (registered as ``transformerencoderlayer``), ``use_koi`` and ``Model`` -- as
parameter containers with the reference's parameter names and shapes:

    self_attn.Wqkv.weight [3D, D] (no bias)   self_attn.out_proj.{weight [D, D], bias [D]}
    ff.fc1.weight [2F, D]  ff.fc2.weight [D, F]  (no biases)   norm1.weight [D]  norm2.weight [D]
    deepnorm_alpha  (persistent buffer, :113)

flash-attn's modules (RotaryEmbedding, GatedMlp, RMSNorm, flash_attn_qkvpacked_func) are replaced by the
HIP kernels in bonito_amd/csrc/attention.hip and the gated GEMM epilogue in gemm.hip; nothing here computes.
"""
import types

import torch

from bonito_amd.crf.model import SeqdistModel  # noqa: F401  (registers `seqdistmodel`)
from bonito_amd.nn import from_dict, register, LinearCRFEncoder, Module, _no_forward


def deepnorm_params(depth):
    """DeepNorm (https://arxiv.org/abs/2203.00555) alpha and beta for an encoder of `depth` layers."""
    return round((2 * depth) ** 0.25, 7), round((8 * depth) ** (-1 / 4), 7)


class _GatedMlp(Module):
    """Parameter container with flash_attn.modules.mlp.GatedMlp's names: fc1 [2F, D], fc2 [D, F]."""

    def __init__(self, d_model, hidden_features):
        super().__init__()
        self.fc1 = torch.nn.Linear(d_model, 2 * hidden_features, bias=False)
        self.fc2 = torch.nn.Linear(hidden_features, d_model, bias=False)

    forward = _no_forward


class _RMSNorm(Module):
    def __init__(self, d_model, eps=1e-5):
        super().__init__()
        self.eps = eps
        self.weight = torch.nn.Parameter(torch.ones(d_model))

    forward = _no_forward


class MultiHeadAttention(Module):
    def __init__(self, d_model, nhead, qkv_bias=False, out_bias=True, rotary_dim=None, attn_window=None):
        super().__init__()
        assert d_model % nhead == 0, "d_model must be divisible by nhead"
        self.d_model, self.nhead = d_model, nhead
        self.head_dim = d_model // nhead
        self.rotary_dim = self.head_dim if rotary_dim is None else rotary_dim
        self.Wqkv = torch.nn.Linear(d_model, 3 * d_model, bias=qkv_bias)
        self.out_proj = torch.nn.Linear(d_model, d_model, bias=out_bias)
        self.attn_window = (-1, -1) if attn_window is None else tuple(attn_window)

    forward = _no_forward


@register
class TransformerEncoderLayer(Module):
    def __init__(self, d_model, nhead, dim_feedforward, deepnorm_alpha, deepnorm_beta, attn_window=None):
        super().__init__()
        self.kwargs = {"d_model": d_model, "nhead": nhead, "dim_feedforward": dim_feedforward,
                       "deepnorm_alpha": deepnorm_alpha, "deepnorm_beta": deepnorm_beta, "attn_window": attn_window}
        self.self_attn = MultiHeadAttention(d_model=d_model, nhead=nhead, qkv_bias=False, out_bias=True,
                                            attn_window=attn_window)
        self.ff = _GatedMlp(d_model, hidden_features=dim_feedforward)
        self.norm1 = _RMSNorm(d_model)
        self.norm2 = _RMSNorm(d_model)
        self.register_buffer("deepnorm_alpha", torch.tensor(deepnorm_alpha))
        self.reset_parameters()

    def reset_parameters(self):
        """DeepNorm init: gain beta on ff / out_proj / the V rows of Wqkv, gain 1 on the Q,K rows (:116-123)."""
        db, d = self.kwargs["deepnorm_beta"], self.kwargs["d_model"]
        xavier = torch.nn.init.xavier_normal_
        xavier(self.ff.fc1.weight, gain=db)
        xavier(self.ff.fc2.weight, gain=db)
        xavier(self.self_attn.out_proj.weight, gain=db)
        xavier(self.self_attn.Wqkv.weight[2 * d:], gain=db)
        xavier(self.self_attn.Wqkv.weight[:2 * d], gain=1)

    forward = _no_forward

    def to_dict(self, include_weights=False):
        if include_weights:
            raise NotImplementedError
        return self.kwargs


def use_koi(self, **kwargs):
    """Reference semantics (:136-146): the CRF head stops expanding blanks and the output becomes NTC
    contiguous -- which is what the HIP engine always produces. Delegates to SeqdistModel.use_hip."""
    def _no_blanks(m):
        if isinstance(m, LinearCRFEncoder):
            m.expand_blanks = False
    self.encoder.apply(_no_blanks)
    return self.use_hip(**kwargs)


def Model(config):
    model_config = {k: v for k, v in config["model"].items() if k != "package"}
    model = from_dict(model_config)
    model.config = config
    model.use_koi = types.MethodType(use_koi, model)
    return model
