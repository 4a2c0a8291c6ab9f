"""
Layer registry of the MI355X engine -- the host-side mirror of the reference's ``bonito.nn``
(/root/reference bonito/nn.py) so that existing ``config.toml`` files and ``weights_N.tar``
checkpoints drop in unchanged:

* same registry surface: ``layers``, ``register``, ``from_dict``, ``to_dict``, ``fuse_bn_`` (nn.py:13-19,418-454);
* same layer names and constructor keywords (``convolution``, ``lstm``, ``linearcrfencoder``, ``clamp``,
  ``permute``, ``serial``, ``stack``, ``namedserial``, ``linearupsample``, ``batchnorm``, ``reverse`` ...);
* same parameter names/shapes/initialisation order, hence identical ``state_dict()`` keys and identical
  seeded weights (the shape-sorted remap of util.match_names, util.py:239-248, keeps working).

The classes here are PARAMETER CONTAINERS. None of them computes anything: every ``forward`` raises.
All arithmetic happens in the hand-written HIP kernels behind ``libbonito_hip.so``; a model becomes
runnable through ``model.use_hip()`` / ``model.use_koi()`` (bonito_amd/crf/model.py), which lowers the
module tree to a ``bh_layer_t`` chain (bonito_amd/engine.py). There is no PyTorch compute fallback.
"""
from collections import OrderedDict

import torch
from torch.nn import Module

layers = {}


def register(layer):
    """Add `layer` to the registry under its lower-cased class name (reference nn.py:16-19)."""
    layer.name = layer.__name__.lower()
    layers[layer.name] = layer
    return layer


class NoTorchCompute(RuntimeError):
    pass


def _no_forward(self, *args, **kwargs):
    raise NoTorchCompute(
        "%s is a parameter container of the MI355X engine and has no PyTorch forward; "
        "call model.use_hip() (or use_koi()) and run the model object instead" % type(self).__name__
    )


class _Act(Module):
    """Activation marker (lowered to a BH_ACT_* epilogue id)."""
    forward = _no_forward


@register
class ReLU(_Act):
    pass


@register
class Tanh(_Act):
    pass


@register
class Swish(_Act):
    pass


def _activation(spec):
    """`spec` is a registry name, None, or already a module (reference nn.py:227)."""
    if spec is None:
        return None
    if isinstance(spec, str):
        return layers[spec]()
    return spec


@register
class Linear(Module):
    def __init__(self, in_features, out_features, bias=True):
        super().__init__()
        self.in_features, self.out_features, self.bias = in_features, out_features, bias
        self.linear = torch.nn.Linear(in_features, out_features, bias=bias)

    forward = _no_forward

    def to_dict(self, include_weights=False):
        cfg = {"in_features": self.in_features, "out_features": self.out_features, "bias": self.bias}
        if include_weights:
            cfg["params"] = {"W": self.linear.weight, "b": self.linear.bias if self.bias is not None else []}
        return cfg


@register
class Clamp(Module):
    def __init__(self, min, max):
        super().__init__()
        self.min, self.max = min, max

    forward = _no_forward

    def to_dict(self, include_weights=False):
        return {"min": self.min, "max": self.max}


@register
class Serial(torch.nn.Sequential):
    def __init__(self, sublayers):
        super().__init__(*sublayers)

    forward = _no_forward

    def to_dict(self, include_weights=False):
        return {"sublayers": [to_dict(m, include_weights) for m in self._modules.values()]}

    def __repr__(self):
        return torch.nn.ModuleList.__repr__(self)


@register
class Stack(Serial):
    """`depth` copies of one layer description (reference nn.py:101-113)."""

    @classmethod
    def from_dict(cls, model_dict, layer_types=None):
        return cls([from_dict(model_dict["layer"], layer_types) for _ in range(model_dict["depth"])])

    def to_dict(self, include_weights=False):
        if include_weights:
            raise NotImplementedError
        dicts = [to_dict(m) for m in self]
        assert all(d == dicts[0] for d in dicts[1:]), "all layers should be the same"
        return {"layer": dicts[0], "depth": len(self)}


@register
class NamedSerial(torch.nn.Sequential):
    @classmethod
    def from_dict(cls, model_dict, layer_types=None):
        return cls({name: from_dict(cfg, layer_types) for name, cfg in model_dict.items()})

    def __init__(self, named):
        super().__init__(OrderedDict(named.items()))

    forward = _no_forward

    def to_dict(self, include_weights=False):
        if include_weights:
            raise NotImplementedError
        return {name: to_dict(m) for name, m in self.named_children()}


class MakeContiguous(Module):
    forward = _no_forward


@register
class LinearUpsample(Module):
    """Linear d_model -> scale_factor*d_model then reshape [N,L,s*E] -> [N,s*L,E] (reference nn.py:140-171)."""

    def __init__(self, d_model, scale_factor, batch_first=True):
        super().__init__()
        self.d_model, self.scale_factor, self.batch_first = d_model, scale_factor, batch_first
        self.linear = torch.nn.Linear(d_model, scale_factor * d_model)

    forward = _no_forward

    def output_stride(self, input_stride):
        return input_stride // self.scale_factor

    def to_dict(self, include_weights=False):
        if include_weights:
            raise NotImplementedError
        return {"d_model": self.d_model, "scale_factor": self.scale_factor, "batch_first": self.batch_first}


@register
class Reverse(Module):
    def __init__(self, sublayers):
        super().__init__()
        self.layer = Serial(sublayers) if isinstance(sublayers, list) else sublayers

    forward = _no_forward

    def to_dict(self, include_weights=False):
        if isinstance(self.layer, Serial):
            return self.layer.to_dict(include_weights)
        return {"sublayers": to_dict(self.layer, include_weights)}


@register
class BatchNorm(Module):
    def __init__(self, num_features, eps=1e-05, momentum=0.1, affine=True, track_running_stats=True):
        super().__init__()
        self.bn = torch.nn.BatchNorm1d(num_features, eps, momentum, affine, track_running_stats)

    forward = _no_forward

    def to_dict(self, include_weights=False):
        bn = self.bn
        cfg = {"num_features": bn.num_features, "eps": bn.eps, "momentum": bn.momentum, "affine": bn.affine,
               "track_running_stats": bn.track_running_stats}
        if include_weights:
            params = {}
            if bn.affine:
                params["W"], params["b"] = bn.weight, bn.bias
            if bn.track_running_stats:
                params["running_mean"], params["running_var"] = bn.running_mean, bn.running_var
            cfg["params"] = params
        return cfg


@register
class Convolution(Module):
    def __init__(self, insize, size, winlen, stride=1, padding=0, bias=True, activation=None, norm=None):
        super().__init__()
        self.conv = torch.nn.Conv1d(insize, size, winlen, stride=stride, padding=padding, bias=bias)
        self.activation = _activation(activation)
        if isinstance(norm, dict):
            self.norm = from_dict(norm)
        elif isinstance(norm, str):
            self.norm = layers[norm](size)
        else:
            self.norm = norm

    forward = _no_forward

    def folded(self):
        """fp32 (weight [Cout,Cin,K], bias [Cout]) with an eval-mode BatchNorm folded in:
        W' = W*g/sqrt(var+eps), b' = (b-mean)*g/sqrt(var+eps)+beta  (what fuse_conv_bn_eval does,
        reference nn.py:447-454). Folding is done in fp32 and rounded to fp16 once by the engine."""
        w = self.conv.weight.detach().float().cpu()
        b = (self.conv.bias.detach().float().cpu() if self.conv.bias is not None
             else torch.zeros(w.shape[0]))
        if isinstance(self.norm, BatchNorm):
            bn = self.norm.bn
            mean = bn.running_mean.detach().float().cpu()
            var = bn.running_var.detach().float().cpu()
            g = bn.weight.detach().float().cpu() if bn.affine else torch.ones_like(mean)
            beta = bn.bias.detach().float().cpu() if bn.affine else torch.zeros_like(mean)
            s = g * torch.rsqrt(var + bn.eps)
            w = w * s[:, None, None]
            b = (b - mean) * s + beta
        elif self.norm is not None:
            raise NoTorchCompute("unsupported norm %r on a convolution" % (self.norm,))
        return w.contiguous(), b.contiguous()

    def to_dict(self, include_weights=False):
        c = self.conv
        cfg = {"insize": c.in_channels, "size": c.out_channels, "bias": c.bias is not None,
               "winlen": c.kernel_size[0], "stride": c.stride[0], "padding": c.padding[0]}
        if self.activation is not None:
            cfg["activation"] = self.activation.name
        if self.norm is not None:
            cfg["norm"] = to_dict(self.norm, include_weights)
            if not include_weights and self.norm.name in layers:
                if cfg["norm"] == to_dict(layers[self.norm.name](cfg["size"])):
                    cfg["norm"] = self.norm.name
        if include_weights:
            cfg["params"] = {"W": c.weight, "b": c.bias if c.bias is not None else []}
        return cfg


@register
class LinearCRFEncoder(Module):
    def __init__(self, insize, n_base, state_len, bias=True, scale=None, activation=None, blank_score=None,
                 expand_blanks=True, permute=None):
        super().__init__()
        self.scale, self.n_base, self.state_len = scale, n_base, state_len
        self.blank_score, self.expand_blanks, self.permute = blank_score, expand_blanks, permute
        size = (n_base + 1) * n_base ** state_len if blank_score is None else n_base ** (state_len + 1)
        self.linear = torch.nn.Linear(insize, size, bias=bias)
        self.activation = _activation(activation)

    forward = _no_forward

    def to_dict(self, include_weights=False):
        cfg = {"insize": self.linear.in_features, "n_base": self.n_base, "state_len": self.state_len,
               "bias": self.linear.bias is not None, "scale": self.scale, "blank_score": self.blank_score,
               "expand_blanks": self.expand_blanks}
        if self.activation is not None:
            cfg["activation"] = self.activation.name
        if self.permute is not None:
            cfg["permute"] = self.permute
        if include_weights:
            cfg["params"] = {"W": self.linear.weight,
                             "b": self.linear.bias if self.linear.bias is not None else []}
        return cfg

    def extra_repr(self):
        text = "n_base={}, state_len={}, scale={}, blank_score={}, expand_blanks={}".format(
            self.n_base, self.state_len, self.scale, self.blank_score, self.expand_blanks)
        return text + (", permute={}".format(self.permute) if self.permute else "")


@register
class Permute(Module):
    def __init__(self, dims):
        super().__init__()
        self.dims = dims

    forward = _no_forward

    def to_dict(self, include_weights=False):
        return {"dims": self.dims}

    def extra_repr(self):
        return "dims={}".format(self.dims)


def truncated_normal(size, dtype=torch.float32, device=None, num_resample=5):
    """First of `num_resample` N(0,1) draws that lies in (-2, 2), clamped (reference nn.py:347-350)."""
    draws = torch.empty(size + (num_resample,), dtype=torch.float32, device=device).normal_()
    first_ok = ((draws < 2) & (draws > -2)).max(-1, keepdim=True)[1]
    return torch.clamp_(draws.gather(-1, first_ok).squeeze(-1), -2, 2)


class RNNWrapper(Module):
    """One torch RNN held for its parameters; init matches the reference (nn.py:353-393): per-gate
    orthogonal weights, 0.5*truncated-normal input bias, zero (frozen) state bias."""

    def __init__(self, rnn_type, *args, reverse=False, orthogonal_weight_init=True, disable_state_bias=True,
                 bidirectional=False, **kwargs):
        super().__init__()
        if reverse and bidirectional:
            raise Exception("'reverse' and 'bidirectional' should not both be set to True")
        self.reverse = reverse
        self.rnn = rnn_type(*args, bidirectional=bidirectional, **kwargs)
        self.init_orthogonal(orthogonal_weight_init)
        self.init_biases()
        if disable_state_bias:
            self.disable_state_bias()

    forward = _no_forward

    def init_biases(self, types=("bias_ih",)):
        for name, param in self.rnn.named_parameters():
            if any(k in name for k in types):
                with torch.no_grad():
                    param.set_(0.5 * truncated_normal(param.shape, dtype=param.dtype, device=param.device))

    def init_orthogonal(self, types=True):
        if not types:
            return
        if types is True:
            types = ("weight_ih", "weight_hh")
        hs = self.rnn.hidden_size
        for name, x in self.rnn.named_parameters():
            if any(k in name for k in types):
                for i in range(0, x.size(0), hs):
                    torch.nn.init.orthogonal_(x[i:i + hs])

    def disable_state_bias(self):
        for name, x in self.rnn.named_parameters():
            if "bias_hh" in name:
                x.requires_grad = False
                x.zero_()

    def extra_repr(self):
        return "reverse={}".format(bool(self.reverse))


@register
class LSTM(RNNWrapper):
    def __init__(self, size, insize, bias=True, reverse=False):
        super().__init__(torch.nn.LSTM, insize, size, bias=bias, reverse=reverse)

    def to_dict(self, include_weights=False):
        r = self.rnn
        cfg = {"size": r.hidden_size, "insize": r.input_size, "bias": r.bias, "reverse": self.reverse}
        if include_weights:
            cfg["params"] = {"iW": r.weight_ih_l0.reshape(4, r.hidden_size, r.input_size),
                             "sW": r.weight_hh_l0.reshape(4, r.hidden_size, r.hidden_size),
                             "b": r.bias_ih_l0.reshape(4, r.hidden_size)}
        return cfg


def to_dict(layer, include_weights=False):
    if hasattr(layer, "to_dict"):
        return {"type": layer.name, **layer.to_dict(include_weights)}
    return {"type": layer.name}


def from_dict(model_dict, layer_types=None):
    """Build a module tree from a config dict (reference nn.py:424-444)."""
    if not isinstance(model_dict, dict):
        return model_dict   # concrete objects pass through
    cfg = dict(model_dict)
    registry = layers if layer_types is None else layer_types
    cls = registry[cfg.pop("type")]
    if hasattr(cls, "from_dict"):
        return cls.from_dict(cfg, registry)
    if "sublayers" in cfg:
        sub = cfg["sublayers"]
        cfg["sublayers"] = [from_dict(x, registry) for x in sub] if isinstance(sub, list) else from_dict(sub, registry)
    try:
        return cls(**cfg)
    except Exception as exc:
        raise Exception("Failed to build layer of type {} with args {}".format(cls, cfg)) from exc


def fuse_bn_(m):
    """``model.apply(fuse_bn_)``: eval mode, and fold a Convolution's BatchNorm into its conv
    (reference nn.py:447-454, called at cli/basecaller.py:61). Folded in fp32 whatever the parameter dtype."""
    m.training = False
    if hasattr(m, "_drop_engine"):      # a model object: its lowered HIP engine holds a snapshot of the old weights
        m._drop_engine()
    if isinstance(m, Convolution) and isinstance(m.norm, BatchNorm):
        w, b = m.folded()
        dtype, device = m.conv.weight.dtype, m.conv.weight.device
        with torch.no_grad():
            m.conv.weight = torch.nn.Parameter(w.to(device=device, dtype=dtype), requires_grad=False)
            m.conv.bias = torch.nn.Parameter(b.to(device=device, dtype=dtype), requires_grad=False)
        m.norm = None
