"""
Synthetic model configs and weights for benchmarking without network access: the `*@v5.0.0` model
directories are CDN downloads (/root/reference bonito/cli/download.py:31,39-66) that cannot be fetched
here, so bench.py / tests build the same ARCHITECTURES from seeded constructors (SURVEY.md section 8:
fast = conv 16/16/96, 5xLSTM-96, state_len 3; hac = conv 16/16/384, 5xLSTM-384, state_len 4;
sup LSTM (v4.3 template in tree) = 1024, state_len 5). Layout follows dna_r10.4.1@v4.3.toml:18-108.
"""
import torch

from bonito_amd import nn as bnn

LSTM_MODELS = {
    #        features, state_len
    "fast": (96, 3),
    "hac": (384, 4),
    "sup_lstm": (1024, 5),
}


def _conv(insize, size, winlen, stride=1, activation="swish"):
    return {"type": "convolution", "insize": insize, "size": size, "bias": True, "winlen": winlen,
            "stride": stride, "padding": winlen // 2, "activation": activation, "norm": "batchnorm"}


def lstm_crf_encoder_config(features, state_len, conv3_activation="tanh", n_lstm=5, blank_score=2.0, clamp=5.0):
    subs = [_conv(1, 16, 5), _conv(16, 16, 5), _conv(16, features, 19, stride=6, activation=conv3_activation),
            {"type": "permute", "dims": [2, 0, 1]}]
    for i in range(n_lstm):
        subs.append({"type": "lstm", "size": features, "insize": features, "bias": True,
                     "reverse": (n_lstm - i) % 2})
    subs.append({"type": "linearcrfencoder", "insize": features, "n_base": 4, "state_len": state_len,
                 "bias": False, "blank_score": blank_score})
    subs.append({"type": "clamp", "min": -clamp, "max": clamp})
    return {"type": "serial", "sublayers": subs}


def transformer_model_config(d_model=512, nhead=8, dim_ff=2048, depth=18, window=(127, 128), state_len=5,
                             conv_channels=(64, 64, 128, 128), batchsize=256, chunksize=12000, overlap=600):
    """The v5 `sup` architecture (layout of bonito/models/configs/dna_r10.4.1@v5.0.toml): conv x5 (strides
    1,1,3,2,2) -> `depth` DeepNorm transformer layers (rotary, sliding window) -> x2 linear upsample -> CRF head."""
    from bonito_amd.transformer.model import deepnorm_params
    alpha, beta = deepnorm_params(depth)
    c1, c2, c3, c4 = conv_channels
    convs = [_conv(1, c1, 5), _conv(c1, c2, 5), _conv(c2, c3, 9, stride=3), _conv(c3, c4, 9, stride=2),
             _conv(c4, d_model, 5, stride=2), {"type": "permute", "dims": [0, 2, 1]}]
    return {
        "model": {
            "type": "seqdistmodel", "package": "bonito_amd.transformer",
            "seqdist": {"state_len": state_len, "alphabet": ["N", "A", "C", "G", "T"]},
            "encoder": {
                "type": "namedserial",
                "conv": {"type": "serial", "sublayers": convs},
                "transformer_encoder": {"type": "stack", "depth": depth, "layer": {
                    "type": "transformerencoderlayer", "d_model": d_model, "nhead": nhead, "dim_feedforward": dim_ff,
                    "deepnorm_alpha": alpha, "deepnorm_beta": beta, "attn_window": list(window)}},
                "upsample": {"type": "linearupsample", "d_model": d_model, "scale_factor": 2},
                "crf": {"type": "linearcrfencoder", "insize": d_model, "n_base": 4, "state_len": state_len,
                        "bias": False, "scale": 5.0, "blank_score": 2.0, "expand_blanks": True, "permute": [1, 0, 2]},
            },
        },
        "basecaller": {"batchsize": batchsize, "chunksize": chunksize, "overlap": overlap},
    }


def make_transformer_model(seed=25, head_gain=1.0, **kw):
    from bonito_amd.transformer import Model
    from bonito_amd.nn import LinearCRFEncoder
    torch.manual_seed(seed)
    model = Model(transformer_model_config(**kw))
    randomise_batchnorm_(model, seed + 1)
    with torch.no_grad():
        for m in model.modules():
            if isinstance(m, LinearCRFEncoder):
                m.linear.weight.mul_(head_gain)
    model.eval()
    return model


def transformer_flops_per_chunk(d_model=512, dim_ff=2048, depth=18, window=(127, 128), state_len=5, chunksize=12000,
                                conv_channels=(64, 64, 128, 128)):
    """Algorithmic FLOPs of one chunk: convs + depth x (QKV, windowed QK^T + PV, out, fc1, fc2) + upsample + CRF."""
    L = chunksize
    c1, c2, c3, c4 = conv_channels
    l3 = (L + 8 - 9) // 3 + 1
    l4 = (l3 + 8 - 9) // 2 + 1
    T = (l4 + 4 - 5) // 2 + 1
    D, F, W = d_model, dim_ff, window[0] + window[1] + 1
    conv = 2 * (c1 * 5 * L + c1 * c2 * 5 * L + c2 * c3 * 9 * l3 + c3 * c4 * 9 * l4 + c4 * D * 5 * T)
    # keys follow the engine's profile classes (include/bonito_hip.h BH_PROF_*): "attention" = Wqkv + out_proj, "attention_core" = the
    # windowed QK^T + PV of the attention kernel, "mlp_fc1" = fc1 (+ SwiGLU), "mlp" = fc2
    parts = {"conv": conv, "attention": depth * (2 * T * D * 3 * D + 2 * T * D * D), "attention_core": depth * 2 * 2 * T * W * D,
             "mlp_fc1": depth * 2 * T * D * 2 * F, "mlp": depth * 2 * T * F * D, "other": 2 * T * D * 2 * D,
             "crf_linear": 2 * (2 * T) * D * 4 ** (state_len + 1)}
    parts["total"] = sum(parts.values())
    return parts


def model_config(name, batchsize=512, chunksize=10000, overlap=500):
    features, state_len = LSTM_MODELS[name]
    return {
        "model": {"package": "bonito_amd.crf"},
        "labels": {"labels": ["N", "A", "C", "G", "T"]},
        "input": {"features": 1},
        "global_norm": {"state_len": state_len},
        "encoder": lstm_crf_encoder_config(features, state_len),
        "basecaller": {"batchsize": batchsize, "chunksize": chunksize, "overlap": overlap},
    }


def randomise_batchnorm_(model, seed=26):
    """Non-trivial running statistics so that BatchNorm folding is exercised (SURVEY.md 8d 'Inputs')."""
    gen = torch.Generator().manual_seed(seed)
    for m in model.modules():
        if isinstance(m, torch.nn.BatchNorm1d):
            with torch.no_grad():
                m.running_mean.copy_(torch.randn(m.running_mean.shape, generator=gen) * 0.1)
                m.running_var.copy_(torch.rand(m.running_var.shape, generator=gen) + 0.5)
    return model


def make_model(name, seed=25, head_gain=24.0, **kw):
    """Seeded random-init model of the named architecture (reference CLI seed, cli/basecaller.py:178).
    A freshly initialised CRF head emits scores of std ~0.1, far below the fixed blank score 2.0, so
    every decoder would only ever "stay"; `head_gain` scales the head so that synthetic runs emit bases
    (trained heads use the whole +-5 clamp range)."""
    from bonito_amd.crf.model import Model
    from bonito_amd.nn import LinearCRFEncoder
    torch.manual_seed(seed)
    model = Model(model_config(name, **kw))
    randomise_batchnorm_(model, seed + 1)
    with torch.no_grad():
        for m in model.modules():
            if isinstance(m, LinearCRFEncoder):
                m.linear.weight.mul_(head_gain)
    model.eval()
    return model


def flops_per_chunk(name, chunksize=10000):
    """Algorithmic FLOPs (2*MAC) of one chunk through the encoder."""
    H, sl = LSTM_MODELS[name]
    L = chunksize
    T = (L + 18 - 19) // 6 + 1
    C = 4 ** (sl + 1)
    parts = {
        "conv": 2 * 16 * 5 * L + 2 * 16 * 16 * 5 * L + 2 * 16 * H * 19 * T,
        "lstm_gemm": 5 * 2 * H * 4 * H * T,
        "lstm_rec": 5 * 2 * H * 4 * H * T,
        "crf_linear": 2 * H * C * T,
    }
    parts["total"] = sum(parts.values())
    return parts
