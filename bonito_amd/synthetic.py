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
