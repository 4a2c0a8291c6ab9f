#!/usr/bin/env python3
"""
Generate the golden fixtures under tests/golden/ by EXECUTING THE REFERENCE (read-only at
/root/reference) on PyTorch-CPU.  Run in the build container only (the GPU box has no /root/reference):

    python tests/golden/make_golden.py

What is produced (all seeded, fp32 CPU):
  nn_<name>.npz      reference bonito/nn.py encoder forward: config (json), state_dict, input x, output y
  util_cases.json    reference bonito/util.py chunk / stitch / batchify / unbatchify on integer ramps
  crf_rc.npz         reference CTC_CRF.reverse_complement (crf/model.py:84-96) on seeded tensors          (make_crf_rc_fixture)
  crf_decode.npz     reference CTC_CRF.logZ / viterbi / path_to_str and SeqdistModel.decode_batch, executed with a torch
                     restatement of the four koi.ctc names they call (koi is closed source)                (make_crf_decode_fixture)
  reader_cases.npz   reference bonito/reader.py normalisation + trim in the order of bonito/pod5.py:61-62 (make_reader_fixture)
  tf_<name>.npz / ctc_<name>.npz   reference transformer / QuartzNet forward
Every fixture is also checked here against the oracle restatements (oracle/nn_ref.py), so a committed
fixture certifies "oracle == reference" at generation time; tests/ re-check the oracle against the files.

bonito/nn.py imports torch only and is loaded by file path.  bonito/util.py imports `toml` and
`parasail` at module level (util.py:16,19), absent here: inert stub modules are registered for those
two names before loading it -- the functions we call (chunk, stitch, batchify, unbatchify) do not touch them.
"""
import importlib.util
import json
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import nn_ref  # noqa: E402


def load_by_path(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def ref_nn():
    return load_by_path("ref_bonito_nn", os.path.join(REF, "bonito", "nn.py"))


def ref_util():
    for stub in ("toml", "parasail"):
        if stub not in sys.modules:
            sys.modules[stub] = types.ModuleType(stub)
    return load_by_path("ref_bonito_util", os.path.join(REF, "bonito", "util.py"))


def conv(insize, size, winlen, stride=1, act="swish"):
    return {"type": "convolution", "insize": insize, "size": size, "bias": True, "winlen": winlen,
            "stride": stride, "padding": winlen // 2, "activation": act, "norm": "batchnorm"}


def lstm_crf_config(c1, c2, feat, n_lstm, state_len, conv3_act="tanh", blank_score=2.0, clamp=5.0,
                    scale=None, crf_act=None, clamp_convs=None):
    subs = []
    for cfg in (conv(1, c1, 5), conv(c1, c2, 5)):
        subs.append(cfg)
        if clamp_convs:
            subs.append({"type": "clamp", "min": clamp_convs[0], "max": clamp_convs[1]})
    subs.append(conv(c2, feat, 19, stride=6, act=conv3_act))
    if clamp_convs:
        subs.append({"type": "clamp", "min": clamp_convs[0], "max": clamp_convs[1]})
    subs.append({"type": "permute", "dims": [2, 0, 1]})
    for i in range(n_lstm):
        subs.append({"type": "lstm", "size": feat, "insize": feat, "bias": True, "reverse": (n_lstm - i) % 2})
    crf = {"type": "linearcrfencoder", "insize": feat, "n_base": 4, "state_len": state_len, "bias": False}
    if blank_score is not None:
        crf["blank_score"] = blank_score
    if scale is not None:
        crf["scale"] = scale
    if crf_act is not None:
        crf["activation"] = crf_act
    subs.append(crf)
    if clamp is not None:
        subs.append({"type": "clamp", "min": -clamp, "max": clamp})
    return {"type": "serial", "sublayers": subs}


def randomise_bn_(model, gen):
    for m in model.modules():
        if isinstance(m, torch.nn.BatchNorm1d):
            with torch.no_grad():
                m.running_mean.copy_(torch.randn(m.running_mean.shape, generator=gen) * 0.1)
                m.running_var.copy_(torch.rand(m.running_var.shape, generator=gen) + 0.5)
                m.weight.copy_(1.0 + 0.1 * torch.randn(m.weight.shape, generator=gen))
                m.bias.copy_(0.1 * torch.randn(m.bias.shape, generator=gen))


def make_nn_fixture(nn, name, cfg, N, L, seed=25):
    torch.manual_seed(seed)
    model = nn.from_dict(cfg)
    gen = torch.Generator().manual_seed(seed + 1)
    randomise_bn_(model, gen)
    model.eval()
    nn_ref.round_params_to_half_(model)   # the engine stores fp16 weights; keep fixture weights representable
    x = torch.randn(N, 1, L, generator=gen).half().float()
    with torch.no_grad():
        y = model(x)
        y_oracle = nn_ref.forward(model, x)
    err = (y - y_oracle).abs().max().item()
    assert err < 2e-4, "oracle/nn_ref.py disagrees with reference nn.py on %s: %g" % (name, err)
    out = {"config": np.array(json.dumps(cfg)), "x": x.numpy(), "y": y.numpy()}
    for k, v in model.state_dict().items():
        out["sd/" + k] = v.numpy()
    path = os.path.join(HERE, "nn_%s.npz" % name)
    np.savez_compressed(path, **out)
    print("%-28s y%s  oracle-vs-reference max|d| = %.2e  -> %s (%d KiB)" %
          (name, tuple(y.shape), err, os.path.basename(path), os.path.getsize(path) // 1024))


def make_util_fixture(util):
    cases = []
    rng = np.random.default_rng(7)
    # (read length, chunksize, overlap, stride)
    grid = [(1000, 400, 100, 5), (1234, 400, 100, 5), (399, 400, 100, 5), (400, 400, 100, 5),
            (4000, 996, 96, 6), (5003, 996, 96, 6), (120, 996, 96, 6), (10000, 3996, 492, 6),
            (701, 400, 0, 5), (2000, 0, 0, 5)]
    for T, cs, ov, stride in grid:
        sig = torch.arange(T, dtype=torch.float32)
        ch = util.chunk(sig, cs, ov)
        case = {"T": T, "chunksize": cs, "overlap": ov, "stride": stride,
                "chunk_shape": list(ch.shape), "chunk_first": ch[:, 0, 0].tolist(),
                "chunk_last": ch[:, 0, -1].tolist()}
        if cs:
            # decode-like per-chunk output: one int per output step = global start sample of the step
            n, _, L = ch.shape
            steps = L // stride
            per = torch.stack([ch[i, 0, ::stride][:steps] for i in range(n)]).to(torch.int64)
            if T < cs:
                st = per[0, : int(np.floor(T / stride))]
            else:
                st = util.stitch(per, cs, ov, T, stride)
            case["stitched"] = st.tolist()
            if T >= cs and n > 1:
                case["stitched_rev"] = util.stitch(per, cs, ov, T, stride, reverse=True).tolist()
        cases.append(case)
    # batchify / unbatchify
    items = [("r%d" % i, torch.arange(n * 3, dtype=torch.float32).reshape(n, 1, 3) + 100 * i)
             for i, n in enumerate([3, 1, 7, 2, 5])]
    batches = list(util.batchify(iter(items), 4))
    b = {"batch_keys": [[[k, list(r)] for k, r in ks] for ks, _ in batches],
         "batch_shapes": [list(v.shape) for _, v in batches],
         "batch_sums": [float(v.sum()) for _, v in batches]}
    rebuilt = list(util.unbatchify(batches))
    b["unbatch"] = [[k, list(v.shape), float(v.sum())] for k, v in rebuilt]
    with open(os.path.join(HERE, "util_cases.json"), "w") as fh:
        json.dump({"chunk_stitch": cases, "batchify": b}, fh)
    print("util_cases.json: %d chunk/stitch cases, %d batches" % (len(cases), len(batches)))


def ref_transformer():
    """Import the reference's bonito.transformer.model without its un-installable dependencies.

    * a bare `bonito` package object (its real __init__ imports every CLI -> mappy/pysam) with the right
      __path__, so `bonito.nn`, `bonito.crf.model`, `bonito.transformer.model` are the REAL files;
    * inert stubs for koi / toml / parasail (only names are needed at import time);
    * flash_attn stubs that implement the PUBLIC semantics the builder assumed (SURVEY.md appendix C,
      [EXT], parity unpinned): RotaryEmbedding(dim, interleaved=False), GatedMlp, RMSNorm(x, residual);
    * torch.cuda.get_device_capability patched so MultiHeadAttention.attn_func takes its in-tree SDPA +
      sliding_window_mask branch (transformer/model.py:62-65), which defines the attention semantics.
    """
    import math
    pkg = types.ModuleType("bonito")
    pkg.__path__ = [os.path.join(REF, "bonito")]
    sys.modules["bonito"] = pkg
    for name in ("toml", "parasail", "koi", "koi.lstm", "koi.ctc", "koi.decode", "flash_attn", "flash_attn.layers",
                 "flash_attn.layers.rotary", "flash_attn.modules", "flash_attn.modules.mlp", "flash_attn.ops",
                 "flash_attn.ops.triton", "flash_attn.ops.triton.layer_norm"):
        sys.modules.setdefault(name, types.ModuleType(name))
    kc = sys.modules["koi.ctc"]
    kc.SequenceDist = type("SequenceDist", (), {"__init__": lambda self: None})
    for n in ("Max", "Log", "semiring", "logZ_cu", "viterbi_alignments", "logZ_cu_sparse", "bwd_scores_cu_sparse",
              "fwd_scores_cu_sparse"):
        setattr(kc, n, object)
    sys.modules["koi.decode"].beam_search = None
    sys.modules["koi.decode"].to_str = None

    class RotaryEmbedding(torch.nn.Module):
        def __init__(self, dim, interleaved=False):
            super().__init__()
            assert not interleaved
            self.dim = dim
            self.register_buffer("inv_freq", 1.0 / (10000 ** (torch.arange(0, dim, 2).float() / dim)), persistent=False)

        def forward(self, qkv):
            N, T, _, h, d = qkv.shape
            ang = torch.outer(torch.arange(T, dtype=torch.float32), self.inv_freq)
            cos, sin = ang.cos()[None, :, None, :], ang.sin()[None, :, None, :]
            out = qkv.clone()
            half = d // 2
            for i in (0, 1):
                x1, x2 = qkv[:, :, i, :, :half], qkv[:, :, i, :, half:]
                out[:, :, i, :, :half] = x1 * cos - x2 * sin
                out[:, :, i, :, half:] = x1 * sin + x2 * cos
            return out

    class GatedMlp(torch.nn.Module):
        def __init__(self, in_features, hidden_features=None, activation=None, bias1=True, bias2=True, multiple_of=1):
            super().__init__()
            self.activation = activation
            self.fc1 = torch.nn.Linear(in_features, 2 * hidden_features, bias=bias1)
            self.fc2 = torch.nn.Linear(hidden_features, in_features, bias=bias2)

        def forward(self, x):
            y, gate = self.fc1(x).chunk(2, dim=-1)
            return self.fc2(y * self.activation(gate))

    class RMSNorm(torch.nn.Module):
        def __init__(self, hidden_size, eps=1e-5):
            super().__init__()
            self.eps = eps
            self.weight = torch.nn.Parameter(torch.ones(hidden_size))

        def forward(self, x, residual=None):
            z = x if residual is None else x + residual
            return z * torch.rsqrt(z.pow(2).mean(-1, keepdim=True) + self.eps) * self.weight

    sys.modules["flash_attn"].flash_attn_qkvpacked_func = None
    sys.modules["flash_attn.layers.rotary"].RotaryEmbedding = RotaryEmbedding
    sys.modules["flash_attn.modules.mlp"].GatedMlp = GatedMlp
    sys.modules["flash_attn.ops.triton.layer_norm"].RMSNorm = RMSNorm
    torch.cuda.get_device_capability = lambda *a, **k: (7, 0)
    import importlib
    return importlib.import_module("bonito.transformer.model")


def make_transformer_fixture(name, d_model, nhead, dim_ff, depth, window, state_len, N, L, seed=25):
    tm = ref_transformer()
    alpha, beta = tm.deepnorm_params(depth)
    convs = [conv(1, 16, 5), conv(16, 16, 5), conv(16, 32, 9, stride=3), conv(32, 32, 9, stride=2),
             conv(32, d_model, 5, stride=2), {"type": "permute", "dims": [0, 2, 1]}]
    cfg = {"model": {
        "type": "seqdistmodel", "package": "bonito.transformer",
        "seqdist": {"state_len": state_len, "alphabet": ["N", "A", "C", "G", "T"]},
        "encoder": {
            "type": "namedserial",
            "conv": {"type": "serial", "sublayers": convs},
            "transformer_encoder": {"type": "stack", "depth": depth, "layer": {
                "type": "transformerencoderlayer", "d_model": d_model, "nhead": nhead, "dim_feedforward": dim_ff,
                "deepnorm_alpha": alpha, "deepnorm_beta": beta, "attn_window": list(window)}},
            "upsample": {"type": "linearupsample", "d_model": d_model, "scale_factor": 2},
            "crf": {"type": "linearcrfencoder", "insize": d_model, "n_base": 4, "state_len": state_len, "bias": False,
                    "scale": 5.0, "blank_score": 2.0, "expand_blanks": True, "permute": [1, 0, 2]},
        }}}
    torch.manual_seed(seed)
    model = tm.Model(cfg)
    gen = torch.Generator().manual_seed(seed + 1)
    randomise_bn_(model, gen)
    with torch.no_grad():
        for m in model.modules():       # non-trivial norm gains
            if type(m).__name__ == "RMSNorm":
                m.weight.copy_(1.0 + 0.1 * torch.randn(m.weight.shape, generator=gen))
    model.eval()
    nn_ref.round_params_to_half_(model)
    x = torch.randn(N, 1, L, generator=gen).half().float()
    with torch.no_grad():
        y = model(x)
        y_oracle = nn_ref.forward(model.encoder, x)
    err = (y - y_oracle).abs().max().item()
    assert err < 5e-4, "oracle/nn_ref.py disagrees with the reference transformer on %s: %g" % (name, err)
    out = {"config": np.array(json.dumps(cfg)), "x": x.numpy(), "y": y.numpy()}
    for k, v in model.state_dict().items():
        out["sd/" + k] = v.numpy()
    path = os.path.join(HERE, "tf_%s.npz" % name)
    np.savez_compressed(path, **out)
    print("%-28s y%s  oracle-vs-reference max|d| = %.2e  -> %s (%d KiB)" %
          (name, tuple(y.shape), err, os.path.basename(path), os.path.getsize(path) // 1024))


def ref_ctc():
    """The reference's bonito.ctc.model (pure torch apart from the fast_ctc_decode import, stubbed)."""
    if "bonito" not in sys.modules or not hasattr(sys.modules["bonito"], "__path__"):
        pkg = types.ModuleType("bonito")
        pkg.__path__ = [os.path.join(REF, "bonito")]
        sys.modules["bonito"] = pkg
    for name in ("toml", "parasail", "fast_ctc_decode"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["fast_ctc_decode"].beam_search = None
    sys.modules["fast_ctc_decode"].viterbi_search = None
    return load_by_path("bonito.ctc.model", os.path.join(REF, "bonito", "ctc", "model.py"))


def quartznet_config(blocks):
    return {"model": {"package": "bonito.ctc"}, "labels": {"labels": ["N", "A", "C", "G", "T"]},
            "input": {"features": 1}, "encoder": {"activation": "swish"},
            "qscore": {"scale": 0.9356, "bias": -0.1721},
            "block": [{"filters": f, "repeat": r, "kernel": [k], "stride": [s], "dilation": [1], "dropout": 0.05,
                       "residual": res, "separable": sep} for f, r, k, s, res, sep in blocks]}


def make_ctc_fixture(name, blocks, N, L, seed=25):
    cm = ref_ctc()
    cfg = quartznet_config(blocks)
    torch.manual_seed(seed)
    model = cm.Model(cfg)
    gen = torch.Generator().manual_seed(seed + 1)
    randomise_bn_(model, gen)
    model.eval()
    nn_ref.round_params_to_half_(model)
    x = torch.randn(N, 1, L, generator=gen).half().float()
    with torch.no_grad():
        y = model(x)
        y_oracle = nn_ref.ctc_forward(model, x)
    err = (y - y_oracle).abs().max().item()
    assert err < 2e-4, "oracle ctc_forward disagrees with reference ctc/model.py on %s: %g" % (name, err)
    out = {"config": np.array(json.dumps(cfg)), "x": x.numpy(), "y": y.numpy()}
    for k, v in model.state_dict().items():
        out["sd/" + k] = v.numpy()
    path = os.path.join(HERE, "ctc_%s.npz" % name)
    np.savez_compressed(path, **out)
    print("%-28s y%s  oracle-vs-reference max|d| = %.2e  -> %s (%d KiB)" %
          (name, tuple(y.shape), err, os.path.basename(path), os.path.getsize(path) // 1024))


def ref_reader():
    """The reference's bonito/reader.py (trim 122-143, normalisation 146-172): imports torch / numpy only at module level."""
    return load_by_path("ref_bonito_reader", os.path.join(REF, "bonito", "reader.py"))


def reader_case_signal(rng, i):
    """Synthetic pA-like trace; even cases carry an open-pore / adapter peak near the start (what `trim` looks for).
    tests/test_cli_cpu.py::_regen replays exactly this sequence of draws."""
    m = int(rng.integers(500, 20000))
    sig = rng.standard_normal(m).astype(np.float32)
    if i % 2 == 0:
        a = int(rng.integers(50, 400))
        b = a + int(rng.integers(60, 600))
        sig[a:b] += 4.0
    return (sig * 12.0 + 90.0).astype(np.float32)


def make_reader_fixture(n_cases=8):
    """reader_cases.npz: shift / scale / trim as computed by the REFERENCE functions in the order bonito/pod5.py:61-62 calls
    them (quantile normalisation of the pA signal, then trim with threshold = scale * 2.4 + shift)."""
    rd = ref_reader()
    rng = np.random.default_rng(5)
    meta = []
    for i in range(n_cases):
        raw = reader_case_signal(rng, i)
        shift, scale = rd.normalisation(raw, None, None)
        trim = rd.trim(raw, threshold=scale * 2.4 + shift)
        meta.append({"seed": i, "n": int(len(raw)), "shift": float(shift), "scale": float(scale), "trim": int(trim),
                     "peak": i % 2 == 0})
    np.savez_compressed(os.path.join(HERE, "reader_cases.npz"), meta=np.array(json.dumps(meta)))
    print("reader_cases.npz: %d cases, trims %s" % (len(meta), [m["trim"] for m in meta]))


def koi_ctc_stub():
    """A torch restatement of the four names of `koi.ctc` that bonito/crf/model.py:9-10 needs for CTC_CRF.logZ / posteriors /
    viterbi, so that the REFERENCE's own class body runs here. koi (ont-koi 0.5.4, requirements.txt:19) is closed binary +
    a thin Python layer; what is assumed of it [EXT]:
      * semirings `Log` / `Max` with `.one == 0.0` and sum = logsumexp / max, mul = +;
      * `logZ_cu_sparse(Ms, idx, alpha_0, beta_T, S)`: alpha_{t+1}[n, j] = S.sum_k(Ms[t, n, j, k] (x) alpha_t[n, idx[j, k]]),
        result S.sum_j(alpha_T[n, j] (x) beta_T[n, j]), differentiable w.r.t. Ms;
      * `SequenceDist.posteriors(scores, S)` = d logZ(scores, S).sum() / d scores (edge marginals; one-hot for Max).
    These are the definitions SURVEY.md section 8(a) D1 and appendix B.1 give; everything downstream of them in the fixture
    (reshape, argmax, `% len(alphabet)`, `// len(alphabet) % n_base`, path_to_str, decode_batch's `+ 1e-8`, `.log()`) is
    the reference's code, executed."""
    mod = types.ModuleType("koi.ctc")

    class Log:
        one = 0.0
        zero = -float("inf")

        @staticmethod
        def sum(x, dim):
            return torch.logsumexp(x, dim=dim)

    class Max:
        one = 0.0
        zero = -float("inf")

        @staticmethod
        def sum(x, dim):
            return torch.max(x, dim=dim).values

    def logZ_cu_sparse(Ms, idx, alpha_0, beta_T, S):
        T = Ms.shape[0]
        alpha = alpha_0
        ix = idx.to(torch.int64)
        for t in range(T):
            alpha = S.sum(Ms[t] + alpha[:, ix], dim=-1)
        return S.sum(alpha + beta_T, dim=-1)

    class SequenceDist:
        def __init__(self):
            pass

        def posteriors(self, scores, S=Log):
            with torch.enable_grad():
                x = scores.detach().clone().requires_grad_(True)
                lz = self.logZ(x, S)
                (g,) = torch.autograd.grad(lz.sum(), x)
            return g

    mod.Log, mod.Max, mod.semiring = Log, Max, object
    mod.SequenceDist, mod.logZ_cu_sparse = SequenceDist, logZ_cu_sparse
    for n in ("logZ_cu", "viterbi_alignments", "bwd_scores_cu_sparse", "fwd_scores_cu_sparse"):
        setattr(mod, n, None)
    return mod


def ref_crf_model():
    """bonito/crf/model.py itself, with `koi` replaced by koi_ctc_stub() and bonito.nn the real file."""
    pkg = types.ModuleType("bonito")
    pkg.__path__ = [os.path.join(REF, "bonito")]
    sys.modules["bonito"] = pkg
    for name in ("koi", "koi.lstm"):
        sys.modules[name] = types.ModuleType(name)
    sys.modules["koi.ctc"] = koi_ctc_stub()
    sys.modules.pop("bonito.crf.model", None)
    sys.modules.pop("bonito.crf", None)
    crf_pkg = types.ModuleType("bonito.crf")          # skip bonito/crf/__init__.py (it pulls basecall -> koi.decode)
    crf_pkg.__path__ = [os.path.join(REF, "bonito", "crf")]
    sys.modules["bonito.crf"] = crf_pkg
    import importlib
    return importlib.import_module("bonito.crf.model")


def make_crf_rc_fixture():
    """crf_rc.npz: the reference's CTC_CRF.reverse_complement (crf/model.py:84-96) on seeded [T, N, 5S] tensors."""
    cm = ref_crf_model()
    out = {}
    gen = torch.Generator().manual_seed(25)
    for sl in (1, 2, 3):
        sd = cm.CTC_CRF(sl, ["N", "A", "C", "G", "T"])
        x = torch.randn(7, 2, 5 * 4 ** sl, generator=gen).half().float()
        out["x%d" % sl] = x.numpy()
        out["y%d" % sl] = sd.reverse_complement(x).numpy()
    np.savez_compressed(os.path.join(HERE, "crf_rc.npz"), **out)
    print("crf_rc.npz: state_len 1..3, shapes %s" % [out["x%d" % s].shape for s in (1, 2, 3)])


def make_crf_decode_fixture():
    """crf_decode.npz: outputs of the REFERENCE's CTC_CRF.logZ / viterbi / path_to_str (crf/model.py:47-52,98-108) and
    SeqdistModel.decode_batch (:196-199) executed here with koi_ctc_stub() supplying the scan, on seeded fp16-valued scores in
    the reference layout [T, N, 5S]. Scores are drawn on a grid of 1/8 so that every partial sum is exact in fp32 and fp64
    alike: the Viterbi paths are then independent of the accumulation precision and bit-comparable."""
    cm = ref_crf_model()
    out = {}
    gen = torch.Generator().manual_seed(26)
    for sl, T, N in ((1, 40, 3), (2, 60, 4), (3, 120, 3), (4, 90, 2)):
        sd = cm.CTC_CRF(sl, ["N", "A", "C", "G", "T"])
        S = 4 ** sl
        x4 = (torch.randn(T, N, 4 * S, generator=gen) * 2.0).clamp(-5, 5)
        x4 = torch.round(x4 * 8) / 8                      # exact in fp16; sums of <= 120 of them exact in fp32
        x5 = torch.nn.functional.pad(x4.view(T, N, S, 4), (1, 0), value=2.0).view(T, N, 5 * S)      # nn.py:291-297
        paths = sd.viterbi(x5.double())                   # [T, N] in {0..4}: the reference's lines, fp64 scan
        logz = sd.logZ(x5.double())
        strs = [sd.path_to_str(p) for p in paths.T.numpy()]
        model = cm.SeqdistModel.__new__(cm.SeqdistModel)
        torch.nn.Module.__init__(model)
        model.seqdist = sd
        post_strs = cm.SeqdistModel.decode_batch(model, x5.double())
        out["x%d" % sl] = x4.permute(1, 0, 2).contiguous().half().numpy()        # engine layout [N, T, 4S], blank fixed at 2.0
        out["viterbi%d" % sl] = paths.T.contiguous().numpy().astype(np.int8)     # [N, T]
        out["logz%d" % sl] = logz.numpy()
        out["str%d" % sl] = np.array(json.dumps(strs))
        out["post_str%d" % sl] = np.array(json.dumps(post_strs))
        print("crf_decode sl=%d: T=%d N=%d bases %s, posterior-decoded %s" % (sl, T, N, [len(s) for s in strs], [len(s) for s in post_strs]))
    np.savez_compressed(os.path.join(HERE, "crf_decode.npz"), **out)


CTC_BLOCKS_SMALL = [(32, 1, 9, 3, False, False), (48, 2, 33, 1, True, True), (48, 3, 5, 1, True, True),
                    (64, 1, 29, 1, False, True), (40, 1, 15, 1, False, False)]


def main():
    if "--ctc-only" in sys.argv:
        make_ctc_fixture("quartz_small", CTC_BLOCKS_SMALL, N=3, L=600)
        return
    if "--crf-only" in sys.argv:
        make_crf_rc_fixture()
        make_crf_decode_fixture()
        make_reader_fixture()
        return
    if "--transformer-only" in sys.argv:
        make_transformer_fixture("d128_w31_32", 128, 2, 256, 2, (31, 32), 3, N=2, L=1200)
        make_transformer_fixture("d64_w127_128", 64, 1, 128, 1, (127, 128), 2, N=2, L=2400)
        return
    nn = ref_nn()
    # v4.3-style (tanh conv3, fixed blank, clamp +-5) at toy width; alternating directions 1,0,1
    make_nn_fixture(nn, "lstm32_sl2", lstm_crf_config(4, 16, 32, 3, 2), N=3, L=600)
    # hac-like proportions (16/16/H, 5 layers) at H=64, state_len 3, swish conv3
    make_nn_fixture(nn, "lstm64_sl3", lstm_crf_config(16, 16, 64, 5, 3, conv3_act="swish"), N=2, L=1200)
    # fast-like: H=96, state_len 3 -- the real `fast` width
    make_nn_fixture(nn, "lstm96_sl3", lstm_crf_config(16, 16, 96, 5, 3, conv3_act="swish"), N=2, L=900)
    # v4.0-style: clamps after every conv
    make_nn_fixture(nn, "lstm32_clampconv", lstm_crf_config(4, 16, 32, 2, 2, conv3_act="swish",
                                                             clamp_convs=(-0.5, 3.5)), N=2, L=480)
    # old-style head: tanh * 5, learned blank column (5S wide), no clamp
    make_nn_fixture(nn, "lstm32_oldstyle", lstm_crf_config(4, 16, 32, 2, 2, conv3_act="swish", blank_score=None,
                                                            clamp=None, scale=5.0, crf_act="tanh"), N=2, L=480)
    make_util_fixture(ref_util())
    make_transformer_fixture("d128_w31_32", 128, 2, 256, 2, (31, 32), 3, N=2, L=1200)
    make_transformer_fixture("d64_w127_128", 64, 1, 128, 1, (127, 128), 2, N=2, L=2400)
    make_ctc_fixture("quartz_small", CTC_BLOCKS_SMALL, N=3, L=600)
    make_crf_rc_fixture()
    make_crf_decode_fixture()
    make_reader_fixture()


if __name__ == "__main__":
    main()
