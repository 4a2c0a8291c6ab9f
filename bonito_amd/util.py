"""
Host-side helpers of the hot path: chunking / stitching / batching and model loading. Behavioural
mirror of the hot-path part of /root/reference bonito/util.py (chunk 142-161, stitch 164-183, batchify
186-205, unbatchify 208-220, concat/select_range/size 66-102, load_symbol 223-234, match_names 239-248,
get_last_checkpoint 251-256, set_config_defaults 259-268, load_model/_load_model 271-311, phred 105-111,
mean_qscore_from_qstring 114-121). Pure host logic; parity-tested against fixtures generated from the
reference functions (tests/golden/make_golden.py).
"""
import os
import re
from collections import OrderedDict
from glob import glob
from importlib import import_module
from itertools import groupby
from operator import itemgetter
from pathlib import Path

import numpy as np
import torch

try:                      # python >= 3.11
    import tomllib as _toml
except ImportError:       # this image: tomli
    import tomli as _toml

__dir__ = Path(__file__).parent
__models_dir__ = __dir__ / "models"

# config.toml files written for the reference name its packages; route them to the HIP engine.
PACKAGE_ALIASES = {
    "bonito.crf": "bonito_amd.crf",
    "bonito.transformer": "bonito_amd.transformer",
    "bonito.ctc": "bonito_amd.ctc",
}


def effective_cpu_count():
    """CPUs this process may actually burn: the affinity mask capped by the cgroup CPU quota (a container that sees
    256 cores but owns a 16-core quota is throttled for the rest of the CFS period once its threads spin past it)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    quota = None
    try:
        q, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]          # cgroup v2
        if q != "max":
            quota = int(q) / int(period)
    except (OSError, ValueError):
        try:                                                                     # cgroup v1
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / period
        except (OSError, ValueError):
            pass
    if quota is not None:
        n = min(n, max(1, int(quota)))
    return max(1, n)


def limit_host_threads(max_threads=4):
    """The host side of the hot path is a handful of small tensor copies: a wide intra-op thread pool only spins.
    On a CPU-quota'd container that spinning exhausts the quota and stalls the process for the rest of the 100 ms
    CFS period (measured on the MI355X boxes: 128 OpenMP threads against a 16-core quota -> every third decode wait
    took 80 ms instead of 9). Returns the thread count now in force."""
    try:                      # one process per GPU (torchrun): the ranks of a host share its CPU budget
        local_world = max(1, int(os.environ.get("LOCAL_WORLD_SIZE", "1")))
    except ValueError:
        local_world = 1
    n = max(1, min(torch.get_num_threads(), max_threads, effective_cpu_count() // local_world))
    torch.set_num_threads(n)
    return n


def load_toml(path):
    with open(path, "rb") as fh:
        return _toml.load(fh)


def init(seed, device, deterministic=True):
    """Seed the host RNGs (reference util.py:40-53). There is no cuDNN to configure here."""
    import random
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if device == "cpu":
        return
    if not torch.cuda.is_available():
        raise RuntimeError("no HIP device visible")


def permute(x, input_layout, output_layout):
    """permute(x, 'TNC', 'NTC')"""
    if input_layout == output_layout:
        return x
    return x.permute(*[input_layout.index(d) for d in output_layout])


def concat(xs, dim=0):
    """Type agnostic concat (tensors, arrays, lists, strings, dicts of those)."""
    head = xs[0]
    if isinstance(head, torch.Tensor):
        return torch.cat(xs, dim=dim)
    if isinstance(head, np.ndarray):
        return np.concatenate(xs, axis=dim)
    if isinstance(head, list):
        return [item for x in xs for item in x]
    if isinstance(head, str):
        return "".join(xs)
    if isinstance(head, dict):
        return {k: concat([x[k] for x in xs], dim) for k in head.keys()}
    raise TypeError(type(head))


def select_range(x, start, end, dim=0):
    """Type agnostic range select."""
    if isinstance(x, dict):
        return {k: select_range(v, start, end, dim) for k, v in x.items()}
    if dim == 0 or isinstance(x, list):
        return x[start:end]
    return x[(slice(None),) * dim + (slice(start, end),)]


def size(x, dim=0):
    """Type agnostic size."""
    if hasattr(x, "shape"):
        return x.shape[dim]
    if dim == 0:
        return len(x)
    raise TypeError(type(x))


def phred(prob, scale=1.0, bias=0.0):
    """Probability -> ASCII phred character (error floor 1e-4, i.e. Q40 before scale/bias)."""
    err = max(1 - prob, 1e-4)
    q = -10 * np.log10(err) * scale + bias
    return chr(int(np.round(q) + 33))


def mean_qscore_from_qstring(qstring):
    """Mean q-score of a phred string, averaged in error-probability space."""
    if len(qstring) == 0:
        return 0.0
    qs = np.array(qstring, "c").view(np.uint8) - 33
    mean_err = np.exp(qs * (-np.log(10) / 10.0)).mean()
    return -10 * np.log10(max(mean_err, 1e-4))


def chunk(signal, chunksize, overlap):
    """Cut a read [T] / [C, T] into overlapping chunks [n, C, chunksize].

    * chunksize == 0: the whole read is one chunk;
    * T < chunksize: the read is tiled to fill exactly one chunk;
    * otherwise windows advance by chunksize-overlap from offset `stub`, and when stub > 0 an extra
      first chunk signal[:chunksize] is prepended so no sample is dropped."""
    if signal.ndim == 1:
        signal = signal.unsqueeze(0)
    T = signal.shape[-1]
    if chunksize == 0:
        return signal[None, :]
    if T < chunksize:
        reps, overhang = divmod(chunksize, T)
        tiled = torch.cat((torch.from_numpy(np.tile(signal, reps)), signal[..., :overhang]), dim=-1)
        return tiled[None, :]
    step = chunksize - overlap
    stub = (T - overlap) % step
    chunks = signal[..., stub:].unfold(-1, chunksize, step).movedim(-2, 0)
    if stub > 0:
        chunks = torch.cat([signal[None, ..., :chunksize], chunks], dim=0)
    return chunks


def stitch(chunks, chunksize, overlap, length, stride, reverse=False):
    """Inverse of `chunk` in output steps: drop half an overlap at every interior edge and concatenate."""
    if chunks.shape[0] == 1:
        return chunks.squeeze(0)
    semi = overlap // 2
    start, end = semi // stride, (chunksize - semi) // stride
    stub = (length - overlap) % (chunksize - overlap)
    first_end = (stub + semi) // stride if stub > 0 else end
    if reverse:
        chunks = list(chunks)
        return concat([chunks[-1][:-start], *(x[-end:-start] for x in reversed(chunks[1:-1])),
                       chunks[0][-first_end:]])
    if isinstance(chunks, (torch.Tensor, np.ndarray)) and chunks.ndim == 2:
        # same pieces as below, the interior ones gathered by one strided copy instead of one slice object per chunk
        return concat([chunks[0, :first_end], chunks[1:-1, start:end].reshape(-1), chunks[-1, start:]])
    return concat([chunks[0, :first_end], *chunks[1:-1, start:end], chunks[-1, start:]])


def batchify(items, batchsize, dim=0):
    """Pack (key, value) items into batches of exactly `batchsize` along `dim` (the last one may be
    short). Yields (keys, batch) with keys = ((key, (lo, hi)), ...) locating each piece in the batch."""
    stack, pos = [], 0
    for key, value in items:
        n = size(value, dim)
        breaks = range(batchsize - pos, n, batchsize)
        for lo, hi in zip([0, *breaks], [*breaks, n]):
            stack.append(((key, (pos, pos + hi - lo)), select_range(value, lo, hi, dim)))
            if pos + hi - lo == batchsize:
                ks, vs = zip(*stack)
                yield ks, concat(vs, dim)
                stack, pos = [], 0
            else:
                pos += hi - lo
    if stack:
        ks, vs = zip(*stack)
        yield ks, concat(vs, dim)


def unbatchify(batches, dim=0):
    """Regroup batched results by key (relies on in-order batches, like the reference)."""
    pieces = ((key, select_range(v, lo, hi, dim)) for sub, v in batches for key, (lo, hi) in sub)
    return ((key, concat([v for _, v in group], dim)) for key, group in groupby(pieces, itemgetter(0)))


def _model_dir(dirname):
    if not os.path.isdir(dirname) and os.path.isdir(os.path.join(__models_dir__, dirname)):
        return os.path.join(__models_dir__, dirname)
    return dirname


def load_symbol(config, symbol):
    """`symbol` ('Model' / 'basecall') of the package named by config['model']['package']; reference
    package names are mapped to their bonito_amd counterparts."""
    if not isinstance(config, dict):
        config = load_toml(os.path.join(_model_dir(config), "config.toml"))
    package = config["model"]["package"]
    package = PACKAGE_ALIASES.get(package, package)
    return getattr(import_module(package), symbol)


def match_names(state_dict, model):
    """Map checkpoint keys onto model keys by sorted (shape, position), so differently named but
    identically shaped/ordered checkpoints load (reference util.py:239-248)."""
    def ordered(sd):
        triples = sorted((tuple(v.shape), i, k) for i, (k, v) in enumerate(sd.items()))
        return [k for _, _, k in triples], [s for s, _, _ in triples]
    k1, s1 = ordered(state_dict)
    k2, s2 = ordered(model.state_dict())
    assert s1 == s2, "checkpoint and model disagree on parameter shapes"
    remap = dict(zip(k1, k2))
    return OrderedDict((k, remap[k]) for k in state_dict.keys())


def get_last_checkpoint(dirname):
    files = glob(os.path.join(dirname, "weights_*.tar"))
    if not files:
        raise FileNotFoundError("no model weights found in '%s'" % dirname)
    last = max(int(re.sub(r".*_([0-9]+).tar", r"\1", f)) for f in files)
    return os.path.join(dirname, "weights_%s.tar" % last)


def set_config_defaults(config, chunksize=None, batchsize=None, overlap=None, quantize=False):
    """[basecaller] defaults 4000/500/64; explicit arguments win over the config (util.py:259-268)."""
    params = config.get("basecaller", {})
    params["chunksize"] = chunksize or params.get("chunksize", 4000)
    params["overlap"] = overlap if overlap is not None else params.get("overlap", 500)
    params["batchsize"] = batchsize or params.get("batchsize", 64)
    params["quantize"] = params.get("quantize") if quantize is None else quantize
    config["basecaller"] = params
    return config


def load_model(dirname, device, weights=None, half=True, chunksize=None, batchsize=None, overlap=None,
               quantize=False, use_koi=False, use_hip=True):
    """Load config.toml + weights_N.tar from a model directory onto the HIP engine."""
    dirname = _model_dir(dirname)
    weights = get_last_checkpoint(dirname) if weights is None else os.path.join(dirname, "weights_%s.tar" % weights)
    config = set_config_defaults(load_toml(os.path.join(dirname, "config.toml")), chunksize, batchsize, overlap, quantize)
    return _load_model(weights, config, device, half, use_koi or use_hip)


def _load_model(model_file, config, device, half=True, use_koi=True):
    device = torch.device(device)
    model = load_symbol(config, "Model")(config)
    bc = config["basecaller"]
    if use_koi:
        bc["chunksize"] -= bc["chunksize"] % model.stride
        bc["overlap"] -= bc["overlap"] % (model.stride * 2)   # even multiple of stride for stitching
        model.use_koi(batchsize=bc["batchsize"], chunksize=bc["chunksize"], quantize=bc["quantize"])
    state = torch.load(model_file, map_location="cpu") if isinstance(model_file, str) else model_file
    state = {k2: state[k1] for k1, k2 in match_names(state, model).items()}
    model.load_state_dict(OrderedDict((k.replace("module.", ""), v) for k, v in state.items()))
    if half:
        model = model.half()
    model.eval()
    model.to(device)
    return model
