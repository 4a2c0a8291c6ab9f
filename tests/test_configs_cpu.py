"""Every model config shipped with the reference (bonito/models/configs/*.toml, copied as TEST DATA into
tests/golden/configs/) must build through the bonito_amd registry, lower to a bh_layer_t chain, and load a state dict
whose keys are renamed (util.match_names, reference util.py:239-248). CPU only: no compute, no engine."""
import glob
import os
from collections import OrderedDict

import pytest
import torch

from conftest import GOLDEN
from bonito_amd import _lib, engine, util

CONFIGS = sorted(glob.glob(os.path.join(GOLDEN, "configs", "*.toml")))


def build_from_config(path, seed=25):
    cfg = util.set_config_defaults(util.load_toml(path))
    torch.manual_seed(seed)
    model = util.load_symbol(cfg, "Model")(cfg)
    model.eval()
    return cfg, model


def test_all_seven_reference_configs_are_present():
    assert len(CONFIGS) == 7, CONFIGS


@pytest.mark.parametrize("path", CONFIGS, ids=[os.path.basename(p) for p in CONFIGS])
def test_config_builds_and_lowers(path):
    cfg, model = build_from_config(path)
    pkg = cfg["model"]["package"]
    if pkg == "bonito.ctc":
        low = engine.lower_ctc(model)
        assert low.descs[-1].kind == _lib.BH_LAYER_CTC_DECODER
        assert model.stride == 3
    else:
        low = engine.lower(model.encoder)
        kinds = [d.kind for d in low.descs]
        assert _lib.BH_LAYER_LINEAR_CRF in kinds
        assert model.stride == (6 if "v5.0" in path or "v4.3" in path else 5)
        if "v4.0" in path:        # LSTM-1024 -> Linear 1024->256 -> CRF head (toml lines 101-104)
            i = kinds.index(_lib.BH_LAYER_LINEAR)
            assert (low.descs[i].in_size, low.descs[i].out_size) == (1024, 256)
            assert kinds[i + 1] == _lib.BH_LAYER_LINEAR_CRF and low.descs[i + 1].in_size == 256
    assert len(low.descs) > 3


@pytest.mark.parametrize("path", CONFIGS, ids=[os.path.basename(p) for p in CONFIGS])
def test_config_loads_renamed_state_dict(path):
    """A checkpoint with foreign key names but the same shape sequence loads through match_names and lands on the
    same parameters (what util._load_model does for every weights_N.tar)."""
    cfg, model = build_from_config(path, seed=1)
    _, donor = build_from_config(path, seed=2)
    renamed = OrderedDict(("module.w%03d" % i, v.clone()) for i, (k, v) in enumerate(donor.state_dict().items()))
    remap = util.match_names(renamed, model)
    state = {k2: renamed[k1] for k1, k2 in remap.items()}
    model.load_state_dict(state)
    for (k, v), (k2, v2) in zip(model.state_dict().items(), donor.state_dict().items()):
        assert k == k2 and torch.equal(v, v2), k


def test_engine_call_size_policy():
    """`batches_per_call` (crf/basecall.py): the 192...512-wide fp16 recurrent models get engine calls of two paired launches per layer
    (2048 chunks at 384 hidden units), capped by 8 GiB of scores per call; every other model, and the 8-bit path, one batch per call.
    Pure host logic: which chunks share a call never changes a result (GPU test `test_batches_per_engine_call_do_not_change_the_calls`)."""
    from bonito_amd import synthetic
    from bonito_amd.crf.basecall import auto_lanes, batches_per_call
    hac = synthetic.make_model("hac", batchsize=64, chunksize=1200)
    assert (auto_lanes(hac), auto_lanes(hac, True)) == (1, 2)
    assert auto_lanes(synthetic.make_model("fast", batchsize=16, chunksize=1200)) == 3
    # the 8-bit kernel exchanges between workgroups at EVERY width it covers: a quantised 96-wide model is a one-lane model
    # (advisor finding, round 3: it was given the three lanes of the fp16 ring-in-a-workgroup kernel)
    from bonito_amd.crf.basecall import max_lanes, q8_covers
    fastq = synthetic.make_model("fast", batchsize=16, chunksize=1200)
    assert (max_lanes(fastq), max_lanes(fastq, True), auto_lanes(fastq, True)) == (1 << 30, 1, 1)
    assert batches_per_call(fastq, 512, quantize=True, chunksize=10000, lanes=3) == 1
    assert [h for h in range(16, 529, 16) if q8_covers(h)] == [64, 96, 128, 144, 192, 256, 288, 336, 384, 512]
    assert auto_lanes(synthetic.make_model("sup_lstm", batchsize=16, chunksize=1200), True) == 1
    assert auto_lanes(synthetic.make_transformer_model(batchsize=16, chunksize=1200)) == 1
    assert [batches_per_call(hac, b) for b in (128, 256, 512, 1024, 2048, 4096)] == [8, 8, 4, 2, 1, 1]
    assert batches_per_call(hac, 512, quantize=True) == 1 and batches_per_call(hac, 512, quantize=True, lanes=2) == 4
    assert batches_per_call(hac, 256, quantize=True, lanes=2) == 4 and batches_per_call(hac, 2048, quantize=True, lanes=2) == 1
    assert [batches_per_call(hac, 512, chunksize=c) for c in (4000, 10000, 20000, 40000)] == [4, 4, 2, 1]
    fast = synthetic.make_model("fast", batchsize=16, chunksize=1200)
    assert batches_per_call(fast, 512, chunksize=10000) == 1 and batches_per_call(fast, 512, chunksize=10000, lanes=3) == 8
    assert batches_per_call(synthetic.make_model("sup_lstm", batchsize=16, chunksize=1200), 512, chunksize=10000) == 1
    # 1024-state models (round 5): calls of 512 chunks - their decode is one wave per chunk, two batches decode in the time of one -
    # bounded by 16 GiB of scores per call
    sup = synthetic.make_transformer_model(batchsize=16, chunksize=1200)
    assert [batches_per_call(sup, b, chunksize=12000) for b in (128, 256, 512)] == [2, 2, 1]
    assert batches_per_call(sup, 256, chunksize=20000) == 2 and batches_per_call(sup, 256, chunksize=80000) == 1
    assert batches_per_call(synthetic.make_model("sup_lstm", batchsize=16, chunksize=1200), 256, chunksize=20000) == 2
