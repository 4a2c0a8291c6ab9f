"""Oracle (oracle/nn_ref.py) vs the committed outputs of the reference's bonito/nn.py (CPU, no GPU)."""
import pytest
import torch

from conftest import NN_FIXTURES, build_model, load_nn_fixture
from oracle import nn_ref


@pytest.mark.parametrize("name", NN_FIXTURES)
def test_oracle_matches_reference_fixture(name):
    cfg, sd, x, y = load_nn_fixture(name)
    model = build_model(cfg, sd)          # bonito_amd containers hold the reference's weights
    with torch.no_grad():
        got = nn_ref.forward(model, x)
    assert got.shape == y.shape
    assert (got - y).abs().max().item() < 2e-4


@pytest.mark.parametrize("name", NN_FIXTURES)
def test_state_dict_keys_match_reference(name):
    """bonito_amd.nn builds the same parameter names/shapes as the reference constructors."""
    cfg, sd, _, _ = load_nn_fixture(name)
    from bonito_amd import nn as bnn
    model = bnn.from_dict(cfg)
    mine = model.state_dict()
    assert list(mine.keys()) == list(sd.keys())
    for k in sd:
        assert tuple(mine[k].shape) == tuple(sd[k].shape), k


def test_containers_refuse_to_compute():
    cfg, sd, x, _ = load_nn_fixture("lstm32_sl2")
    model = build_model(cfg, sd)
    from bonito_amd.nn import NoTorchCompute
    with pytest.raises(NoTorchCompute):
        model(x)


def test_fuse_bn_matches_unfused():
    from bonito_amd import nn as bnn
    cfg, sd, x, y = load_nn_fixture("lstm32_sl2")
    model = build_model(cfg, sd)
    model.apply(bnn.fuse_bn_)
    assert all(m.norm is None for m in model.modules() if isinstance(m, bnn.Convolution))
    with torch.no_grad():
        got = nn_ref.forward(model, x)
    assert (got - y).abs().max().item() < 5e-4


# ---- transformer (v5 sup architecture) ------------------------------------------------------------
from conftest import TF_FIXTURES, build_tf_model, load_tf_fixture  # noqa: E402


@pytest.mark.parametrize("name", TF_FIXTURES)
def test_oracle_matches_reference_transformer_fixture(name):
    """in-tree transformer/model.py (SDPA + sliding_window_mask branch) executed at fixture time; the
    flash-attn pieces are [EXT] stubs (tests/golden/make_golden.py::ref_transformer) -> parity unpinned there."""
    cfg, sd, x, y = load_tf_fixture(name)
    model = build_tf_model(cfg, sd)
    assert list(model.state_dict().keys()) == list(sd.keys())
    with torch.no_grad():
        got = nn_ref.forward(model.encoder, x)
    assert got.shape == y.shape and (got - y).abs().max().item() < 1e-3


def test_sup_v5_architecture_builds_with_reference_shapes():
    """Shape contract of the v5 sup checkpoint: 182 tensors, stride 6."""
    from bonito_amd import synthetic, util
    from bonito_amd.transformer import Model
    model = synthetic.make_transformer_model()
    sd = model.state_dict()
    assert model.stride == 6 and len(sd) == 182
    assert tuple(sd["encoder.transformer_encoder.0.self_attn.Wqkv.weight"].shape) == (1536, 512)
    assert tuple(sd["encoder.transformer_encoder.17.ff.fc1.weight"].shape) == (4096, 512)
    assert tuple(sd["encoder.transformer_encoder.17.ff.fc2.weight"].shape) == (512, 2048)
    assert tuple(sd["encoder.upsample.linear.weight"].shape) == (1024, 512)
    assert tuple(sd["encoder.crf.linear.weight"].shape) == (4096, 512)
    assert abs(float(sd["encoder.transformer_encoder.3.deepnorm_alpha"]) - 2.4494897) < 1e-6
    assert util.load_symbol({"model": {"package": "bonito.transformer"}}, "Model") is Model
    fl = synthetic.transformer_flops_per_chunk()
    assert abs(fl["total"] / 1.722e11 - 1) < 0.02          # SURVEY.md 8d figure
