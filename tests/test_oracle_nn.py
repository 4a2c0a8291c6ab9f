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
