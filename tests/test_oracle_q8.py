"""Q8-1, the definition of an 8-bit recurrent path (oracle/lstm_q8_ref.py): what it computes and what it costs in accuracy
against the fp32 oracle. CPU only; there is no int8 kernel yet (quantize=True runs the fp16 kernels) - this pins the
semantics such a kernel will be tested against. koi's own int8 LSTM is closed source: parity unpinned."""
import numpy as np
import pytest
import torch

from conftest import NN_FIXTURES, build_model, load_nn_fixture
from oracle import crf_ref, lstm_q8_ref, nn_ref


def test_row_quantisation_properties():
    rng = np.random.default_rng(0)
    W = (rng.standard_normal((40, 96)) * rng.uniform(0.01, 2.0, (40, 1))).astype(np.float32)
    W[7] = 0.0
    q, s = lstm_q8_ref.quantise_rows(W)
    assert q.dtype == np.int8 and np.abs(q.astype(np.int32)).max() == 127 and s[7] == 1.0 and not q[7].any()
    assert (np.abs(q.astype(np.float32) * s[:, None] - W) <= s[:, None] * 0.5 + 1e-7).all()
    rows = np.delete(np.arange(40), 7)
    assert (np.abs(q[rows].astype(np.int32)).max(axis=1) == 127).all()           # every non-zero row uses the full range
    a = lstm_q8_ref.quantise_act(np.array([0.0, 0.5, -1.0, 3.0, 0.0039, 2.5 / 127, 3.5 / 127]), 1.0)
    assert a.tolist() == [0, 64, -127, 127, 0, 2, 4]                              # round half to even, saturating


@pytest.mark.parametrize("name", NN_FIXTURES)
def test_q8_scores_stay_close_to_fp32_on_reference_fixtures(name):
    cfg, sd, x, _ = load_nn_fixture(name)
    model = build_model(cfg, sd)
    with torch.no_grad():
        ref = nn_ref.forward(model, x)
        q8 = lstm_q8_ref.forward_q8(model, x)
    d = (q8 - ref).abs()
    assert d.max().item() < 0.05 and d.mean().item() < 0.01          # measured: max 0.018, mean 0.004 (worst fixture)


def test_q8_decodes_like_fp32_on_a_fast_shaped_model():
    from bonito_amd import synthetic
    model = synthetic.make_model("fast", batchsize=4, chunksize=3000)
    nn_ref.round_params_to_half_(model)
    x = torch.randn(4, 1, 3000, generator=torch.Generator().manual_seed(3)).half().float()
    with torch.no_grad():
        ref = nn_ref.forward(model.encoder, x, expand_blanks=False)
        q8 = lstm_q8_ref.forward_q8(model.encoder, x, expand_blanks=False)
    d = (q8 - ref).abs()
    assert d.max().item() < 0.4 and d.mean().item() < 0.05           # measured 0.12 / 0.02 on scores in [-5, 5]
    sl = model.seqdist.state_len
    paths = [crf_ref.viterbi(s.permute(1, 0, 2).contiguous().numpy().astype(np.float16), sl, layout_5s=False, blank=2.0)[1]
             for s in (ref, q8)]
    assert (paths[0] == paths[1]).mean() > 0.99                       # measured 1.0
