"""Parity at the sizes and architectures BASELINE.json names (-m gpu).

1. every reference config (tests/golden/configs/*.toml = bonito/models/configs/*.toml, test data) runs on the HIP engine and
   matches the fp32 oracle (oracle/nn_ref.py, pinned to the reference's bonito/nn.py by tests/golden/nn_*.npz) on a small batch;
2. the four BASELINE configurations at FULL batch x chunk size: >= 4 chunks of the full batch are compared with the fp32
   oracle (scores) and the HIP decode of the HIP scores with oracle/crf_oracle.c (sequence / moves bit-exact, q < 1e-3).

The measured errors are written to gpurun_out/parity_<name>.json (copies of round 2: profiles/r02_parity.json); the bounds
below are <= 5x what was measured on MI355X.
"""
import copy
import glob
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, ROOT, assert_qstrings_agree
from bonito_amd import decode, synthetic, util
from bonito_amd.crf.basecall import fmt
from oracle import crf_ref, nn_ref

pytestmark = pytest.mark.gpu
CONFIGS = sorted(glob.glob(os.path.join(GOLDEN, "configs", "*.toml")))


def _record(name, **vals):
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "parity_%s.json" % name), "w") as fh:
        json.dump(vals, fh)


def _head_gain_(model, gain):
    """A freshly initialised CRF head never beats the blank score: scale it so that decoding emits bases."""
    from bonito_amd.nn import LinearCRFEncoder
    with torch.no_grad():
        for m in model.modules():
            if isinstance(m, LinearCRFEncoder):
                m.linear.weight.mul_(gain)


def _gpu_copy(model, batch, chunk):
    g = copy.deepcopy(model)
    g.use_koi(batchsize=batch, chunksize=chunk, quantize=False)
    return g.half().to("cuda")


def _oracle_scores(model, rows):
    """fp32 oracle scores of `rows` [n,1,L] in the engine's layout [n,T,C] (koi layout when a blank score is fixed)."""
    with torch.no_grad():
        y = nn_ref.forward(model.encoder, rows.float(), expand_blanks=False)
    enc = model.encoder
    last = [m for m in enc.modules() if type(m).__name__ == "LinearCRFEncoder"][-1]
    if last.permute is None:          # TNC out of the recurrent stacks; the transformer head permutes to TNC as well
        y = y.permute(1, 0, 2)
    elif list(last.permute) == [1, 0, 2]:
        y = y.permute(1, 0, 2)
    return y.contiguous()


# ---------------------------------------------------------------------------------------------------------------
# 1. every in-tree config of the reference, small batch
SMALL = {   # chunk samples (multiple of the stride), bound on (max, mean) |HIP - fp32 oracle| relative to max(1, range / 5):
            # <= 5x the errors measured on MI355X (gpurun_out/parity_small_*.json of round 2, quoted behind each line)
    "dna_r10.4.1@v4.0.toml": (1500, 6e-3, 9e-4),      # 1.3e-3, 1.7e-4
    "dna_r10.4.1@v4.3.toml": (1800, 1.2e-2, 1.3e-3),  # 2.5e-3, 2.6e-4
    "dna_r10.4.1@v5.0.toml": (2400, 4e-2, 6e-3),      # 8.0e-3, 1.3e-3 (0.086 / 0.0134 on a score range of 53)
    "dna_r9.4.1@v3.1.toml": (1500, 3e-2, 4.5e-3),     # 6.3e-3, 8.9e-4
    "dna_r9.4.1@v3.toml": (1500, 3e-2, 4.5e-3),       # 6.5e-3, 8.9e-4
}


@pytest.mark.parametrize("path", CONFIGS, ids=[os.path.basename(p) for p in CONFIGS])
def test_reference_config_matches_oracle_small(path):
    name = os.path.basename(path)
    cfg = util.set_config_defaults(util.load_toml(path))
    torch.manual_seed(25)
    model = util.load_symbol(cfg, "Model")(cfg).eval()
    synthetic.randomise_batchnorm_(model)
    nn_ref.round_params_to_half_(model)
    if cfg["model"]["package"] == "bonito.ctc":
        x = torch.randn(3, 1, 1200, generator=torch.Generator().manual_seed(1)).half()
        g = _gpu_copy(model, 3, 1200)
        got = g(x.cuda()).cpu().float()
        g._hip.check()
        with torch.no_grad():
            want = nn_ref.ctc_forward(model, x.float())
        assert got.shape == want.shape
        d = (got - want).abs()
        _record("small_" + name, max=d.max().item(), mean=d.mean().item())
        assert d.max().item() < 2.5e-3 and d.mean().item() < 1.3e-3, (d.max().item(), d.mean().item())      # measured 5.1e-4, 2.5e-4
        return
    L, tol_max, tol_mean = SMALL[name]
    _head_gain_(model, 8.0)
    nn_ref.round_params_to_half_(model)
    x = torch.randn(3, 1, L, generator=torch.Generator().manual_seed(1)).half()
    g = _gpu_copy(model, 3, L)
    got = g(x.cuda())
    g._hip.check()
    want = _oracle_scores(model, x)
    assert tuple(got.shape) == tuple(want.shape), (got.shape, want.shape)
    d = (got.cpu().float() - want).abs()
    rng = max(1.0, want.abs().max().item() / 5.0)
    _record("small_" + name, max=d.max().item(), mean=d.mean().item(), range=want.abs().max().item())
    assert d.max().item() < tol_max * rng and d.mean().item() < tol_mean * rng, (d.max().item(), d.mean().item(), rng)
    # decode of the engine's scores == oracle decode of the same scores
    S = 4 ** model.seqdist.state_len
    if got.shape[-1] == 4 * S:
        mv, path = decode.viterbi(got)
        om, op, _ = crf_ref.viterbi(got.cpu().numpy(), model.seqdist.state_len, blank=2.0)
    else:                              # learned blank column (dna_r9.4.1@v3: no blank_score): reference 5S layout
        N, T, C = got.shape
        tnc = got.permute(1, 0, 2).contiguous()
        mv, path = decode.viterbi_5s(tnc, model.seqdist.state_len)
        om, op, _ = crf_ref.viterbi(tnc.cpu().numpy(), model.seqdist.state_len, layout_5s=True, time_major=True)
    assert np.array_equal(path.numpy(), op) and np.array_equal(mv.numpy(), om)


# ---------------------------------------------------------------------------------------------------------------
# 2. BASELINE configurations at full size
ROWS = [0, 17, 255, -1]


def _full_size(name, model, batch, chunk, tol_max, tol_mean, rna=False):
    sl = model.seqdist.state_len
    nn_ref.round_params_to_half_(model)
    x = torch.randn(batch, 1, chunk, generator=torch.Generator().manual_seed(25)).half()
    g = _gpu_copy(model, batch, chunk)
    scores = g(x.cuda())
    g._hip.check()
    rows = [r % batch for r in ROWS]
    want = _oracle_scores(model, x[rows])
    got = scores[rows].cpu().float()
    assert got.shape == want.shape, (got.shape, want.shape)
    d = (got - want).abs()
    rng = max(1.0, want.abs().max().item() / 5.0)
    # HIP decode of the whole batch vs the C oracle on the same fp16 scores, selected rows
    seq, qs, mv, qf = decode.beam_search(scores, return_qfloat=True)
    sub = scores[rows].cpu().numpy()
    oseq, oqs, omv, oqf = crf_ref.beam_search(sub, sl)
    vm, vp = decode.viterbi(scores)
    om, op, _ = crf_ref.viterbi(sub, sl, blank=2.0)
    qd = float(np.abs(qf.numpy()[rows] - oqf).max())
    _record("full_" + name, max=d.max().item(), mean=d.mean().item(), range=want.abs().max().item(), q_max=qd,
            bases=int((omv != 0).sum()), T=int(scores.shape[1]), C=int(scores.shape[2]))
    assert np.array_equal(mv.numpy()[rows], omv) and np.array_equal(seq.numpy()[rows], oseq)
    assert np.array_equal(vp.numpy()[rows], op) and np.array_equal(vm.numpy()[rows], om)
    assert qd < 1e-3
    assert_qstrings_agree(qs.numpy()[rows], oqs, oqf)
    assert int((omv != 0).sum()) > 100            # the synthetic head does emit bases
    assert d.max().item() < tol_max * rng and d.mean().item() < tol_mean * rng, (d.max().item(), d.mean().item(), rng)
    # the strings the basecaller would write for these chunks (rna = reversed, crf/basecall.py:48-55)
    for i, r in enumerate(rows):
        res = fmt(g.stride, {"moves": mv[r], "qstring": qs[r], "sequence": seq[r]}, rna=rna)
        want_seq = decode.to_str(torch.from_numpy(oseq[i]))
        assert res["sequence"] == (want_seq[::-1] if rna else want_seq)
        assert len(res["qstring"]) == len(res["sequence"])
    return scores


def test_full_size_fast_512x10000():
    model = synthetic.make_model("fast", batchsize=512, chunksize=10000)
    sc = _full_size("fast", model, 512, 10000, 2.1e-2, 3e-3)         # measured max 4.3e-3, mean 6.0e-4
    assert sc.shape == (512, 1667, 256)


def test_full_size_hac_512x10000():
    model = synthetic.make_model("hac", batchsize=512, chunksize=10000)
    sc = _full_size("hac", model, 512, 10000, 2.4e-2, 3e-3)          # measured max 4.9e-3, mean 6.0e-4
    assert sc.shape == (512, 1667, 1024)


def test_full_size_sup_v5_transformer_256x12000():
    """Exactly the graph of bonito/models/configs/dna_r10.4.1@v5.0.toml (d=512, 8 heads, 18 layers, window 127/128, C=4096)."""
    cfg = util.set_config_defaults(util.load_toml(os.path.join(GOLDEN, "configs", "dna_r10.4.1@v5.0.toml")))
    torch.manual_seed(25)
    model = util.load_symbol(cfg, "Model")(cfg).eval()
    synthetic.randomise_batchnorm_(model)
    _head_gain_(model, 4.0)
    sc = _full_size("sup_v5", model, 256, 12000, 4.5e-2, 6e-3)       # measured 0.049 / 0.0067 on a score range of 27 (x 5 / 27)
    assert sc.shape == (256, 2000, 4096)


def test_full_size_sup_lstm_v43_256x20000_rna():
    """bonito/models/configs/dna_r10.4.1@v4.3.toml (LSTM-1024, state_len 5) at BASELINE config 5's shape, rna=True."""
    cfg = util.set_config_defaults(util.load_toml(os.path.join(GOLDEN, "configs", "dna_r10.4.1@v4.3.toml")))
    torch.manual_seed(25)
    model = util.load_symbol(cfg, "Model")(cfg).eval()
    synthetic.randomise_batchnorm_(model)
    _head_gain_(model, 24.0)
    sc = _full_size("sup_lstm_v43", model, 256, 20000, 2.5e-2, 3.2e-3, rna=True)    # measured max 5.2e-3, mean 6.5e-4
    assert sc.shape == (256, 3334, 4096)


def test_bench_call_shape_hac_2048x10000():
    """The default bench (and `basecaller` with automatic --per-call at batchsize 512 x 4... two launches of the paired recurrent
    kernel per layer) hands the engine FOUR BASELINE batches per call and decodes them through one `CRFDecoder(2048, ...)`:
    the timed call shape itself against the oracles - one chunk out of each 512-batch, plus the last chunk: scores vs the fp32
    restatement, beam sequence / moves and the Viterbi path bit-exact vs oracle/crf_oracle.c, q within 1e-3 - exactly as
    bench.py drives them (decode.CRFDecoder.submit on the engine's scores, the int8 planes on the host)."""
    batch, chunk = 2048, 10000
    model = synthetic.make_model("hac", batchsize=batch, chunksize=chunk)
    sl = model.seqdist.state_len
    nn_ref.round_params_to_half_(model)
    x = torch.randn(batch, 1, chunk, generator=torch.Generator().manual_seed(2048)).half()
    g = _gpu_copy(model, batch, chunk)
    scores = g(x.cuda())
    g._hip.check()
    assert "lstm_layer_wgx2_kernel<12,3>" in g._hip.describe() and scores.shape == (batch, 1667, 1024)
    rows = [17, 512 + 300, 1024 + 5, 1536 + 511, 2047]
    want = _oracle_scores(model, x[rows])
    d = (scores[rows].cpu().float() - want).abs()
    dec = decode.CRFDecoder(batch, scores.shape[1], scores.shape[2], torch.device("cuda", 0), mode="beam")
    planes = dec.submit(scores).result_planes()                 # [3, 2048, T] int8: sequence, qstring, moves
    seq, qs, mv = planes[0].numpy(), planes[1].numpy(), planes[2].numpy()
    vdec = decode.CRFDecoder(batch, scores.shape[1], scores.shape[2], torch.device("cuda", 0), mode="viterbi")
    vplanes = vdec.submit(scores).result_planes()
    sub = scores[rows].cpu().numpy()
    oseq, oqs, omv, oqf = crf_ref.beam_search(sub, sl)
    om, op, _ = crf_ref.viterbi(sub, sl, blank=2.0)
    _, _, _, qf = decode.beam_search(scores, return_qfloat=True)
    qd = float(np.abs(qf.numpy()[rows] - oqf).max())
    _record("bench_shape_hac_2048", max=d.max().item(), mean=d.mean().item(), q_max=qd, bases=int((omv != 0).sum()))
    assert np.array_equal(mv[rows], omv) and np.array_equal(seq[rows], oseq)
    assert np.array_equal(vplanes[1].numpy()[rows], op) and np.array_equal(vplanes[2].numpy()[rows], om)
    assert qd < 1e-3
    assert_qstrings_agree(qs[rows], oqs, oqf)
    assert int((omv != 0).sum()) > 100
    assert d.max().item() < 2.4e-2 and d.mean().item() < 3e-3, (d.max().item(), d.mean().item())      # as at 512 x 10000 (5 x measured)


def test_full_size_sup_v5_transformer_256x20000_rna():
    """SURVEY 8(d) config 5 says "run both the @v5.0.toml graph and the @v4.3.toml graph at this shape" (rna004 sup: batch 256 x chunk
    20000, beam decode, rna=True): the transformer graph of bonito/models/configs/dna_r10.4.1@v5.0.toml at 1667 tokens / 3334 CRF steps
    per chunk - the attention ring kernel, the rotary table and the four-wave GEMM's rotary / residual / SwiGLU epilogues at a token
    count that is not a multiple of anything - against nn_ref + crf_oracle.c on four chunks of the batch (review, round 3)."""
    cfg = util.set_config_defaults(util.load_toml(os.path.join(GOLDEN, "configs", "dna_r10.4.1@v5.0.toml")))
    torch.manual_seed(25)
    model = util.load_symbol(cfg, "Model")(cfg).eval()
    synthetic.randomise_batchnorm_(model)
    _head_gain_(model, 4.0)
    sc = _full_size("sup_v5_256x20000_rna", model, 256, 20000, 4.5e-2, 6e-3, rna=True)     # bounds of the 256 x 12000 test
    assert sc.shape == (256, 3334, 4096)


def test_every_row_of_a_2048_chunk_call_against_the_oracle():
    """The full-size tests compare a handful of rows with the oracle and reach the rest through "variant X is bit-identical to variant
    Y". Here EVERY chunk of a 2048-chunk engine call - all 128 rings, 64 ring pairs, every XCD position of the paired recurrent kernel -
    is compared directly: scores vs the fp32 restatement, the Viterbi path of every chunk and the beam decode of every 8th one vs
    oracle/crf_oracle.c on the engine's scores. Short chunks (1200 samples = 200 steps) keep the CPU oracle under a minute (review,
    round 3)."""
    batch, chunk = 2048, 1200
    model = synthetic.make_model("hac", batchsize=batch, chunksize=chunk)
    sl = model.seqdist.state_len
    nn_ref.round_params_to_half_(model)
    x = torch.randn(batch, 1, chunk, generator=torch.Generator().manual_seed(4)).half()
    g = _gpu_copy(model, batch, chunk)
    scores = g(x.cuda())
    g._hip.check()
    assert "lstm_layer_wgx2_kernel<12,3>" in g._hip.describe() and scores.shape == (batch, 200, 1024)
    worst_max, worst_mean, worst_row = 0.0, 0.0, -1
    for lo in range(0, batch, 256):
        want = _oracle_scores(model, x[lo:lo + 256])
        d = (scores[lo:lo + 256].cpu().float() - want).abs()
        per_row_max = d.flatten(1).max(1).values
        if per_row_max.max().item() > worst_max:
            worst_max, worst_row = per_row_max.max().item(), lo + int(per_row_max.argmax())
        worst_mean = max(worst_mean, d.flatten(1).mean(1).max().item())
    sub = scores.cpu().numpy()
    vm, vp = decode.viterbi(scores)
    om, op, _ = crf_ref.viterbi(sub, sl, blank=2.0)
    seq, qs, mv, qf = decode.beam_search(scores, return_qfloat=True)
    rows = list(range(0, batch, 8))
    oseq, oqs, omv, oqf = crf_ref.beam_search(sub[rows], sl)
    qd = float(np.abs(qf.numpy()[rows] - oqf).max())
    _record("all_rows_hac_2048x1200", max=worst_max, mean_of_worst_row=worst_mean, worst_row=worst_row, q_max=qd, bases=int((omv != 0).sum()))
    assert np.array_equal(vp.numpy(), op) and np.array_equal(vm.numpy(), om)
    assert np.array_equal(mv.numpy()[rows], omv) and np.array_equal(seq.numpy()[rows], oseq)
    assert qd < 1e-3
    assert_qstrings_agree(qs.numpy()[rows], oqs, oqf)
    assert int((omv != 0).sum()) > 100
    assert worst_max < 2.4e-2 and worst_mean < 3e-3, (worst_max, worst_mean, worst_row)      # the bounds of the 512 x 10000 test, for every row
