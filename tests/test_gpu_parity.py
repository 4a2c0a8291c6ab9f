"""END-TO-END identity of the hot path against the CPU path (-m gpu): HIP encoder -> HIP decoders vs fp32 oracle encoder -> C decoders.

The per-stage tests compare the encoder within a tolerance and the decoders bit-exactly ON IDENTICAL SCORES. This file closes the
composite (north_star: "Outputs match the reference CPU path's basecalls"; reference seam `compute_scores`,
/root/reference bonito/crf/basecall.py:27-45, and CTC_CRF.viterbi, bonito/crf/model.py:98-108): the same chunks go through both whole
paths, and the Viterbi path, the beam sequence and the move table are compared entry by entry and by alignment (oracle/parity.py).
With scores that differ by a few 1e-3 (fp16 arithmetic against fp32) near-ties of the recursion do flip; the floors below are the
identity measured on MI355X with at most twice its error rate allowed (figures behind each line; every run rewrites
gpurun_out/parity_e2e_<name>.json, the bench line carries the same object as `parity`).

The oracle's chunks ride at the head of a FULL engine call (512 / 256 chunks), so the kernels compared are the ones the bench times.

Round 6: 64 chunks for fast / hac (8 before); every comparison is made against TWO CPU oracles - the fp32 path and nn_ref's fp16-storage mode
(values rounded where the engine stores fp16) - so that precision is told apart from summation order; BASELINE configs 5 (256 x 20000, both
candidate graphs) and 1 (CTC, 16 x 4000) have their own rows; the BS-2-against-BS-1 quality guard runs on every sample.
"""
import copy
import json
import os

import numpy as np
import pytest
import torch

from conftest import ROOT
from bonito_amd import synthetic
from oracle import nn_ref, parity

pytestmark = pytest.mark.gpu


def _record(name, res):
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "parity_e2e_%s.json" % name), "w") as fh:
        json.dump(res, fh, indent=1)


def _run(name, model, batch, chunk, n, quantize=False):
    """-> (HIP vs the fp32 CPU path, HIP vs the fp16-storage oracle, the two oracles against each other) on the same n chunks."""
    nn_ref.round_params_to_half_(model)
    xo = torch.randn(n, 1, chunk, generator=torch.Generator().manual_seed(25)).half()
    threads = min(16, n)
    ora = parity.oracle_outputs(model, xo, threads=threads)
    ora16 = parity.oracle_outputs(model, xo, threads=threads, fp16=True)
    g = copy.deepcopy(model)
    g.use_koi(batchsize=batch, chunksize=chunk, quantize=quantize)
    g = g.half().to("cuda")
    x = torch.randn(batch, 1, chunk, generator=torch.Generator().manual_seed(7)).half()
    x[:n] = xo
    hip = parity.hip_outputs(g, x.cuda(), n)
    res = parity.compare(hip, ora)
    res16 = parity.compare(hip, ora16)
    between = parity.compare(ora16, ora)
    guard = parity.bs2_vs_bs1(ora["scores"], int(ora["state_len"]), ora["beam_seq"], threads=threads)
    _record(name, dict(res, vs_fp16_storage_oracle=res16, fp16_storage_oracle_vs_fp32_oracle=between, bs2_vs_bs1=guard))
    assert res["beam_alignment_columns"] > 100 * n and res["viterbi_alignment_columns"] > 100 * n      # the synthetic head does emit bases
    # the decoder's redefinition (BS-1 -> BS-2) costs no sequence probability on these scores (exact fp64 path sums)
    assert guard["dlogp_mean"] > -0.05 and guard["dlogp_min"] > -1.0, guard
    return res, res16, between


def _floors(res, **floors):
    bad = {k: (res[k], v) for k, v in floors.items() if res[k] < v}
    assert not bad, "identity below its floor (measured, floor): %r -- all figures: %r" % (bad, res)


def _precision_not_order(res, res16, between):
    """What separates the engine from the fp32 CPU path is fp16 STORAGE, not its arithmetic: the fp16-storage oracle (a CPU program) sits as
    far from the fp32 path as the engine does, and the engine is CLOSER to it than to the fp32 path."""
    assert res16["scores_mean_abs"] < res["scores_mean_abs"], (res16["scores_mean_abs"], res["scores_mean_abs"])
    assert between["scores_mean_abs"] > 0.5 * res["scores_mean_abs"], (between["scores_mean_abs"], res["scores_mean_abs"])


# floors: 1 - 2 x (1 - identity measured on MI355X), figures of the first measured run behind each line
def test_end_to_end_identity_fast_512x10000():
    res, res16, between = _run("fast", synthetic.make_model("fast", batchsize=512, chunksize=10000), 512, 10000, 64)
    # measured (round 5, 8 chunks): Viterbi path 1.0 (8 of 8 chunks bit-identical), beam sequence 1.0, beam MOVE TABLE 0.9507 - the sequences agree
    # base for base, the step at which a base is emitted does not always: the search folds a move into the stay that spells the same sequence and
    # follows the higher of the two, and a random-weight head leaves those two within the fp16 error of each other
    _floors(res, viterbi_path_identity=0.998, viterbi_seq_identity=0.998, beam_seq_identity=0.998, moves_identity=0.90)
    assert res["scores_max_abs"] < 2.1e-2
    _precision_not_order(res, res16, between)


def test_end_to_end_identity_hac_512x10000():
    res, res16, between = _run("hac", synthetic.make_model("hac", batchsize=512, chunksize=10000), 512, 10000, 64)
    # measured (round 5, 8 chunks; the bench line's `parity`): Viterbi path 1.0, beam sequence 1.0, beam move table 0.9613
    _floors(res, viterbi_path_identity=0.998, viterbi_seq_identity=0.998, beam_seq_identity=0.998, moves_identity=0.92)
    assert res["scores_max_abs"] < 2.4e-2
    _precision_not_order(res, res16, between)


def test_end_to_end_identity_hac_quantize_512x10000():
    """The 8-bit recurrence (Q8-1) against the fp32 CPU path: a different arithmetic, so a lower identity - stated, not hidden."""
    res, _, _ = _run("hac_q8", synthetic.make_model("hac", batchsize=512, chunksize=10000), 512, 10000, 32, quantize=True)
    # measured (round 5, 8 chunks): scores max |d| 0.148 - and still Viterbi path 1.0, beam sequence 1.0; beam move table 0.831, 43 % of the bases
    # emitted at the same step (the synthetic head saturates its tanh * 5 scores: WHICH base is robust, WHEN it is emitted is not)
    # round 6, 32 chunks: 29-31 of 32 Viterbi paths bit-identical; a chunk that differs is the same sequence emitted a step apart over a stretch
    # (path identity 0.937-0.979 entry by entry, sequence identity 0.9997-0.9999)
    _floors(res, viterbi_path_identity=0.87, viterbi_seq_identity=0.999, beam_seq_identity=0.999, moves_identity=0.66)


def test_end_to_end_identity_sup_v5_transformer_256x12000():
    model = synthetic.make_transformer_model(head_gain=4.0, batchsize=256, chunksize=12000)
    res, _, _ = _run("sup_v5", model, 256, 12000, 4)
    # measured (round 5, 2 chunks): everything 1.0 (both chunks bit-identical in path, sequence and moves) at a score error of 0.047 on a range of 27
    _floors(res, viterbi_path_identity=0.995, viterbi_seq_identity=0.995, beam_seq_identity=0.995, moves_identity=0.99)


# BASELINE config 5 (rna004 sup, 256 x 20000, 1024 states; architecture unknown offline -> BOTH candidate graphs). `rna=True` only
# reverses the called strings on the host (crf/basecall.py `fmt`, reference bonito/crf/basecall.py:48-55): test_gpu_configs.py holds that.
def test_end_to_end_identity_config5_transformer_graph_256x20000():
    model = synthetic.make_transformer_model(head_gain=4.0, batchsize=256, chunksize=20000)
    res, _, _ = _run("config5_v5_graph", model, 256, 20000, 2)
    # measured (round 6, 4 chunks in the bench line): Viterbi path 0.99498 (3 of 4 chunks bit-identical), sequence 0.9994, beam sequence 0.9980,
    # move table 0.9995 against the fp32 path; 4 of 4 identical against the fp16-storage oracle
    _floors(res, viterbi_path_identity=0.98, viterbi_seq_identity=0.997, beam_seq_identity=0.995, moves_identity=0.98)


def test_end_to_end_identity_config5_lstm1024_graph_256x20000():
    res, res16, between = _run("config5_v43_graph", synthetic.make_model("sup_lstm", batchsize=256, chunksize=20000), 256, 20000, 2)
    # measured (round 6): paths and sequences identical, beam move table 0.844 (2 chunks) / 0.878 (4 chunks, the bench line): with 1024 states
    # and a random-weight head many alignments of the same sequence score within the fp16 error of each other
    _floors(res, viterbi_path_identity=0.995, viterbi_seq_identity=0.995, beam_seq_identity=0.995, moves_identity=0.72)
    # (no `_precision_not_order` here: at 1024 hidden units the accumulation order of a 2048-long dot product weighs as much as the fp16
    # storage - engine vs fp32 path 5.4e-4 mean, engine vs fp16-storage oracle 5.6e-4, the two oracles 5.1e-4 apart: all the same size)
    assert res16["scores_mean_abs"] < 2 * res["scores_mean_abs"] and between["scores_mean_abs"] > 0.3 * res["scores_mean_abs"]


def test_end_to_end_identity_config1_ctc_16x4000():
    """BASELINE config 1 (dna_r9.4.1 QuartzNet CTC, 16 chunks x 4000, greedy + prefix beam 5) through the product pipeline against the fp32
    CPU path: bench.py's `config1_ctc` leg, asserted."""
    import bench
    r = bench.config1_worker()
    _record("config1_ctc", r)
    assert r["greedy_alignment_columns"] > 16 * 50 and r["beam5_alignment_columns"] > 16 * 50
    # measured (round 6): greedy 0.9978 over 13280 alignment columns, prefix beam 0.9762 over 9346 (the seeded head's best label has probability
    # ~ 0.25: a flat posterior, every near-tie is a coin flip between the fp16 engine and the fp32 CPU path); floors at twice the error rate
    assert r["greedy_seq_identity"] > 0.995 and r["beam5_seq_identity"] > 0.95, r
    assert r["logp_max_abs"] < 5e-2, r


def test_identity_metrics_are_one_for_the_oracle_against_itself_and_drop_for_a_planted_error():
    """The checker itself, on real decoder outputs of the device: identical planes -> every fraction 1.0; one substituted base and one
    deleted base in a chunk -> the alignment identity drops by exactly those two columns."""
    model = synthetic.make_model("fast", batchsize=16, chunksize=3000)
    nn_ref.round_params_to_half_(model)
    g = copy.deepcopy(model)
    g.use_koi(batchsize=16, chunksize=3000, quantize=False)
    g = g.half().to("cuda")
    x = torch.randn(16, 1, 3000, generator=torch.Generator().manual_seed(3)).half()
    hip = parity.hip_outputs(g, x.cuda(), 4)
    same = parity.compare(hip, hip)
    for k in ("viterbi_path_identity", "viterbi_seq_identity", "beam_seq_identity", "moves_identity", "bases_matching_in_place"):
        assert same[k] == 1.0
    assert same["q_max_abs_on_matching_bases"] == 0.0 and same["beam_chunks_bit_identical"] == 4
    bad = {k: np.array(v, copy=True) for k, v in hip.items() if isinstance(v, np.ndarray)}
    pos = np.flatnonzero(bad["beam_seq"][0])
    assert len(pos) > 20
    a, b = pos[5], pos[11]
    bad["beam_seq"][0, a] = 65 + (bad["beam_seq"][0, a] - 65 + 2) % 20          # another letter
    bad["beam_seq"][0, b] = 0
    bad["beam_moves"][0, b] = 0
    res = parity.compare(bad, hip)
    cols = res["beam_alignment_columns"]
    assert cols == same["beam_alignment_columns"] and abs(res["beam_seq_identity"] - (cols - 2) / cols) < 1e-12
    assert res["beam_chunks_bit_identical"] == 3
