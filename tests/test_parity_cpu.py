"""CPU checks of the end-to-end identity checker (oracle/parity.py) and of bench.py's CPU leg that feeds it (no GPU)."""
import numpy as np
import torch

from bonito_amd import synthetic
from oracle import crf_ref, nn_ref, parity


def test_alignment_identity_counts_matches_over_alignment_columns():
    assert parity.alignment_identity(b"ACGTACGT", b"ACGTACGT") == (8, 8)
    assert parity.alignment_identity(b"ACGTACGT", b"ACGAACGT") == (7, 8)            # substitution
    assert parity.alignment_identity(b"ACGTACGT", b"ACGACGT") == (7, 8)             # deletion
    assert parity.alignment_identity(b"ACGTACGT", b"ACGTTACGT") == (8, 9)           # insertion
    assert parity.alignment_identity(b"", b"") == (0, 0)
    assert parity.alignment_identity(np.array([65, 67], np.int8), np.array([65, 67], np.int8)) == (2, 2)


def test_oracle_outputs_and_compare_on_a_small_model():
    model = synthetic.make_model("fast", batchsize=4, chunksize=1200)
    nn_ref.round_params_to_half_(model)
    x = torch.randn(3, 1, 1200, generator=torch.Generator().manual_seed(2)).half()
    tm = {}
    ora = parity.oracle_outputs(model, x, timers=tm)
    assert ora["scores"].dtype == np.float16 and ora["scores"].shape == (3, 200, 256)
    assert set(tm) == {"forward", "viterbi", "beam"} and all(v >= 0 for v in tm.values())
    # the planes are what the oracle's decoders say on those scores
    mv, path, _ = crf_ref.viterbi(ora["scores"], 3, blank=2.0)
    assert np.array_equal(path, ora["vit_path"]) and np.array_equal(mv, ora["vit_moves"])
    same = parity.compare(ora, ora)
    assert same["viterbi_path_identity"] == 1.0 and same["beam_seq_identity"] == 1.0 and same["moves_identity"] == 1.0
    assert same["scores_max_abs"] == 0.0 and same["beam_chunks_bit_identical"] == 3 and same["viterbi_chunks_bit_identical"] == 3
    # scores perturbed at the level of fp16 arithmetic: the figures stay high but the comparison is not vacuous
    noisy = dict(ora)
    rng = np.random.default_rng(0)
    sc = (ora["scores"].astype(np.float32) + rng.normal(0, 1.5, ora["scores"].shape)).astype(np.float16)
    noisy["scores"] = sc
    mv, path, _ = crf_ref.viterbi(sc, 3, blank=2.0)
    noisy.update(vit_moves=mv, vit_path=path)
    seq, qs, bmv, qf = crf_ref.beam_search(sc, 3)
    noisy.update(beam_seq=seq, beam_qs=qs, beam_moves=bmv, beam_qf=qf)
    res = parity.compare(noisy, ora)
    assert 0.2 < res["viterbi_seq_identity"] < 1.0 and 0.2 < res["beam_seq_identity"] < 1.0
    assert res["scores_max_abs"] > 0.5


def test_bench_cpu_leg_keeps_the_oracle_outputs_for_the_parity_leg(tmp_path):
    import bench
    keep = str(tmp_path / "o.npz")
    r = bench.cpu_baseline_worker("fast", 1200, "beam", seconds_budget=0.01, keep=keep)
    assert r["kind"] in ("reference", "port") and r["value"] > 0 and r["cores"] >= 1
    if r["kind"] == "reference":                       # this container holds /root/reference: the live agreement is reported
        assert r["oracle_vs_reference_max_abs"] < 1e-4
    d = dict(np.load(keep))
    assert d["scores"].shape == (8, 200, 256) and d["beam_seq"].shape == (8, 200) and d["vit_path"].shape == (8, 200)
