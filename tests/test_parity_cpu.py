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
    import json

    import bench
    keep = str(tmp_path / "o.npz")
    r = bench.cpu_baseline_worker("fast", 1200, "beam", n=8, keep=keep)
    # the bench never executes the reference's code (advisor, round 5): the CPU leg is the restatement, labelled as such
    assert r["kind"] == "port" and r["value"] > 0 and r["cores"] >= 1 and "8 chunks x 1200" in r["sample"]
    d = dict(np.load(keep))
    assert d["scores"].shape == (8, 200, 256) and d["beam_seq"].shape == (8, 200) and d["vit_path"].shape == (8, 200)
    # ... and (untimed) the fp16-storage oracle's outputs on the same chunks and the BS-2-against-BS-1 guard
    assert d["h_scores"].shape == (8, 200, 256) and d["h_beam_seq"].shape == (8, 200)
    assert 0 < np.abs(d["h_scores"].astype(np.float32) - d["scores"].astype(np.float32)).max() < 2e-2
    g = json.loads(bytes(d["bs2_vs_bs1_json"]).decode())
    assert g["chunks"] == 8 and g["dlogp_min"] > -0.5


def test_fp16_storage_oracle_rounds_where_the_engine_stores_fp16():
    """nn_ref's second mode: same graph, values rounded to fp16 at the engine's storage points. Its outputs are fp16-representable at every
    such point, it stays within fp16 resolution of the fp32 path, and it is NOT the fp32 path (the comparison is not vacuous)."""
    model = synthetic.make_model("fast", batchsize=4, chunksize=1800)
    nn_ref.round_params_to_half_(model)
    x = torch.randn(2, 1, 1800, generator=torch.Generator().manual_seed(5)).half().float()
    layers = list(model.encoder.children())
    with torch.no_grad():
        h32 = h16 = x
        for i, layer in enumerate(layers[:6]):            # conv x3, permute, two recurrent layers
            h32 = nn_ref.forward(layer, h32)
            h16 = nn_ref.forward(layer, h16, fp16=True)
            assert torch.equal(h16, h16.half().float()), "layer %d output is not fp16-representable" % i
        assert 0 < (h32 - h16).abs().max() < 4e-3
        a = nn_ref.forward(model.encoder, x, expand_blanks=False)
        b = nn_ref.forward(model.encoder, x, expand_blanks=False, fp16=True)
    assert 0 < (a - b).abs().max() < 2e-2


def test_bs2_quality_guard_against_bs1_on_hac_shaped_scores():
    """The product decoder's definition (BS-2) against the decoder of rounds 1-4 (BS-1) at the hac shape (256 states, T = 1667), scored by
    the model's exact fp64 sequence log-probability: on sharp scores the two call the same sequence; on flat scores (where a beam of 32
    actually prunes) BS-2 must not be worse on average and never by more than a fraction of a nat per chunk."""
    rng = np.random.default_rng(11)
    for scale, n in ((2.0, 3), (0.7, 3)):
        sc = np.clip(rng.normal(0, scale, (n, 1667, 1024)), -5, 5).astype(np.float16)
        seq = crf_ref.beam_search(sc, 4)[0]
        g = parity.bs2_vs_bs1(sc, 4, seq, threads=3)
        assert g["chunks"] == n and g["dlogp_mean"] > -0.05 and g["dlogp_min"] > -1.0, g
    # the guard is not vacuous: one substituted base in the product decoder's answer costs log-probability, and it shows
    bad = seq.copy()
    pos = np.flatnonzero(bad[0])[40]
    bad[0, pos] = ord("ACGT"[("ACGT".index(chr(bad[0, pos])) + 1) % 4])
    g = parity.bs2_vs_bs1(sc, 4, bad, threads=3)
    assert g["bs2_worse"] == 1 and g["dlogp_min"] < -0.1 and g["sequences_identical"] == n - 1, g
