"""bonito.ctc (QuartzNet) surface on CPU: state_dict contract, oracle vs the reference fixture, greedy decode oracle."""
import numpy as np
import torch

from conftest import build_ctc_model, load_ctc_fixture
from oracle import ctc_ref, nn_ref


def test_ctc_state_dict_matches_reference_and_oracle_matches_fixture():
    cfg, sd, x, y = load_ctc_fixture()
    model = build_ctc_model(cfg, sd)
    assert list(model.state_dict().keys()) == list(sd.keys())
    assert model.stride == 3 and model.alphabet == ["N", "A", "C", "G", "T"]
    with torch.no_grad():
        got = nn_ref.ctc_forward(model, x)
    assert got.shape == y.shape == (200, 3, 5)
    assert (got - y).abs().max().item() < 2e-4
    assert (got.exp().sum(-1) - 1).abs().max().item() < 1e-5


def test_ctc_seeded_init_matches_reference_init():
    """Same seed -> same weights as the reference constructors (checked through one fixture-independent stat)."""
    from bonito_amd.ctc import Model
    cfg, sd, _, _ = load_ctc_fixture()
    torch.manual_seed(25)
    mine = Model(cfg).state_dict()
    # the fixture's BN stats were randomised and weights rounded to fp16 after seeding: conv weights must agree to fp16 rounding
    k = "encoder.encoder.1.conv.0.pointwise.weight"
    assert (mine[k].half().float() - sd[k]).abs().max().item() == 0.0


def test_greedy_oracle_semantics():
    alphabet = ["N", "A", "C", "G", "T"]
    #            t: 0    1    2    3    4    5    6
    labels = [1, 1, 0, 1, 2, 2, 0]
    lp = np.full((7, 5), np.log(0.025), np.float32)
    for t, l in enumerate(labels):
        lp[t, l] = np.log(0.9)
    seq, qs, path = ctc_ref.viterbi_search(lp, alphabet)
    assert seq == "AAC" and path == [0, 3, 4]
    assert qs == chr(33 + 10) * 3                      # p = 0.9 -> Q10
    seq2, qs2, _ = ctc_ref.viterbi_search(lp, alphabet, qscale=2.0, qbias=1.0)
    assert qs2 == chr(33 + 21) * 3
    assert ctc_ref.viterbi_search(np.log(np.full((4, 5), 0.2, np.float32)), alphabet) == ("", "", [])   # ties -> blank
    assert ctc_ref.phred(1.0) == 33 + 40 and ctc_ref.phred(0.0) == 33


def test_prefix_beam_oracle_equals_exhaustive_map_on_tiny_inputs():
    import itertools
    from collections import defaultdict
    rng = np.random.default_rng(1)
    alphabet = ["N", "A", "C", "G", "T"]
    for _ in range(12):
        T = int(rng.integers(1, 6))
        lp = torch.log_softmax(torch.from_numpy(rng.standard_normal((T, 5)).astype(np.float32) * 2), -1).numpy()
        tot = defaultdict(float)
        for al in itertools.product(range(5), repeat=T):
            p = float(np.exp(sum(lp[t, a] for t, a in enumerate(al))))
            seq, prev = [], 0
            for a in al:
                if a != 0 and a != prev:
                    seq.append(a)
                prev = a
            tot[tuple(seq)] += p
        best = max(tot.items(), key=lambda kv: kv[1])[0]
        seq, path = ctc_ref.beam_search(lp, alphabet, beam_size=16, beam_cut_threshold=1e-9)
        assert seq == "".join(alphabet[i] for i in best)
        assert len(path) == len(seq)
    # beam of 1 with no threshold never beats the wide beam
    lp = torch.log_softmax(torch.from_numpy(rng.standard_normal((200, 5)).astype(np.float32) * 3), -1).numpy()
    assert len(ctc_ref.beam_search(lp, alphabet, 1, 1e-3)[0]) > 0
