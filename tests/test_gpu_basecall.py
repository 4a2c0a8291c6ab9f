"""End-to-end basecall pipeline on the GPU (chunk -> batch -> HIP encoder -> HIP beam search -> stitch)
against an independent serial restatement that decodes the same HIP scores with the CPU oracle (-m gpu)."""
import numpy as np
import pytest
import torch

from conftest import load_nn_fixture
from bonito_amd import decode, util
from bonito_amd.crf import basecall as crf_basecall_fn
from bonito_amd.crf.basecall import stitch_results, fmt
from bonito_amd.crf.model import Model
from oracle import crf_ref

pytestmark = pytest.mark.gpu


class Read:
    def __init__(self, read_id, signal):
        self.read_id, self.signal = read_id, signal


def _model(chunksize, batchsize):
    cfg, sd, _, _ = load_nn_fixture("lstm64_sl3")
    config = {"model": {"package": "bonito.crf"}, "labels": {"labels": ["N", "A", "C", "G", "T"]},
              "input": {"features": 1}, "global_norm": {"state_len": 3}, "encoder": cfg,
              "basecaller": {"chunksize": chunksize, "overlap": 96, "batchsize": batchsize}}
    model = Model(config)
    model.use_koi(batchsize=batchsize, chunksize=chunksize, quantize=False)
    sd = {"encoder." + k: (v * 30.0 if k.endswith("linear.weight") else v) for k, v in sd.items()}   # make it emit bases
    model.load_state_dict(sd)
    return model.half().eval().to("cuda")


def _reads(rng, lengths):
    return [Read("read_%d" % i, rng.standard_normal(n).astype(np.float32)) for i, n in enumerate(lengths)]


@pytest.mark.parametrize("decoder", ["beam", "viterbi"])
def test_basecall_pipeline_matches_serial_oracle_decode(decoder):
    chunksize, overlap, batchsize = 996, 96, 7
    model = _model(chunksize, batchsize)
    rng = np.random.default_rng(11)
    reads = _reads(rng, [5003, 996, 700, 2500, 12345, 997, 100])     # ragged: stub chunks, short reads, exact fits
    got = list(crf_basecall_fn(model, iter(reads), chunksize=chunksize, overlap=overlap, batchsize=batchsize,
                               decoder=decoder))
    assert [r.read_id for r, _ in got] == [r.read_id for r in reads]          # order preserved
    for read, res in got:
        ch = util.chunk(torch.from_numpy(read.signal), chunksize, overlap)
        scores = model(ch.half().cuda())
        if decoder == "beam":
            seq, qs, mv, _ = crf_ref.beam_search(scores.cpu().numpy(), 3)
        else:
            mv, path, _ = crf_ref.viterbi(scores.cpu().numpy(), 3)
            seq = decode.path_to_sequence(path).numpy()
            qs = np.where(seq != 0, 33 + 20, 0).astype(np.int8)
        attrs = {"sequence": torch.from_numpy(seq), "qstring": torch.from_numpy(qs), "moves": torch.from_numpy(mv)}
        want = fmt(model.stride, stitch_results(attrs, len(read.signal), chunksize, overlap, model.stride))
        assert res["sequence"] == want["sequence"], read.read_id
        assert np.array_equal(res["moves"], want["moves"])
        if decoder == "beam":
            diff = sum(a != b for a, b in zip(res["qstring"], want["qstring"]))
            assert len(res["qstring"]) == len(want["qstring"]) and diff <= max(1, len(want["qstring"]) // 500)
        assert res["stride"] == 6
        assert len(res["sequence"]) == len(res["qstring"]) == int(res["moves"].sum())
        assert len(res["moves"]) in (len(read.signal) // 6, (len(read.signal) + 5) // 6)


def test_basecall_rna_flips_and_empty_input():
    model = _model(996, 4)
    assert list(crf_basecall_fn(model, iter([]), chunksize=996, overlap=96, batchsize=4)) == []
    rng = np.random.default_rng(2)
    reads = _reads(rng, [3000])
    (r, dna), = list(crf_basecall_fn(model, iter(reads), chunksize=996, overlap=96, batchsize=4))
    (r2, rna), = list(crf_basecall_fn(model, iter(reads), chunksize=996, overlap=96, batchsize=4, rna=True))
    assert rna["sequence"] == dna["sequence"][::-1] and rna["qstring"] == dna["qstring"][::-1]


def test_decoder_context_reuse_and_ragged_batch():
    """CRFDecoder handles a short final batch and repeated submission; equals the allocating entry point."""
    rng = np.random.default_rng(3)
    sc = np.clip(rng.standard_normal((9, 64, 256)) * 2.5, -5, 5).astype(np.float16)
    dec = decode.CRFDecoder(16, 64, 256, "cuda:0", mode="beam")
    a = dec.submit(torch.from_numpy(sc).cuda()).result()
    b = dec.submit(torch.from_numpy(sc[:4]).cuda()).result()
    ref = decode.beam_search(torch.from_numpy(sc).cuda())
    for x, y, z in zip(a, b, ref):
        assert torch.equal(x, z) and torch.equal(y, z[:4])
    with pytest.raises(ValueError):
        dec.submit(torch.zeros(17, 64, 256, dtype=torch.float16, device="cuda"))
