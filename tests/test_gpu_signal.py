"""Device signal ingest (csrc/signal.hip) vs the host reader (bonito_amd/reader.py, pinned on reference fixtures): same
shift / scale / trim per read and the same fp16 chunk rows, bit for bit (-m gpu)."""
import numpy as np
import pytest
import torch

from bonito_amd import reader, signal
from bonito_amd.util import chunk

pytestmark = pytest.mark.gpu


def _raw_read(rng, n, peak):
    """int16 ADC samples shaped like a nanopore read: an open-pore / adapter stretch, then the read proper."""
    x = rng.normal(480, 60, n)
    if peak and n > 600:
        a = int(rng.integers(20, 200))
        b = a + int(rng.integers(80, 400))
        x[a:b] += rng.normal(420, 30, b - a)
    return np.clip(np.round(x), -32768, 32767).astype(np.int16)


def _cases():
    rng = np.random.default_rng(17)
    raws = [_raw_read(rng, int(n), peak) for n, peak in
            [(50000, True), (12345, True), (9000, False), (3000, True), (700, True), (45, False), (100000, True), (4001, False)]]
    raws.append(np.full(5000, 517, np.int16))                            # constant: scale falls back to the literal 1.0
    raws.append((np.arange(6000) % 7).astype(np.int16))                  # tiny values: shift falls back to the literal 10
    scal = [0.1755, 0.2, 0.15, 0.1755, 0.18, 0.1755, 0.21, 0.1755, 0.1755, 0.05]
    offs = [-243.0, 10.0, -200.0, 3.0, 0.0, -243.0, -100.0, 12.0, -243.0, 0.0]
    return raws, scal, offs


def test_device_normalisation_and_trim_equal_reader():
    raws, scal, offs = _cases()
    batch = signal.RawBatch(raws, scal, offs)
    shift, scale, trim = batch.normalise()
    for i, (raw, sc, of) in enumerate(zip(raws, scal, offs)):
        rd = reader.Read("r%d" % i, raw, scaling=sc, offset=of)
        assert float(rd.shift) == shift[i], (i, rd.shift, shift[i])
        assert float(rd.scale) == scale[i], (i, rd.scale, scale[i])
        assert rd.trimmed_samples == trim[i], (i, rd.trimmed_samples, trim[i])


@pytest.mark.parametrize("chunksize,overlap", [(4000, 500), (996, 498)])
def test_device_chunks_equal_reader_chunk_cast(chunksize, overlap):
    raws, scal, offs = _cases()
    batch = signal.RawBatch(raws, scal, offs)
    batch.normalise()
    table = batch.chunk_table(chunksize, overlap)
    got = batch.chunks(table, chunksize).cpu()
    want = []
    for i, (raw, sc, of) in enumerate(zip(raws, scal, offs)):
        rd = reader.Read("r%d" % i, raw, scaling=sc, offset=of)
        if len(rd.signal):
            want.append(chunk(torch.from_numpy(rd.signal), chunksize, overlap).to(torch.float16))
    want = torch.cat(want)
    assert got.shape == want.shape
    assert torch.equal(got, want)


def test_device_fixed_pa_strategy_and_no_trim():
    raws, scal, offs = _cases()
    batch = signal.RawBatch(raws[:4], scal[:4], offs[:4])
    strat, prm = {"strategy": "pa"}, {"standardise": 1, "mean": 91.25, "stdev": 22.5}
    shift, scale, trim = batch.normalise(strat, prm, do_trim=False)
    assert (shift == 91.25).all() and (scale == 22.5).all() and (trim == 0).all()
    table = batch.chunk_table(2000, 100)
    got = batch.chunks(table, 2000).cpu()
    want = torch.cat([chunk(torch.from_numpy(reader.Read("r", raw, scaling=sc, offset=of, do_trim=False, scaling_strategy=strat,
                                                         norm_params=prm).signal), 2000, 100).to(torch.float16)
                      for raw, sc, of in zip(raws[:4], scal[:4], offs[:4])])
    assert torch.equal(got, want)


def test_basecall_raw_equals_basecall_on_reader_reads():
    """End to end: raw int16 reads through the device ingest give the same calls as reader.Read + the host chunking path."""
    from bonito_amd import synthetic
    from bonito_amd.crf import basecall
    from bonito_amd.crf.basecall import basecall_raw

    class Raw:
        def __init__(self, i, raw, scaling, offset):
            self.read_id, self.raw, self.scaling, self.offset = "read_%d" % i, raw, scaling, offset

    raws, scal, offs = _cases()
    model = synthetic.make_model("fast", batchsize=16, chunksize=3996)
    model.use_koi(batchsize=16, chunksize=3996, quantize=False)
    model = model.half().cuda()
    host_reads = [reader.Read("read_%d" % i, r, scaling=s, offset=o) for i, (r, s, o) in enumerate(zip(raws, scal, offs))]
    want = {rd.read_id: res for rd, res in basecall(model, host_reads, chunksize=3996, overlap=498, batchsize=16)}
    raw_reads = [Raw(i, r, s, o) for i, (r, s, o) in enumerate(zip(raws, scal, offs))]
    got = {rd.read_id: (rd, res) for rd, res in basecall_raw(model, raw_reads, chunksize=3996, overlap=498, batchsize=16)}
    assert set(got) == set(want)
    for hr in host_reads:
        rd, res = got[hr.read_id]
        assert rd.trimmed_samples == hr.trimmed_samples and rd.shift == float(hr.shift) and rd.scale == float(hr.scale)
        assert res["sequence"] == want[hr.read_id]["sequence"]
        assert res["qstring"] == want[hr.read_id]["qstring"]
        assert np.array_equal(res["moves"], want[hr.read_id]["moves"])
