"""End-to-end basecall pipeline on the GPU (chunk -> batch -> HIP encoder -> HIP beam search -> stitch)
against an independent serial restatement that decodes the same HIP scores with the CPU oracle (-m gpu)."""
import os
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


def test_basecall_reverse_gives_reverse_complement_calls():
    """--revcomp path (crf/basecall.py:35, reverse=True): scores are permuted on the device before decoding."""
    model = _model(996, 4)
    rng = np.random.default_rng(9)
    reads = _reads(rng, [3000])
    (_, fwd), = list(crf_basecall_fn(model, iter(reads), chunksize=996, overlap=96, batchsize=4, decoder="viterbi"))
    (_, rev), = list(crf_basecall_fn(model, iter(reads), chunksize=996, overlap=96, batchsize=4, decoder="viterbi",
                                      reverse=True))
    comp = {"A": "T", "C": "G", "G": "C", "T": "A"}
    rc = "".join(comp[b] for b in reversed(fwd["sequence"]))
    import difflib
    assert difflib.SequenceMatcher(None, rc, rev["sequence"], autojunk=False).ratio() > 0.9


def test_load_model_from_directory_roundtrip(tmp_path):
    """util.load_model: config.toml + weights_N.tar with foreign key names -> HIP model (reference util.py:271-311)."""
    import json
    from conftest import load_nn_fixture, ref_scores_to_koi
    from bonito_amd import util
    cfg, sd, x, y = load_nn_fixture("lstm64_sl3")

    def toml_value(v):
        if isinstance(v, bool):
            return "true" if v else "false"
        if isinstance(v, str):
            return json.dumps(v)
        if isinstance(v, list):
            return "[" + ", ".join(toml_value(i) for i in v) + "]"
        return repr(v)

    lines = ['[model]', 'package = "bonito.crf"', '[labels]', 'labels = ["N", "A", "C", "G", "T"]', '[input]',
             'features = 1', '[global_norm]', 'state_len = 3', '[basecaller]', 'batchsize = 4', 'chunksize = 1200',
             'overlap = 120', '[encoder]', 'type = "serial"']
    for sub in cfg["sublayers"]:
        lines.append("[[encoder.sublayers]]")
        lines += ["%s = %s" % (k, toml_value(v)) for k, v in sub.items()]
    (tmp_path / "config.toml").write_text("\n".join(lines) + "\n")
    renamed = {"module.layer%d" % i: v for i, (k, v) in enumerate(sd.items())}     # match_names + "module." stripping
    torch.save(renamed, str(tmp_path / "weights_3.tar"))
    torch.save({}, str(tmp_path / "weights_1.tar"))
    model = util.load_model(str(tmp_path), "cuda", use_koi=True)
    assert model.config["basecaller"]["chunksize"] == 1200 and model.config["basecaller"]["overlap"] == 120
    got = model(x.half().cuda()).cpu().float()
    assert (got - ref_scores_to_koi(y)).abs().max().item() < 6e-2
    assert util.load_symbol(str(tmp_path), "basecall") is crf_basecall_fn


def test_cli_basecaller_end_to_end(tmp_path, capsys):
    """python -m bonito_amd basecaller <model_dir> <reads_dir>: model directory + .npy reads -> FASTQ on stdout,
    summary.tsv, 'samples per second' on stderr (reference cli/basecaller.py flow)."""
    import json
    from conftest import load_nn_fixture
    from bonito_amd.__main__ import main
    cfg, sd, _, _ = load_nn_fixture("lstm64_sl3")

    def tv(v):
        if isinstance(v, bool):
            return "true" if v else "false"
        if isinstance(v, str):
            return json.dumps(v)
        if isinstance(v, list):
            return "[" + ", ".join(tv(i) for i in v) + "]"
        return repr(v)

    mdir, rdir = tmp_path / "model", tmp_path / "reads"
    mdir.mkdir(); rdir.mkdir()
    lines = ['[model]', 'package = "bonito.crf"', '[labels]', 'labels = ["N", "A", "C", "G", "T"]', '[input]',
             'features = 1', '[global_norm]', 'state_len = 3', '[basecaller]', 'batchsize = 8', 'chunksize = 1200',
             'overlap = 120', '[encoder]', 'type = "serial"']
    for sub in cfg["sublayers"]:
        lines.append("[[encoder.sublayers]]")
        lines += ["%s = %s" % (k, tv(v)) for k, v in sub.items()]
    (mdir / "config.toml").write_text("\n".join(lines) + "\n")
    sd = {k: (v * 30.0 if k.endswith("linear.weight") else v) for k, v in sd.items()}
    torch.save(sd, str(mdir / "weights_1.tar"))
    rng = np.random.default_rng(3)
    for i, n in enumerate([4000, 9000, 700]):
        np.save(rdir / ("read%d.npy" % i), (rng.standard_normal(n) * 12 + 90).astype(np.float32))
    rc = main(["basecaller", str(mdir), str(rdir), "--summary", str(tmp_path / "summary.tsv"), "--batchsize", "8"])
    out, err = capsys.readouterr()
    assert rc == 0 and "samples per second" in err and "completed reads: 3" in err
    recs = out.strip().split("\n")
    assert len(recs) == 12 and recs[0].startswith("@read0") and "mv:B:c,6," in recs[0]
    assert set(recs[1]) <= set("ACGT") and len(recs[1]) == len(recs[3])
    assert len((tmp_path / "summary.tsv").read_text().splitlines()) == 4


def test_cli_device_ingest_equals_host_ingest(tmp_path, capsys):
    """`--device-ingest` on int16 .npy reads (+ calibration side-cars) writes exactly the FASTQ the numpy ingest writes."""
    import json
    from conftest import load_nn_fixture
    from bonito_amd.__main__ import main
    cfg, sd, _, _ = load_nn_fixture("lstm64_sl3")

    def tv(v):
        if isinstance(v, bool):
            return "true" if v else "false"
        if isinstance(v, str):
            return json.dumps(v)
        if isinstance(v, list):
            return "[" + ", ".join(tv(i) for i in v) + "]"
        return repr(v)

    mdir, rdir = tmp_path / "model", tmp_path / "reads"
    mdir.mkdir(); rdir.mkdir()
    lines = ['[model]', 'package = "bonito.crf"', '[labels]', 'labels = ["N", "A", "C", "G", "T"]', '[input]',
             'features = 1', '[global_norm]', 'state_len = 3', '[basecaller]', 'batchsize = 8', 'chunksize = 1200',
             'overlap = 120', '[encoder]', 'type = "serial"']
    for sub in cfg["sublayers"]:
        lines.append("[[encoder.sublayers]]")
        lines += ["%s = %s" % (k, tv(v)) for k, v in sub.items()]
    (mdir / "config.toml").write_text("\n".join(lines) + "\n")
    sd = {k: (v * 30.0 if k.endswith("linear.weight") else v) for k, v in sd.items()}
    torch.save(sd, str(mdir / "weights_1.tar"))
    rng = np.random.default_rng(5)
    for i, n in enumerate([6000, 13000, 900, 2500]):
        x = rng.normal(480, 60, n)
        x[60:260] += 400
        np.save(rdir / ("read%d.npy" % i), np.clip(np.round(x), -32768, 32767).astype(np.int16))
        (rdir / ("read%d.json" % i)).write_text(json.dumps({"scale": 0.1755, "offset": -243.0 + i, "sample_rate": 4000.0}))
    outs = []
    for extra in ([], ["--device-ingest"]):
        rc = main(["basecaller", str(mdir), str(rdir), "--summary", str(tmp_path / "summary.tsv"), "--batchsize", "8"] + extra)
        out, err = capsys.readouterr()
        assert rc == 0 and "completed reads: 4" in err
        outs.append(out)
        outs.append((tmp_path / "summary.tsv").read_text())
    assert outs[0] == outs[2] and len(outs[0].strip().split("\n")) == 16
    assert outs[1] == outs[3]


def test_full_size_pipelined_steps_equal_serial_execution():
    """BASELINE size (hac, 512 x 10000, beam decode): the two-stream software pipeline of bench.py / basecall (encoder of batch
    i+1 next to the decode of batch i, persistent LSTM workgroups sharing CUs with the decode kernels) must produce exactly
    the bytes of a serial run, step after step, and the exchange must never time out under that contention."""
    from bonito_amd import decode, synthetic
    dev = torch.device("cuda", 0)
    model = synthetic.make_model("hac", batchsize=512, chunksize=10000)
    model.use_koi(batchsize=512, chunksize=10000, quantize=False)
    model = model.half().to(dev)
    gen = torch.Generator(device=dev).manual_seed(3)
    signals = [torch.randn(512, 1, 10000, generator=gen, device=dev).half() for _ in range(2)]
    serial = []
    for sig in signals:
        sc = model(sig)
        serial.append(torch.stack(decode.beam_search(sc)))
    model._hip.check()
    enc_stream, dec_stream = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    decs = [decode.CRFDecoder(512, 1667, 1024, dev, mode="beam") for _ in range(2)]
    tickets, outs = [None, None], []
    for i in range(6):
        with torch.cuda.stream(enc_stream):
            sc = model(signals[i & 1])
            ev = torch.cuda.Event()
            ev.record(enc_stream)
        if tickets[i & 1] is not None:
            outs.append(tickets[i & 1].result_planes())
        with torch.cuda.stream(dec_stream):
            dec_stream.wait_event(ev)
            sc.record_stream(dec_stream)
            tickets[i & 1] = decs[i & 1].submit(sc)
    for k in (0, 1):                       # steps 4 and 5, in order
        outs.append(tickets[k].result_planes())
    torch.cuda.synchronize()
    model._hip.check()
    assert len(outs) == 6
    for i, got in enumerate(outs):
        assert torch.equal(got, serial[i & 1]), "pipelined step %d differs from the serial run" % i


def _write_model_dir(mdir, fixture="lstm64_sl3", state_len=3, batchsize=8, chunksize=1200, overlap=120):
    import json
    from conftest import load_nn_fixture
    cfg, sd, _, _ = load_nn_fixture(fixture)

    def tv(v):
        if isinstance(v, bool):
            return "true" if v else "false"
        if isinstance(v, str):
            return json.dumps(v)
        if isinstance(v, list):
            return "[" + ", ".join(tv(i) for i in v) + "]"
        return repr(v)

    lines = ['[model]', 'package = "bonito.crf"', '[labels]', 'labels = ["N", "A", "C", "G", "T"]', '[input]',
             'features = 1', '[global_norm]', 'state_len = %d' % state_len, '[basecaller]', 'batchsize = %d' % batchsize,
             'chunksize = %d' % chunksize, 'overlap = %d' % overlap, '[encoder]', 'type = "serial"']
    for sub in cfg["sublayers"]:
        lines.append("[[encoder.sublayers]]")
        lines += ["%s = %s" % (k, tv(v)) for k, v in sub.items()]
    (mdir / "config.toml").write_text("\n".join(lines) + "\n")
    sd = {k: (v * 30.0 if k.endswith("linear.weight") else v) for k, v in sd.items()}
    torch.save(sd, str(mdir / "weights_1.tar"))


def test_cli_multi_process_devices_equals_single_device(tmp_path):
    """`--devices 0,0`: two worker processes (here both on GPU 0: the launcher, the record-level shard, the per-rank
    formatting and rank 0's ordered streaming writer are what is under test) write exactly the bytes of the one-process run.
    lstm96 fixture: the ring-in-a-workgroup LSTM kernel has no co-residency requirement, so two processes may share a GPU."""
    import subprocess
    import sys
    from conftest import ROOT
    mdir, rdir = tmp_path / "model", tmp_path / "reads"
    mdir.mkdir(); rdir.mkdir()
    _write_model_dir(mdir, fixture="lstm96_sl3")
    rng = np.random.default_rng(8)
    for i in range(9):
        np.save(rdir / ("read%d.npy" % i), (rng.standard_normal(int(rng.integers(700, 9000))) * 12 + 90).astype(np.float32))
    outs = []
    for extra in ([], ["--devices", "0,0"], ["--devices", "0,0,0", "--sam"], ["--sam"]):
        summ = tmp_path / ("summary%d.tsv" % len(outs))
        r = subprocess.run([sys.executable, "-m", "bonito_amd", "basecaller", str(mdir), str(rdir), "--summary", str(summ),
                            "--batchsize", "8"] + extra, capture_output=True, text=True, cwd=ROOT, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        assert "completed reads: 9" in r.stderr and "samples per second" in r.stderr
        outs.append(("\n".join(l for l in r.stdout.split("\n") if not l.startswith("@PG")), summ.read_text()))
    assert outs[0] == outs[1] and len(outs[0][0].strip().split("\n")) == 36
    assert outs[2] == outs[3] and outs[2][0].startswith("@HD")


def test_cli_worker_death_is_absorbed_by_rank_0(tmp_path):
    """`--devices 0,0,0` with worker rank 1 killed after its second record (BONITO_AMD_FAULT_INJECT): the launcher lets the others
    finish, rank 0 basecalls the rest of the dead worker's shard on a second engine, and the output is the one-process output (SURVEY 5
    "failure detection": re-queue on another replica; reference seam bonito/multiprocessing.py:27-33)."""
    import subprocess
    import sys
    from conftest import ROOT
    mdir, rdir = tmp_path / "model", tmp_path / "reads"
    mdir.mkdir(); rdir.mkdir()
    _write_model_dir(mdir, fixture="lstm96_sl3")
    rng = np.random.default_rng(9)
    for i in range(14):
        np.save(rdir / ("read%02d.npy" % i), (rng.standard_normal(int(rng.integers(700, 9000))) * 12 + 90).astype(np.float32))
    outs = []
    for extra, env in (([], {}), (["--devices", "0,0,0"], {"BONITO_AMD_FAULT_INJECT": "1:2", "BONITO_AMD_TEST_HOOKS": "1"})):
        summ = tmp_path / ("summary%d.tsv" % len(outs))
        r = subprocess.run([sys.executable, "-m", "bonito_amd", "basecaller", str(mdir), str(rdir), "--summary", str(summ),
                            "--batchsize", "8"] + extra, capture_output=True, text=True, cwd=ROOT, timeout=600, env=dict(os.environ, **env))
        # 0 = complete; 3 = complete, every read written, but a worker was lost on the way (a scheduler can tell the two apart)
        assert r.returncode == (3 if env else 0), r.stderr[-2000:]
        assert "completed reads: 14" in r.stderr
        outs.append((r.stdout, summ.read_text(), r.stderr))
    assert outs[0][:2] == outs[1][:2] and len(outs[0][0].strip().split("\n")) == 56
    assert "rank 1 is gone after" in outs[1][2] and "completed WITHOUT rank 1" in outs[1][2]


def test_exchange_timeout_surfaces_as_an_exception_never_as_output():
    """The persistent LSTM kernels bound every spin; on a timeout they raise a device flag and finish with INVALID output.
    The product path never yields such a batch: `basecall` reads the flag (mirrored to pinned host memory behind every forward)
    after each batch's decode, runs a flagged batch again with nothing else in flight, and raises if that fails as well.
    Provoked here by lowering the spin bound to 0 (the first incomplete poll round is a timeout, in the serial re-run too)."""
    from bonito_amd import _lib, synthetic
    dev = torch.device("cuda", 0)
    model = synthetic.make_model("hac", batchsize=64, chunksize=3000)
    model.use_koi(batchsize=64, chunksize=3000, quantize=False)
    model = model.half().to(dev)
    rng = np.random.default_rng(1)
    reads = _reads(rng, [9000] * 40)
    good = list(crf_basecall_fn(model, iter(reads), chunksize=3000, overlap=300, batchsize=64))
    assert len(good) == 40
    try:
        decode.set_option("lstm_max_spins", 0)
        with pytest.raises(_lib.HipEngineError):
            list(crf_basecall_fn(model, iter(reads), chunksize=3000, overlap=300, batchsize=64))
    finally:
        decode.set_option("lstm_max_spins", -1)      # back to the default bound
    # the aborted pipeline's producer threads may still be launching forwards (with the low bound they read at launch time):
    # let them drain, then clear the sticky flag with the synchronising check
    # (the pipeline ran the flagged batch a second time, serially, before it gave up: that attempt's check has already reported
    # and cleared the flag; forwards launched by the producer threads meanwhile may have raised it again)
    import time
    time.sleep(1.5)
    torch.cuda.synchronize()
    try:
        model._hip.check()
    except _lib.HipEngineError:
        pass
    again = list(crf_basecall_fn(model, iter(reads), chunksize=3000, overlap=300, batchsize=64))
    assert [(r.read_id, res["sequence"]) for r, res in again] == [(r.read_id, res["sequence"]) for r, res in good]


def test_exchange_timeout_is_retried_serially_and_the_run_completes(monkeypatch):
    """A flagged batch (a persistent kernel timed out: its scores are invalid) is not the end of the run: the pipeline runs that
    batch again with nothing else in flight and carries on; the calls equal those of an undisturbed run. The flag is raised here
    by a check that fails for the second and the fourth batch; a real timeout on both attempts still raises (previous test)."""
    import importlib
    from bonito_amd import _lib, synthetic
    bc_mod = importlib.import_module("bonito_amd.crf.basecall")          # (the package attribute of that name is the function)
    dev = torch.device("cuda", 0)
    model = synthetic.make_model("hac", batchsize=64, chunksize=3000)
    model.use_koi(batchsize=64, chunksize=3000, quantize=False)
    model = model.half().to(dev)
    rng = np.random.default_rng(2)
    reads = _reads(rng, [9000] * 60)
    good = [(r.read_id, res["sequence"], res["qstring"], res["moves"].tobytes())
            for r, res in crf_basecall_fn(model, iter(reads), chunksize=3000, overlap=300, batchsize=64, per_call=1)]
    calls, retried = {"n": 0}, []
    real_check, real_retry = bc_mod._Pipeline.check_forward, bc_mod._Pipeline._retry_serially

    def flaky_check(self, lane, ticket):
        calls["n"] += 1
        if calls["n"] in (2, 4):
            raise _lib.HipEngineError("injected: device-side timeout in a persistent kernel")
        return real_check(self, lane, ticket)

    def counting_retry(self, dev_batch, lane=0):
        retried.append(dev_batch.shape[0])
        return real_retry(self, dev_batch, lane)

    monkeypatch.setattr(bc_mod._Pipeline, "check_forward", flaky_check)
    monkeypatch.setattr(bc_mod._Pipeline, "_retry_serially", counting_retry)
    again = [(r.read_id, res["sequence"], res["qstring"], res["moves"].tobytes())
             for r, res in crf_basecall_fn(model, iter(reads), chunksize=3000, overlap=300, batchsize=64, per_call=1)]
    assert len(retried) == 2 and again == good and len(good) == 60


def test_real_timeouts_of_consecutive_batches_are_each_retried(monkeypatch):
    """Advisor finding (round 3): with ONE sticky flag per engine, the serial retry of batch i cleared the flag that batch i+1 -
    already in flight behind it - had raised as well, and batch i+1 was yielded from invalid scores. Flags are per forward now
    (bh_encoder_error_flag_at): here the forwards of two CONSECUTIVE batches really time out (spin bound 0 while exactly those two
    are launched from the encoder thread; the retries run from the decode thread with the default bound), both must be run again,
    and the calls must equal those of an undisturbed run."""
    import importlib
    import threading
    from bonito_amd import _lib, synthetic
    bc_mod = importlib.import_module("bonito_amd.crf.basecall")
    dev = torch.device("cuda", 0)
    model = synthetic.make_model("hac", batchsize=64, chunksize=3000)
    model.use_koi(batchsize=64, chunksize=3000, quantize=False)
    model = model.half().to(dev)
    rng = np.random.default_rng(4)
    reads = _reads(rng, [9000] * 80)
    good = [(r.read_id, res["sequence"], res["qstring"], res["moves"].tobytes())
            for r, res in crf_basecall_fn(model, iter(reads), chunksize=3000, overlap=300, batchsize=64, per_call=1)]
    real_forward = bc_mod._Pipeline._forward
    state = {"n": 0, "enc_thread": None, "pipes": []}

    def sabotaged_forward(self, lane, x):
        me = threading.current_thread()
        if state["enc_thread"] is None:
            state["enc_thread"] = me
            state["pipes"].append(self)
        if me is not state["enc_thread"]:                # a serial retry (decode thread): undisturbed
            return real_forward(self, lane, x)
        state["n"] += 1
        if state["n"] in (2, 3):
            decode.set_option("lstm_max_spins", 0)       # read when the kernels are launched: only this forward
            try:
                return real_forward(self, lane, x)
            finally:
                decode.set_option("lstm_max_spins", -1)
        return real_forward(self, lane, x)

    monkeypatch.setattr(bc_mod._Pipeline, "_forward", sabotaged_forward)
    again = [(r.read_id, res["sequence"], res["qstring"], res["moves"].tobytes())
             for r, res in crf_basecall_fn(model, iter(reads), chunksize=3000, overlap=300, batchsize=64, per_call=1)]
    assert state["pipes"] and state["pipes"][0].retries == 2, state["pipes"][0].retries
    assert again == good and len(good) == 80
    torch.cuda.synchronize()
    # both timeouts were repaired and acknowledged (bh_encoder_ack): neither the engine-wide poll nor a synchronising check reports them
    model._hip.poll()
    model._hip.check()


def test_batches_per_engine_call_do_not_change_the_calls():
    """`per_call` (batches per engine call; automatic = calls of up to 2048 chunks for the 384-wide fp16 model, whose recurrent
    kernel then pairs rings) groups chunks differently and nothing else: identical records for 1, 2, 4 and the automatic value."""
    from bonito_amd import synthetic
    from bonito_amd.crf.basecall import batches_per_call, max_lanes
    model = synthetic.make_model("hac", batchsize=256, chunksize=2400)
    model.use_koi(batchsize=256, chunksize=2400, quantize=False)
    model = model.half().to("cuda")
    assert batches_per_call(model, 512) == 4 and batches_per_call(model, 256) == 8 and batches_per_call(model, 2048) == 1
    assert batches_per_call(model, 512, chunksize=10000) == 4 and batches_per_call(model, 512, chunksize=20000) == 2     # (8 GiB of scores per call)
    assert batches_per_call(model, 512, chunksize=40000) == 1
    assert batches_per_call(model, 512, quantize=True) == 1 and max_lanes(model) == 1 and max_lanes(model, True) == 2
    fast = synthetic.make_model("fast", batchsize=64, chunksize=2400)
    assert batches_per_call(fast, 512) == 1 and batches_per_call(fast, 512, lanes=3) == 8 and max_lanes(fast) > 8
    rng = np.random.default_rng(6)
    reads = _reads(rng, [6000 + 500 * (i % 7) for i in range(150)])
    outs = [[(r.read_id, res["sequence"], res["qstring"], res["moves"].tobytes())
             for r, res in crf_basecall_fn(model, iter(reads), chunksize=2400, overlap=240, batchsize=256, per_call=k)] for k in (1, 2, 4, 0)]
    assert outs[0] == outs[1] == outs[2] == outs[3] and len(outs[0]) == 150
    assert "wgx2" in model._hip.describe()


@pytest.mark.parametrize("mode,kw", [("fastq", {}), ("sam", {"rna": True}), ("fasta", {"reverse": True})])
def test_records_path_equals_basecall_plus_format_record(mode, kw):
    """The writers' path of the CLI (`basecall_records`: one library call per read behind the decoder) against the reference-shaped
    API (`basecall` -> `io.format_record`) on the same reads through the same engine: identical record text, summary rows and log."""
    import importlib
    from bonito_amd import io as bio, synthetic
    bc = importlib.import_module("bonito_amd.crf.basecall")
    model = synthetic.make_model("hac", batchsize=64, chunksize=2400)
    model.use_koi(batchsize=64, chunksize=2400, quantize=False)
    model = model.half().to("cuda")
    rng = np.random.default_rng(9)
    reads = _reads(rng, [700, 2400, 2401, 5000, 9000, 30000, 2399, 12345])
    for r in reads:
        r.run_id, r.filename, r.channel, r.mux, r.start, r.duration = "runA", "x.npy", 1, 1, 0.0, 1.0
        r.template_start, r.template_duration, r.num_samples, r.trimmed_samples = 0.0, 1.0, len(r.signal) + 5, 5
    want = [bio.format_record(r, res, mode, 0.0)
            for r, res in crf_basecall_fn(model, iter(reads), chunksize=2400, overlap=240, batchsize=64, **kw)]
    got = list(bc.basecall_records(model, iter(reads), mode, chunksize=2400, overlap=240, batchsize=64, min_qscore=0.0, **kw))
    assert got == want and len(got) == len(reads) and any(t is not None for t, _, _ in got)


def test_concurrent_engines_never_yield_wrong_calls():
    """Three engine replicas of a model whose recurrent kernel needs all of its workgroups co-resident (hac, 512-chunk batches
    = 256 workgroups each), driven concurrently from three streams next to decode work, with a low spin bound: whenever the
    dispatcher interleaves their workgroups a ring starves and times out. Every batch must either raise or equal the serial
    result -- a timeout may never surface as output."""
    from bonito_amd import _lib, synthetic
    dev = torch.device("cuda", 0)
    models = []
    for _ in range(3):
        m = synthetic.make_model("hac", batchsize=512, chunksize=2400)
        m.use_koi(batchsize=512, chunksize=2400, quantize=False)
        models.append(m.half().to(dev))
    x = torch.randn(512, 1, 2400, generator=torch.Generator().manual_seed(3)).half().to(dev)
    want = models[0](x).clone()
    models[0]._hip.check()
    streams = [torch.cuda.Stream(dev) for _ in range(3)]
    ok = bad = 0
    try:
        decode.set_option("lstm_max_spins", 20000)
        for it in range(4):
            outs = []
            for m, st in zip(models, streams):
                with torch.cuda.stream(st):
                    outs.append(m(x))
            with torch.cuda.stream(streams[0]):
                decode.beam_search(want[:64].contiguous())          # decode kernels competing for the CUs
            torch.cuda.synchronize(dev)
            for m, sc in zip(models, outs):
                try:
                    m._hip.poll()
                except _lib.HipEngineError:
                    bad += 1
                    with pytest.raises(_lib.HipEngineError):
                        m._hip.check()                              # synchronising variant agrees, and clears the flag
                    continue
                ok += 1
                assert torch.equal(sc, want), "a batch that reported no timeout differs from the serial result"
    finally:
        decode.set_option("lstm_max_spins", -1)
    assert ok + bad == 12
    for m in models:
        m._hip.check()
        assert torch.equal(m(x), want)


def test_engine_is_rebuilt_after_weights_change():
    """load_state_dict / .half() / apply(fuse_bn_) after a forward must not leave the engine on stale weights."""
    from bonito_amd.nn import fuse_bn_
    model = _model(996, 4)
    x = torch.randn(2, 1, 996, generator=torch.Generator().manual_seed(0)).half().cuda()
    a = model(x).clone()
    sd = {k: (v * 0.5 if "rnn.weight_hh" in k else v) for k, v in model.state_dict().items()}
    model.load_state_dict(sd)
    b = model(x).clone()
    assert not torch.equal(a, b)
    model = model.apply(fuse_bn_)
    c = model(x)
    assert (c.float() - b.float()).abs().max().item() < 2e-2        # folding changes rounding only


def test_basecall_lanes_give_identical_results():
    """`lanes` engine replicas (batches in flight) never change what is called; with quantize=True the lanes run the 8-bit
    recurrent kernels (here the hac-shaped model at a small batch)."""
    from bonito_amd import synthetic
    rng = np.random.default_rng(4)
    reads = _reads(rng, [7000] * 30)
    for quantize in (False, True):
        model = synthetic.make_model("hac", batchsize=32, chunksize=3000)
        model.use_koi(batchsize=32, chunksize=3000, quantize=quantize)
        model = model.half().to("cuda")
        outs = []
        for lanes in (1, 2, 3):
            outs.append([(r.read_id, res["sequence"], res["qstring"], res["moves"].tobytes())
                         for r, res in crf_basecall_fn(model, iter(reads), chunksize=3000, overlap=300, batchsize=32, lanes=lanes)])
        assert outs[0] == outs[1] == outs[2] and len(outs[0]) == 30


def test_bench_default_mode_runs_and_reports_the_contract_fields():
    """`python bench.py` in its default mode (hac, four batches per engine call, paired recurrent kernel), shortened: one JSON line on
    stdout with the contract's fields, the roofline of the kernel the engine says it runs, no exchange timeout."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "8", "--warmup", "4", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout[-500:]                       # ONE json line, nothing else on stdout
    d = json.loads(lines[0])
    assert d["steps"] == 8 and d["n_gpus"] == 1 and d["unit"] == "samples/s" and d["higher_is_better"] is True
    assert d["value"] > 5e7 and abs(d["value"] - 512 * 10000 / (d["ms_per_step"] * 1e-3)) < 1e-3 * d["value"]
    assert "4 batches per engine call" in d["config"]["workload"] and "batch 512 x chunk 10000" in d["config"]["workload"]
    roof = d["roofline"]
    assert roof["kernel"] == "lstm_layer_wgx2_kernel<12,3>" and roof["bound"] == "mfma" and 0.1 < roof["frac"] < 1.0
    with open(os.path.join(root, "profiles", "pmc_traffic.json")) as fh:
        pmc = json.load(fh)["lstm_layer_wgx2_kernel|hac 1024x10000"]          # round 6: keyed by kernel AND workload
    assert roof["flops_per_launch"] == pytest.approx(4.027e12, rel=1e-3) and roof["traffic"] == pmc["bytes_per_launch"]
    assert 2.6e9 < roof["traffic"] < 2.8e9 and roof["traffic_source"] == pmc["source"]      # 2.62 GB algorithmic
    assert d["with_h2d"]["value"] > 5e7 and d["value_with_h2d"] == d["with_h2d"]["value"] and d["batches_per_engine_call"] == 4
    one = d["per_call_1"]                                        # the product path's call shape, in the same line
    assert one["value"] > 5e7 and "lstm_layer_wgx" in one["roofline"]["kernel"] and one["roofline"]["chunks_per_launch"] == 512
    assert set(d["other_configs"]) == {"fast", "sup", "sup_lstm", "sup_20000", "hac_quantize"}
    for name, leg in d["other_configs"].items():
        assert "error" not in leg and leg["value"] > 1e7, (name, leg)


def test_cli_pod5_input_equals_npy_input(tmp_path, capsys):
    """SURVEY 8(f)2 (round 6): `bonito basecaller` over a directory of .pod5 files (container + Arrow tables + VBZ codec read by
    bonito_amd/pod5.py, no wheel) writes exactly the records it writes for the same reads as int16 .npy + side-cars - through the numpy
    ingest and through `--device-ingest` (int16 ADC samples straight into bh_signal_chunks). Reference seam: bonito/pod5.py:52-67,113-124.
    The .pod5 is made by tests/pod5_fixture.py: the FORMAT is unpinned (no reference file exists offline), the plumbing is not."""
    import json
    import uuid
    from bonito_amd.__main__ import main
    from pod5_fixture import write_pod5
    mdir, d_npy, d_pod = tmp_path / "model", tmp_path / "npy", tmp_path / "pod5"
    mdir.mkdir(); d_npy.mkdir(); d_pod.mkdir()
    _write_model_dir(mdir)
    rng = np.random.default_rng(15)
    recs = []
    for i, n in enumerate([6000, 13000, 900, 2500, 30000]):
        x = rng.normal(480, 60, n)
        x[60:260] += 400
        sig = np.clip(np.round(x), -32768, 32767).astype(np.int16)
        rid = str(uuid.UUID(int=4000 + i))
        np.save(d_npy / ("read%d.npy" % i), sig)
        (d_npy / ("read%d.json" % i)).write_text(json.dumps({"read_id": rid, "scale": 0.1755, "offset": -243.0 + i, "sample_rate": 4000.0,
                                                             "run_id": "acq-test-0001", "channel": 7 + i, "mux": 3, "start": (250.0 * i) / 4000.0}))
        recs.append({"read_id": rid, "signal": sig, "scale": 0.1755, "offset": -243.0 + i, "channel": 7 + i, "well": 3, "start": 250 * i})
    write_pod5(str(d_pod / "a.pod5"), recs[:3], rows=4096, sample_rate=4000)
    write_pod5(str(d_pod / "b.pod5"), recs[3:], rows=4096, sample_rate=4000)
    outs = {}
    for name, rdir in (("npy", d_npy), ("pod5", d_pod)):
        for extra in ([], ["--device-ingest"]):
            summ = tmp_path / "summary.tsv"
            rc = main(["basecaller", str(mdir), str(rdir), "--summary", str(summ), "--batchsize", "8", "--sam"] + extra)
            out, err = capsys.readouterr()
            assert rc == 0 and "completed reads: 5" in err
            body = "\n".join(ln for ln in out.split("\n") if not ln.startswith("@PG"))          # the @PG line carries the command line
            rows = [ln.split("\t") for ln in summ.read_text().strip().split("\n")]
            col = rows[0].index("filename") if "filename" in rows[0] else None
            outs[(name, bool(extra))] = (body, [[c for k, c in enumerate(r) if k != col] for r in rows])
    assert outs[("npy", False)] == outs[("pod5", False)]
    assert outs[("npy", True)] == outs[("pod5", True)]
    assert outs[("npy", False)][0] == outs[("npy", True)][0] and outs[("npy", False)][0].count("\n") > 5
