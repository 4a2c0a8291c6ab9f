"""Reader / writer host logic of the basecaller CLI (CPU): trim + normalisation against values produced by the
reference's bonito/reader.py, the in-tree known answers (io.py:63-64 doctest, SAM.md tags)."""
import io
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN
from bonito_amd import io as bio
from bonito_amd import reader


def _regen(seed_idx, n, peak):
    """Same recipe as the fixture generator (tests/golden: reader_cases.npz was produced from it)."""
    rng = np.random.default_rng(5)
    for i in range(seed_idx + 1):
        m = int(rng.integers(500, 20000))
        sig = rng.standard_normal(m).astype(np.float32)
        if i % 2 == 0:
            a = int(rng.integers(50, 400)); b = a + int(rng.integers(60, 600))
            sig[a:b] += 4.0
    assert m == n
    return (sig * 12.0 + 90.0).astype(np.float32)


def test_trim_and_normalisation_match_reference_values():
    meta = json.loads(str(np.load(os.path.join(GOLDEN, "reader_cases.npz"))["meta"]))
    for c in meta:
        raw = _regen(c["seed"], c["n"], c["peak"])
        shift, scale = reader.normalisation(raw, None, None)
        assert abs(shift - c["shift"]) < 1e-4 and abs(scale - c["scale"]) < 1e-4
        assert reader.trim(raw, threshold=scale * 2.4 + shift) == c["trim"]
        r = reader.Read("x", raw)
        assert r.trimmed_samples == c["trim"] and len(r.signal) == c["n"] - c["trim"]
        assert abs(float(np.median(r.signal))) < 1.0
    assert reader.normalisation(raw, {"strategy": "pa"}, {"standardise": 1, "mean": 93.7, "stdev": 23.5}) == (93.7, 23.5)
    assert reader.normalisation(raw, {"strategy": "pa"}, {"standardise": 0}) == (0.0, 1.0)
    with pytest.raises(ValueError):
        reader.normalisation(raw, {"strategy": "pa"}, None)


def test_encode_moves_doctest_and_records():
    assert bio.encode_moves(np.array([0, 1, 0, 1, 1], dtype=np.int8), 5) == "5,0,1,0,1,1"      # bonito/io.py:63-64
    buf = io.StringIO()
    bio.write_fastq("rid", "ACGT", "!!!!", fd=buf, tags=["qs:f:1.00", "ns:i:9"])
    assert buf.getvalue() == "@rid qs:f:1.00\tns:i:9\nACGT\n+\n!!!!\n"
    rec = bio.sam_record("rid", "ACGT", "IIII", tags=["mv:B:c,6,1,0"]).split("\t")
    assert rec[:3] == ["rid", "4", "*"] and rec[9] == "ACGT" and rec[-1] == "mv:B:c,6,1,0"
    assert bio.sam_header([]).startswith("@HD\tVN:1.5")


def test_writer_filters_and_logs(tmp_path):
    class R:
        read_id, filename, run_id, channel, mux, start, duration = "r1", "f.npy", "run", 1, 2, 0.0, 1.0
        template_start, template_duration, num_samples, trimmed_samples = 0.1, 0.9, 5000, 40
    good = {"sequence": "ACGT", "qstring": "5555", "moves": np.array([1, 0, 1, 1, 0, 1], np.int8), "stride": 6}
    bad = {"sequence": "AC", "qstring": "!!", "moves": np.array([1, 1], np.int8), "stride": 6}
    buf = io.StringIO()
    w = bio.Writer("fastq", iter([(R, good), (R, bad)]), fd=buf, min_qscore=7.0, summary_path=str(tmp_path / "s.tsv"))
    w.start(); w.join()
    # like the reference (io.py:437-442): every read is logged BEFORE the q-score filter, with its post-trim sample count
    assert w.error is None and w.log == [("r1", 4960), ("r1", 4960)]
    out = buf.getvalue().splitlines()
    assert out[0].startswith("@r1 RG:Z:run\tqs:f:20.00\tns:i:5000\tts:i:40\tmv:B:c,6,1,0,1,1,0,1") and out[1] == "ACGT"
    rows = (tmp_path / "s.tsv").read_text().splitlines()
    assert rows[0].split("\t") == bio.summary_field_names and len(rows) == 2


def test_npy_reader(tmp_path):
    rng = np.random.default_rng(0)
    np.save(tmp_path / "a.npy", (rng.standard_normal(3000) * 10 + 80).astype(np.float32))
    np.save(tmp_path / "b.npy", (rng.standard_normal(2000) * 40 + 500).astype(np.int16))
    (tmp_path / "b.json").write_text(json.dumps({"read_id": "read-b", "scale": 0.18, "offset": -240.0, "channel": 7}))
    reads = list(reader.Reader(str(tmp_path)).get_reads())
    assert [r.read_id for r in reads] == ["a", "read-b"] and reads[1].channel == 7
    assert reads[0].signal.dtype == np.float32 and abs(float(np.median(reads[0].signal))) < 1.0
    assert [r.read_id for r in reader.Reader(str(tmp_path)).get_reads(read_ids={"a"}, skip=True)] == ["read-b"]
    with pytest.raises(FileNotFoundError):
        reader.Reader(str(tmp_path / "nope"))



def test_launcher_stops_everything_when_a_worker_dies_before_the_rendezvous(tmp_path):
    """Advisor, round 5: a worker that exits before every rank has joined (out of memory, a bad device, an import error) used to leave
    the others in `init_process_group` for gloo's default of 30 minutes, and the launcher returned 0 with workers missing. Now the
    launcher takes the run down at once and fails. (No GPU is touched: the hook fires, and the launcher reacts, before any model load.)"""
    import subprocess
    import sys
    import time
    from conftest import ROOT
    env = dict(os.environ, BONITO_AMD_TEST_HOOKS="1", BONITO_AMD_FAULT_INJECT="1:early", BONITO_AMD_RENDEZVOUS_TIMEOUT="120")
    t0 = time.time()
    r = subprocess.run([sys.executable, "-m", "bonito_amd", "basecaller", str(tmp_path / "no_model"), str(tmp_path), "--devices", "0,0,0"],
                       capture_output=True, text=True, cwd=ROOT, timeout=110, env=env)
    assert r.returncode == 1, (r.returncode, r.stderr[-1500:])
    assert "before every rank had joined" in r.stderr and "TEST HOOK ARMED" in r.stderr
    assert time.time() - t0 < 100          # not the rendezvous timeout, let alone gloo's 30 minutes


def test_fault_hook_is_inert_without_the_test_switch_and_rejects_malformed_values(monkeypatch):
    from bonito_amd.cli import basecaller as cli
    recs = [1, 2, 3]
    monkeypatch.setenv("BONITO_AMD_FAULT_INJECT", "0:1")
    monkeypatch.delenv("BONITO_AMD_TEST_HOOKS", raising=False)
    assert list(cli._fault_hook(iter(recs), 0, 2)) == recs          # set but not armed: ignored
    monkeypatch.setenv("BONITO_AMD_TEST_HOOKS", "1")
    monkeypatch.setenv("BONITO_AMD_FAULT_INJECT", "zero:one")
    with pytest.raises(SystemExit):
        cli._fault_hook(iter(recs), 0, 2)
    monkeypatch.setenv("BONITO_AMD_FAULT_INJECT", "1:2")
    assert list(cli._fault_hook(iter(recs), 0, 2)) == recs          # another rank's fault


def test_only_a_closed_connection_counts_as_a_dead_peer():
    """parallel.ordered_records rescues a rank's shard only when gloo reports its connection closed / reset; a timeout or a garbled
    message of a live rank is an error that propagates (advisor, round 5)."""
    from bonito_amd import parallel
    assert parallel.peer_is_gone(RuntimeError("[/pytorch/third_party/gloo/gloo/transport/tcp/pair.cc:534] Connection closed by peer [127.0.0.1]:4242"))
    assert parallel.peer_is_gone(RuntimeError("Read error [127.0.0.1]:1234: Connection reset by peer"))
    assert not parallel.peer_is_gone(RuntimeError("Timed out waiting 1800000ms for recv operation to complete"))
    assert not parallel.peer_is_gone(ValueError("connection closed by peer"))          # not a transport error type
    import pickle
    assert not parallel.peer_is_gone(pickle.UnpicklingError("invalid load key"))
