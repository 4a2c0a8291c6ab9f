import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from bonito_amd import signal
rng = np.random.default_rng(17)
x = rng.normal(480, 60, 50000)
a = int(rng.integers(20, 200)); b = a + int(rng.integers(80, 400)); x[a:b] += rng.normal(420, 30, b - a)
raw = np.clip(np.round(x), -32768, 32767).astype(np.int16)
sc, of = 0.1755, -243.0
scaled = np.array(sc * (raw.astype(np.float32) + of), dtype=np.float32)
batch = signal.RawBatch([raw], [sc], [of])
for q in (0.2, 0.9, 0.5, 0.37):
    prm = {"quantile_a": q, "quantile_b": q, "shift_multiplier": 0.5, "scale_multiplier": 1.0}
    shift, scale, trim = batch.normalise(None, prm, do_trim=False)
    print(q, "device", repr(shift[0]), "numpy", repr(float(np.quantile(scaled, q))), "sorted raw", np.sort(raw)[int((len(raw)-1)*q)])
