#!/usr/bin/env python3
"""Builds library variants that differ in the generated stream of the paired kernel's main loop (timing experiments):
    python tools/build_variants.py name1="--polls-at 30" name2="--poll-flags sc1" ...
Each variant = `gen_ringstep.py --preset unrolled <args>` -> lstm.hip recompiled -> build/variants/lib_<name>.so (cross-compiled here,
shipped to the GPU box with the snapshot; tools/gpu_variants.sh runs a command once per variant). The committed include is restored."""
import os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import build as B
from concurrent.futures import ThreadPoolExecutor

inc = os.path.join(ROOT, "bonito_amd", "csrc", "ringstep3u_mfma.inc")
keep = open(inc).read()
vdir = os.path.join(ROOT, "build", "variants")
shutil.rmtree(vdir, ignore_errors=True)
os.makedirs(vdir)
B.build_hip()
objs = [os.path.join(B.OBJ, f) for f in sorted(os.listdir(B.OBJ)) if f.endswith(".o")]
jobs = []
for spec in sys.argv[1:]:
    name, _, args = spec.partition("=")
    src = os.path.join(vdir, name + ".hip")
    vinc = os.path.join(vdir, name)
    os.makedirs(vinc)
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_ringstep.py"), "--preset", "unrolled"] + args.split() +
                   ["--out", os.path.join(vinc, "ringstep3u_mfma.inc")], check=True, stdout=subprocess.DEVNULL)
    jobs.append((name, vinc))


def one(job):
    name, vinc = job
    obj = os.path.join(vdir, name + ".o")
    # -I<variant dir> first: `#include "ringstep3u_mfma.inc"` resolves next to lstm.hip before the include path, so compile a copy of
    # lstm.hip from the variant directory with the other includes reachable through -I csrc
    shutil.copy(os.path.join(B.CSRC, "lstm.hip"), os.path.join(vinc, "lstm.hip"))
    cmd = [B.HIPCC] + B.FLAGS + ["-I" + B.CSRC, "-c", os.path.join(vinc, "lstm.hip"), "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode:
        raise RuntimeError(r.stderr[-3000:])
    lib = os.path.join(vdir, "lib_%s.so" % name)
    others = [o for o in objs if not o.endswith("lstm.hip.o")]
    subprocess.run([B.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, obj] + others, check=True)
    shutil.rmtree(vinc); os.remove(obj)
    return lib


with ThreadPoolExecutor(max_workers=6) as ex:
    for lib in ex.map(one, jobs):
        print("built", lib)
assert open(inc).read() == keep
