#!/usr/bin/env python3
"""Builds build/gemmlab/lab_<name>: tools/gemm_lab.hip around variants of the generated K-tile stream (tools/gen_gemmstep.py options), here on
the CPU box (hipcc cross-compiles); tools/gpu_call.sh `gemmlab` runs them all on the GPU box. Variants with --strip give wrong results
on purpose (timing of the stream without its DMA / fragment reads / barrier)."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "build", "gemmlab")
VARIANTS = {
    "base": [],
    "wait0": ["--wait-at", "0"],
    "wait6": ["--wait-at", "6"],
    "d1fast": ["--d1-every", "1"],
    "dma_nt": ["--dma-flags", "nt"],
    "dma_sc1": ["--dma-flags", "sc1"],
    "dma_sc01": ["--dma-flags", "sc0 sc1"],
    "nodma": ["--strip", "dma"],
    "nolds": ["--strip", "lds"],
    "nobar": ["--strip", "bar"],
    "mfma": ["--strip", "dma,lds,bar"],
    # epilogue experiments on the MFMA-only stream ("-D..." entries are compiler flags)
    "epi_nostore": ["--strip", "dma,lds,bar", "-DBH_EPI_NOSTORE"],
    "epi_nolds": ["--strip", "dma,lds,bar", "-DBH_EPI_NOLDS"],
    "epi_valu": ["--strip", "dma,lds,bar", "-DBH_EPI_NOLDS", "-DBH_EPI_NOSTORE"],
    "epi_ldsonly": ["--strip", "dma,lds,bar", "-DBH_EPI_LDSONLY"],
}


def build(name, opts):
    inc = os.path.join(OUT, "ktile_%s.inc" % name)
    defs = [o for o in opts if o.startswith("-D")]
    opts = [o for o in opts if not o.startswith("-D")]
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_gemmstep.py"), "--out", inc] + opts, check=True, stdout=subprocess.DEVNULL)
    exe = os.path.join(OUT, "lab_%s" % name)
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + os.path.join(ROOT, "include"),
           "-I" + os.path.join(ROOT, "bonito_amd", "csrc"), '-DBH_GEMM_KTILE_INC="%s"' % inc, "-Wno-unused-function"] + defs + [
           os.path.join(ROOT, "tools", "gemm_lab.hip"), "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode:
        raise RuntimeError(r.stderr[-3000:])
    return exe


def main():
    os.makedirs(OUT, exist_ok=True)
    names = sys.argv[1:] or list(VARIANTS)
    with ThreadPoolExecutor(max_workers=8) as ex:
        for exe in ex.map(lambda n: build(n, VARIANTS[n]), names):
            print(exe)


if __name__ == "__main__":
    main()
