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
    # round 6: the K-tile on 16x16x32 MFMAs (gen_gemmstep.py --tile16; run the binaries with LAB_T16=1). t16_nowN: rows the epilogue stores itself
    "t16_base": ["--tile16"],
    "t16_b74": ["--tile16", "--barrier-at", "74"],
    "t16_b86": ["--tile16", "--barrier-at", "86"],
    "t16_r2": ["--tile16", "--read-every", "2"],
    "t16_d2fast": ["--tile16", "--d2-every16", "2"],
    "t16_now12": ["--tile16", "--now16", "12"],
    "t16_now16": ["--tile16", "--now16", "16"],
    "t16_now24": ["--tile16", "--now16", "24"],
    "t16_nowg0": ["--tile16", "--nowg", "0"],
    "t16_nowg4": ["--tile16", "--nowg", "4"],
    "t16_nowg12": ["--tile16", "--nowg", "12"],
    "t16_s16": ["--tile16", "--stagger", "16"],
    "t16_mfma": ["--tile16", "--strip", "dma,lds,bar"],
    "t16_nodma": ["--tile16", "--strip", "dma"],
    "t16_nolds": ["--tile16", "--strip", "lds"],
}


def build(name, opts):
    inc = os.path.join(OUT, "ktile_%s.inc" % name)
    defs = [o for o in opts if o.startswith("-D")]
    opts = [o for o in opts if not o.startswith("-D")]
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_gemmstep.py"), "--out", inc] + opts, check=True, stdout=subprocess.DEVNULL)
    exe = os.path.join(OUT, "lab_%s" % name)
    which = "BH_GEMM_KTILE16_INC" if "--tile16" in opts else "BH_GEMM_KTILE_INC"
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + os.path.join(ROOT, "include"),
           "-I" + os.path.join(ROOT, "bonito_amd", "csrc"), '-D%s="%s"' % (which, inc), "-Wno-unused-function"] + defs + [
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
