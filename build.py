"""Build libbonito_hip.so (gfx950) and the CPU oracle helpers in-tree.

    python build.py            # everything
    python build.py --force    # ignore timestamps

hipcc cross-compiles for gfx950 without a GPU present. Outputs (git-ignored, shipped to the GPU box):
    bonito_amd/libbonito_hip.so      the product: HIP kernels + engine + C ABI (include/bonito_hip.h)
    oracle/liboracle.so              CPU restatement used ONLY by tests / smoke / bench cpu_baseline
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(ROOT, "bonito_amd", "csrc")
OBJ = os.path.join(ROOT, "build", "obj")
LIB = os.path.join(ROOT, "bonito_amd", "libbonito_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
         "-I" + os.path.join(ROOT, "include")]
# signal.hip reproduces NumPy's arithmetic operation by operation: no fused multiply-adds there (hipcc's default
# -ffp-contract=fast ignores the in-source pragma)
# lstm_q8.hip: the pre-activation of the 8-bit recurrent path is DEFINED operation by operation (oracle/lstm_q8_ref.py); with
# contraction on, the compiler fused different multiply-add pairs in different template instances
EXTRA_FLAGS = {"signal.hip": ["-ffp-contract=off"], "lstm_q8.hip": ["-ffp-contract=off"]}
# Timing experiments only (e.g. BH_EXTRA_LSTM_FLAGS=-DBH_EXPT_SHARE: wrong results on purpose). Such a build NEVER writes the product
# library: objects go to build/obj_expt, the result is bonito_amd/libbonito_hip_expt.so, and bonito_amd/_lib.py loads that file only
# when BONITO_HIP_LIB names it (a stray environment variable used to produce a silently wrong libbonito_hip.so; review, round 3).
# BH_EXTRA_GEMM_FLAGS (e.g. -DBH_GEMM_STATS: another GemmArgs layout + cycle stamps) is treated the same way (advisor, round 4: it used
# to rebuild gemm.hip into the product object directory, and unsetting it did not rebuild - staleness is checked by mtime only).
EXPERIMENT = bool(os.environ.get("BH_EXTRA_LSTM_FLAGS") or os.environ.get("BH_EXTRA_GEMM_FLAGS") or os.environ.get("BH_EXTRA_BEAM_FLAGS")
                  or os.environ.get("BH_EXTRA_ATTN_FLAGS") or os.environ.get("BH_EXTRA_CONV_FLAGS"))
if EXPERIMENT:
    if os.environ.get("BH_EXTRA_LSTM_FLAGS"):
        EXTRA_FLAGS["lstm.hip"] = os.environ["BH_EXTRA_LSTM_FLAGS"].split()
    if os.environ.get("BH_EXTRA_GEMM_FLAGS"):
        EXTRA_FLAGS["gemm.hip"] = os.environ["BH_EXTRA_GEMM_FLAGS"].split()
    if os.environ.get("BH_EXTRA_BEAM_FLAGS"):
        EXTRA_FLAGS["beam.hip"] = os.environ["BH_EXTRA_BEAM_FLAGS"].split()
    if os.environ.get("BH_EXTRA_CONV_FLAGS"):          # e.g. -DBH_CONV_EXPT_NOSTORE: conv_front3_kernel without its output stores (timing only)
        EXTRA_FLAGS["conv.hip"] = os.environ["BH_EXTRA_CONV_FLAGS"].split()
    if os.environ.get("BH_EXTRA_ATTN_FLAGS"):          # e.g. -DBH_ATTN_EXPT: the elimination variants of attention_ring2_kernel (tools/attn_bench.py)
        EXTRA_FLAGS["attention.hip"] = os.environ["BH_EXTRA_ATTN_FLAGS"].split()
    OBJ = os.path.join(ROOT, "build", "obj_expt")
    LIB = os.path.join(ROOT, "bonito_amd", "libbonito_hip_expt.so")
    sys.stderr.write("build.py: BH_EXTRA_LSTM_FLAGS / BH_EXTRA_GEMM_FLAGS / BH_EXTRA_BEAM_FLAGS / BH_EXTRA_ATTN_FLAGS / BH_EXTRA_CONV_FLAGS set -> experimental library %s (the product library is not touched)\n" % LIB)


def _newer(dst, srcs):
    if not os.path.exists(dst):
        return True
    t = os.path.getmtime(dst)
    return any(os.path.getmtime(s) > t for s in srcs)


def build_hip(force=False):
    os.makedirs(OBJ, exist_ok=True)
    srcs = sorted(f for f in os.listdir(CSRC) if f.endswith((".hip", ".cpp")))
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".inc"))]
    hdrs.append(os.path.join(ROOT, "include", "bonito_hip.h"))
    objs, jobs = [], []
    for s in srcs:
        src = os.path.join(CSRC, s)
        obj = os.path.join(OBJ, s + ".o")
        objs.append(obj)
        if force or _newer(obj, [src] + hdrs):
            # host*.cpp: plain C++ (host-only helpers with x86 intrinsics); other .cpp files carry HIP runtime calls
            as_hip = s.endswith(".cpp") and not s.startswith("host")
            flags = [f for f in FLAGS if as_hip or s.endswith(".hip") or not f.startswith("--offload-arch")]
            cmd = [HIPCC] + flags + EXTRA_FLAGS.get(s, []) + (["-x", "hip"] if as_hip else []) + ["-c", src, "-o", obj]
            jobs.append(cmd)

    def run(cmd):
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("build failed: %s\n%s\n%s" % (" ".join(cmd), r.stdout, r.stderr))
        return r.stderr

    with ThreadPoolExecutor(max_workers=min(8, max(1, len(jobs)))) as ex:
        for warn in ex.map(run, jobs):
            if warn.strip():
                sys.stderr.write(warn)
    if force or jobs or _newer(LIB, objs):
        run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs)
    return LIB


def build_oracle(force=False):
    odir = os.path.join(ROOT, "oracle")
    src = os.path.join(odir, "crf_oracle.c")
    lib = os.path.join(odir, "liboracle.so")
    hdr = os.path.join(ROOT, "include", "bh_lse_table.h")
    if os.path.exists(src) and (force or _newer(lib, [src, hdr])):
        r = subprocess.run(["gcc", "-O2", "-std=c11", "-ffp-contract=off", "-fPIC", "-shared", "-o", lib, src, "-lm"],
                           capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("oracle build failed:\n" + r.stderr)
    return lib


def main():
    force = "--force" in sys.argv
    print(build_hip(force))
    print(build_oracle(force))


if __name__ == "__main__":
    main()
