#!/usr/bin/env python3
"""Generates bonito_amd/csrc/gemm_ktile_mfma.inc: ONE K-tile (64 deep) of the 256 x 256 linear-layer GEMM `gemm_w4_kernel` (gemm.hip)
as a hand-scheduled instruction stream for ONE wave per SIMD.

    wave tile 128 (features, MFMA A operand = W rows) x 128 (tokens, B operand = X rows), v_mfma_f32_32x32x16_f16:
    acc[i][j] += FA[i][ks] * FB[j][ks],   i, j in 0..3 (32 x 32 tiles), ks in 0..3 (16 deep)  = 64 MFMAs = 2048 matrix-pipe cycles
    per k-step 8 fragment reads (ds_read_b128, 1 KiB each) for 16 MFMAs: 4 MFMAs per KiB of LDS traffic (the 64 x 128 wave tile of
    gemm_big_kernel on 16 x 16 x 32: 2.7) and half the operand-register reads per FLOP.

What one stream instance does for K-tile g of a workgroup (stage s = g & 1 of the two 64 KiB LDS stages):
    ks = 0      16 MFMAs on the fragments of ks 0 (read by the previous instance); the 8 reads of ks 1; D1: the wave's eight
                LDS-DMA pieces of the X operand of K-tile g + 1 (into stage s ^ 1)
    ks = 1, 2   16 MFMAs each; the reads of ks 2 / 3
    ks = 3      WAIT_AT MFMAs, then: lgkmcnt(0) (all my reads of stage s are back), vmcnt(0) (my pieces of K-tile g + 1 have landed),
                s_barrier (everybody's have; nobody reads stage s any more); behind it the 8 reads of ks 0 of K-tile g + 1 (stage s ^ 1)
                and D2: my eight pieces of the W operand of K-tile g + 2 (into stage s)
One barrier per K-tile; every DMA has >= 1.1 k-tiles of matrix work between issue and the wait that retires it; nothing in the
stream waits for an instruction issued fewer than ~6 MFMAs (190 cycles) earlier. Fragment registers are double buffered by k-step
parity; reads are waited for with COUNTED lgkmcnt (they return in order), so a late fragment never holds up the MFMAs in front of it.
`FIRST` instances (first K-tile of an output tile) start the accumulators from the inline constant 0 instead of reading them.

The stream ends with lgkmcnt(0): the compiler may touch the fragment registers behind the statement. It does NOT pad the
MFMA -> vector-ALU hazard at its end: gemm_w4_kernel puts 16 wait states in front of the epilogue's first accumulator read.

usage: gen_gemmstep.py [--wait-at n] [--read-at n] [--d1-every n] [--out path] [--name fn]
"""
import argparse
import os


def build(wait_at=2, read_at=1, d1_at=0, d1_every=2, d2_every=1, first=False, name="gemm_ktile"):
    lines = []          # instruction text
    counts = {"mfma": 0, "lds": 0, "vmem": 0, "salu": 0, "wait": 0, "nop": 0}
    fifo = []           # fragment names with a read in flight, oldest first

    def emit(kind, text):
        counts[kind] += 1
        lines.append(text)

    def read(dst, addr, off):
        emit("lds", "ds_read_b128 %%[%s], %%[%s] offset:%d" % (dst, addr, off))
        fifo.append(dst)

    def need(*frags):
        """counted wait: everything up to the youngest of `frags` still in flight must be back"""
        idx = max((fifo.index(f) for f in frags if f in fifo), default=-1)
        if idx >= 0:
            emit("wait", "s_waitcnt lgkmcnt(%d)" % (len(fifo) - 1 - idx))
            del fifo[:idx + 1]

    def dma(mbase, imm, voff, sbase):
        emit("salu", "s_add_u32 m0, %%[%s], 0x%x" % (mbase, imm))
        emit("nop", "s_nop 0")
        emit("vmem", "global_load_lds_dwordx4 %%[%s], %%[%s]" % (voff, sbase))

    def frag_reads(p, ra, rb):
        """the 8 reads of one k-step in the order the MFMAs need them (i-major MFMA order): A0 B0 B1 B2 B3 A1 A2 A3"""
        order = [("a", 0)] + [("b", j) for j in range(4)] + [("a", i) for i in range(1, 4)]
        return [("f%s%d%d" % (op, p, n), ra if op == "a" else rb, n * 4096) for op, n in order]

    for ks in range(4):
        p = ks & 1
        if ks < 3:
            pending = frag_reads(p ^ 1, "ra%d" % (ks + 1), "rb%d" % (ks + 1))
            start = read_at
        else:
            pending = frag_reads(0, "rao", "rbo")        # K-tile g + 1, k-step 0, from the other stage: only behind the barrier
            start = wait_at
        d = []
        if ks == 0:
            d = [(d1_at + n * d1_every, ("md1", n * 1024, "vd1_%d" % n, "sd1")) for n in range(8)]
        elif ks == 3:
            d = [(wait_at + n * d2_every, ("md2", n * 1024, "vd2_%d" % n, "sd2")) for n in range(8)]
        m = 0
        for i in range(4):
            for j in range(4):
                if ks == 3 and m == wait_at:
                    fifo_was = list(fifo)
                    emit("wait", "s_waitcnt lgkmcnt(0)")
                    del fifo[:]
                    emit("wait", "s_waitcnt vmcnt(0)")
                    emit("wait", "s_barrier")
                    assert all(f[1] == str(p) or True for f in fifo_was)
                need("fa%d%d" % (p, i), "fb%d%d" % (p, j))
                c = "%%[c%d%d]" % (i, j)
                src = "0" if (first and ks == 0) else c
                emit("mfma", "v_mfma_f32_32x32x16_f16 %s, %%[fa%d%d], %%[fb%d%d], %s" % (c, p, i, p, j, src))
                # side instructions behind MFMA m of this k-step
                if m >= start and pending and (m - start) < 8:
                    read(*pending.pop(0))
                for at, args in d:
                    if at == m:
                        dma(*args)
                m += 1
        # anything that did not fit behind an MFMA (late wait_at): issue it now
        while pending:
            read(*pending.pop(0))
        for at, args in d:
            if at >= 16:
                dma(*args)
    emit("wait", "s_waitcnt lgkmcnt(0)")
    del fifo[:]

    # ---- the C++ wrapper ----------------------------------------------------------------------------------------------------------
    outs, ins = [], []
    for i in range(4):
        for j in range(4):
            outs.append('[c%d%d] "%s"(acc[%d][%d])' % (i, j, "=&a" if first else "+a", i, j))
    for n in range(4):
        outs.append('[fa0%d] "+v"(fa[0][%d])' % (n, n))
        outs.append('[fb0%d] "+v"(fb[0][%d])' % (n, n))
    for n in range(4):
        outs.append('[fa1%d] "=&v"(fa[1][%d])' % (n, n))
        outs.append('[fb1%d] "=&v"(fb[1][%d])' % (n, n))
    for ks in range(1, 4):
        ins.append('[ra%d] "v"(ra[%d])' % (ks, ks))
        ins.append('[rb%d] "v"(rb[%d])' % (ks, ks))
    ins += ['[rao] "v"(rao)', '[rbo] "v"(rbo)']
    for n in range(8):
        ins.append('[vd1_%d] "v"(vd1[%d])' % (n, n))
    for n in range(8):
        ins.append('[vd2_%d] "v"(vd2[%d])' % (n, n))
    ins += ['[sd1] "s"(sd1)', '[sd2] "s"(sd2)', '[md1] "s"(md1)', '[md2] "s"(md2)']
    stat = " ".join("%s=%d" % kv for kv in sorted(counts.items()))
    body = '"\n        "'.join(l + "\\n\\t" for l in lines[:-1])
    text = []
    text.append("// %s<%s>: wait_at=%d read_at=%d d1_at=%d d1_every=%d d2_every=%d : %s" % (
        name, "FIRST" if first else "", wait_at, read_at, d1_at, d1_every, d2_every, stat))
    sig = ("__device__ __forceinline__ void %s%s(float16_t (&acc)[4][4], half8_t (&fa)[2][4], half8_t (&fb)[2][4], "
           "const unsigned (&ra)[4], const unsigned (&rb)[4], unsigned rao, unsigned rbo, const unsigned (&vd1)[8], "
           "const unsigned (&vd2)[8], const char* sd1, const char* sd2, unsigned md1, unsigned md2) {" % (name, "_first" if first else ""))
    text.append(sig)
    text.append('    asm volatile("' + body + '"\n        "' + lines[-1] + '"')
    text.append("        : " + ", ".join(outs))
    text.append("        : " + ", ".join(ins))
    text.append('        : "memory", "scc");')
    text.append("}")
    return "\n".join(text) + "\n"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--wait-at", type=int, default=2)
    ap.add_argument("--read-at", type=int, default=1)
    ap.add_argument("--d1-at", type=int, default=0)
    ap.add_argument("--d1-every", type=int, default=2)
    ap.add_argument("--d2-every", type=int, default=1)
    ap.add_argument("--name", default="gemm_ktile")
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    kw = dict(wait_at=a.wait_at, read_at=a.read_at, d1_at=a.d1_at, d1_every=a.d1_every, d2_every=a.d2_every, name=a.name)
    head = ("// GENERATED by tools/gen_gemmstep.py - do not edit. One K-tile (64 deep) of gemm_w4_kernel's 128 x 128 wave tile as one\n"
            "// instruction stream: 64 v_mfma_f32_32x32x16_f16, the 32 fragment reads of the next k-steps, the wave's 16 LDS-DMA pieces of the\n"
            "// K-tiles ahead, one barrier. See the generator's docstring for the schedule and gemm.hip for the operands.\n")
    text = head + build(first=False, **kw) + "\n" + build(first=True, **kw)
    path = a.out or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bonito_amd", "csrc", "gemm_ktile_mfma.inc")
    with open(path, "w") as f:
        f.write(text)
    print(path)


if __name__ == "__main__":
    main()
