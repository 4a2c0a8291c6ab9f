#!/usr/bin/env python3
"""Generates bonito_amd/csrc/gemm_ktile_mfma.inc: ONE K-tile (64 deep) of the 256 x 256 linear-layer GEMM `gemm_w4_kernel` (gemm.hip)
as a hand-scheduled instruction stream for ONE wave per SIMD.

    wave tile 128 tokens (X rows, LDS "A ring", fragments FA) x 128 features (W rows, "B ring", FB), v_mfma_f32_32x32x16_f16 with the W
    fragment as the instruction's A input (features along the accumulator registers, tokens along the lanes):
    acc[i][j] += FB[j][ks] * FA[i][ks]^T,   i, j in 0..3 (32 x 32 tiles), ks in 0..3 (16 deep)  = 64 MFMAs = 2048 matrix-pipe cycles
    per k-step 8 fragment reads (ds_read_b128, 1 KiB each) for 16 MFMAs: 4 MFMAs per KiB of LDS traffic (the 64 x 128 wave tile of
    gemm_big_kernel on 16 x 16 x 32: 2.7) and half the operand-register reads per FLOP.

LDS: the A operand (X rows: streamed from HBM, every fetch pays memory latency) has a ring of THREE 32 KiB stages, the B operand
(W rows: L2 resident) two. What one stream instance does for K-tile g of a workgroup (A stage g % 3, B stage g % 2; the stage
offsets are scalar operands, the read addresses are made from them by eight v_add_u32 at the head of the stream):
    ks = 0      16 MFMAs on the fragments of ks 0 (read by the previous instance); the 8 reads of ks 1; D1: the wave's eight
                LDS-DMA pieces of the A operand of K-tile g + 2 (into A stage (g + 2) % 3, free since the barrier of K-tile g - 1)
    ks = 1, 2   16 MFMAs each; the reads of ks 2 / 3
    ks = 3      WAIT_AT MFMAs, then: lgkmcnt(0) (all my reads of K-tile g are back), vmcnt(8) (my pieces of K-tile g + 1 - A issued
                1.75 K-tiles ago, B one K-tile ago - have landed; only D1 of this instance is younger), s_barrier (everybody's have;
                nobody reads K-tile g any more); behind it the 8 reads of ks 0 of K-tile g + 1 and D2: my eight pieces of the B
                operand of K-tile g + 2 (into B stage g % 2)
One barrier per K-tile. With two stages per operand every K-tile waited for the A pieces issued ONE K-tile earlier, and the loop ran
at the memory latency (64 KiB in flight per CU / ~1.5 us = 43 GB/s per CU = ~1.5 us per K-tile on every shape, 2100 cycles of MFMA
in it); the third A stage doubles the lead of the operand that comes from HBM. Nothing in the stream waits for an instruction
issued fewer than ~6 MFMAs (190 cycles) earlier. Fragment registers are double buffered by k-step
parity; reads are waited for with COUNTED lgkmcnt (they return in order), so a late fragment never holds up the MFMAs in front of it.
`FIRST` instances (first K-tile of an output tile) start the accumulators from the inline constant 0 instead of reading them.

The stream ends with lgkmcnt(0): the compiler may touch the fragment registers behind the statement. It does NOT pad the
MFMA -> vector-ALU hazard at its end: gemm_w4_kernel puts 16 wait states in front of the epilogue's first accumulator read.

usage: gen_gemmstep.py [--wait-at n] [--read-at n] [--d1-every n] [--out path] [--name fn]
       gen_gemmstep.py --tile16 [--barrier-at n] [--read-every n] [--d2-every16 n] [--now16 n] [--stagger c] [--out path]
           -> gemm_ktile16_mfma.inc: the same K-tile on v_mfma_f32_16x16x32_f16 (build16(); the library's default since round 6)
"""
import argparse
import os


def build(wait_at=2, read_at=1, d1_at=0, d1_every=2, d2_every=1, first=False, name="gemm_ktile", strip=(), dma_flags="", stores=0, wide=True, st_at=3, st_every=2):
    lines = []          # instruction text
    counts = {"mfma": 0, "lds": 0, "vmem": 0, "salu": 0, "wait": 0, "nop": 0, "valu": 0}
    fifo = []           # fragment names with a read in flight, oldest first

    def emit(kind, text):
        # --strip (timing experiments, wrong results on purpose): drop the DMA, the fragment reads and / or the barrier from the stream
        if ("dma" in strip and (kind in ("vmem", "salu", "nop") or "vmcnt" in text)) or ("lds" in strip and kind == "lds") or \
                ("bar" in strip and text == "s_barrier"):
            return
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
        emit("vmem", "global_load_lds_dwordx4 %%[%s], %%[%s]%s" % (voff, sbase, (" " + dma_flags) if dma_flags else ""))

    def frag_reads(p, ra, rb):
        """the 8 reads of one k-step in the order the MFMAs need them (i-major MFMA order): A0 B0 B1 B2 B3 A1 A2 A3"""
        order = [("a", 0)] + [("b", j) for j in range(4)] + [("a", i) for i in range(1, 4)]
        return [("f%s%d%d" % (op, p, n), ra if op == "a" else rb, n * 4096) for op, n in order]

    # read addresses of this instance: per-lane bases + the (scalar) stage offsets
    head = [("ra%d" % k, "sa", "rab%d" % k) for k in (1, 2, 3)] + [("rb%d" % k, "sb", "rbb%d" % k) for k in (1, 2, 3)] + \
           [("rao", "san", "rab0"), ("rbo", "sbn", "rbb0")]
    head_at = {0: head[0:2], 1: head[2:6], 8: head[6:8]}      # behind MFMA m of k-step 0 (ra1 / rb1 are needed behind MFMA 1)

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
        st = []
        if ks == 3 and stores:
            # the parked output rows of the PREVIOUS tile (gemm.hip: deferred epilogue): right behind the barrier, so that they have a
            # whole K-tile to retire before the next counted vmcnt wait has to sit them out (loads and stores share the counter)
            st = [(min(15, wait_at + st_at + (n * st_every if stores <= 6 else n)), n) for n in range(stores)]
        m = 0
        for i in range(4):
            for j in range(4):
                if ks == 3 and m == wait_at:
                    fifo_was = list(fifo)
                    emit("wait", "s_waitcnt lgkmcnt(0)")
                    del fifo[:]
                    emit("wait", "s_waitcnt vmcnt(8)")
                    emit("wait", "s_barrier")
                    assert all(f[1] == str(p) or True for f in fifo_was)
                need("fa%d%d" % (p, i), "fb%d%d" % (p, j))
                c = "%%[c%d%d]" % (i, j)
                src = "0" if (first and ks == 0) else c
                # the B-ring operand (W rows: features) is the MFMA's A input, the A-ring operand (X rows: tokens) its B input: features
                # run along the accumulator registers, tokens along the lanes (gemm.hip: w4_epilogue)
                emit("mfma", "v_mfma_f32_32x32x16_f16 %s, %%[fb%d%d], %%[fa%d%d], %s" % (c, p, j, p, i, src))
                if ks == 0:
                    for dst, sreg, base in head_at.get(m, ()):
                        emit("valu", "v_add_u32_e32 %%[%s], %%[%s], %%[%s]" % (dst, sreg, base))
                # side instructions behind MFMA m of this k-step
                if m >= start and pending and (m - start) < 8:
                    read(*pending.pop(0))
                for at, args in d:
                    if at == m:
                        dma(*args)
                for at, n in st:
                    if at == m:
                        emit("salu", "s_mul_i32 %%[stt], %%[rowb], %%[k%d]" % n)
                        emit("vmem", "buffer_store_dwordx%d %%[pk%d], %%[stv], %%[srd], %%[stt] offen offset:%%[o%d]" % (4 if wide else 2, n, n))
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
        outs.append('[ra%d] "=&v"(ra[%d])' % (ks, ks))
        outs.append('[rb%d] "=&v"(rb[%d])' % (ks, ks))
    outs += ['[rao] "=&v"(ra[0])', '[rbo] "=&v"(rb[0])']
    for ks in range(4):
        ins.append('[rab%d] "v"(rab[%d])' % (ks, ks))
        ins.append('[rbb%d] "v"(rbb[%d])' % (ks, ks))
    ins += ['[sa] "s"(sa)', '[sb] "s"(sb)', '[san] "s"(san)', '[sbn] "s"(sbn)']
    for n in range(8):
        ins.append('[vd1_%d] "v"(vd1[%d])' % (n, n))
    for n in range(8):
        ins.append('[vd2_%d] "v"(vd2[%d])' % (n, n))
    ins += ['[sd1] "s"(sd1)', '[sd2] "s"(sd2)', '[md1] "s"(md1)', '[md2] "s"(md2)']
    if stores:
        outs.append('[stt] "=&s"(stt)')
        for n in range(stores):
            ins.append('[pk%d] "v"(park[(IDX0 + %d) %% W4_NPARK])' % (n, n))
            ins.append('[k%d] "n"(w4_store_row((IDX0 + %d) %% W4_NPARK))' % (n, n))
            ins.append('[o%d] "n"(w4_store_col((IDX0 + %d) %% W4_NPARK) * %d)' % (n, n, 128 if wide else 64))
        ins += ['[srd] "s"(srd)', '[stv] "v"(stv)', '[rowb] "s"(rowb)']
    stat = " ".join("%s=%d" % kv for kv in sorted(counts.items()))
    body = '"\n        "'.join(l + "\\n\\t" for l in lines[:-1])
    text = []
    text.append("// %s<%s>: wait_at=%d read_at=%d d1_at=%d d1_every=%d d2_every=%d stores=%d%s : %s" % (
        name, "FIRST" if first else "", wait_at, read_at, d1_at, d1_every, d2_every, stores, ("" if not stores else (" x16B" if wide else " x8B")), stat))
    fname = name + ("_first" if first else "") + (("_st%d%s" % (stores, "w" if wide else "n")) if stores else "")
    sig = ("__device__ __forceinline__ void %s(float16_t (&acc)[4][4], half8_t (&fa)[2][4], half8_t (&fb)[2][4], "
           "const unsigned (&rab)[4], const unsigned (&rbb)[4], unsigned sa, unsigned sb, unsigned san, unsigned sbn, const unsigned (&vd1)[8], "
           "const unsigned (&vd2)[8], const char* sd1, const char* sd2, unsigned md1, unsigned md2%s) {" % (
               fname, (", const %s (&park)[W4_NPARK], uint4_t srd, unsigned stv, unsigned rowb" % ("uint4_t" if wide else "uint2_t")) if stores else ""))
    if stores:
        text.append("template <int IDX0>")
    text.append(sig)
    text.append("    unsigned ra[4], rb[4];")
    if stores:
        text.append("    unsigned stt;")
    text.append('    asm volatile("' + body + '"\n        "' + lines[-1] + '"')
    text.append("        : " + ", ".join(outs))
    text.append("        : " + ", ".join(ins))
    text.append('        : "memory", "scc");')
    text.append("    (void)ra; (void)rb;" + (" (void)stt;" if stores else ""))
    text.append("}")
    return "\n".join(text) + "\n"


def build16(barrier_at=78, read_every=3, d1_at=0, d1_every=4, d2_every=4, first=False, name="gemm_ktile16", strip=(), dma_flags="", stores=0, wide=True,
            st_at=1, st_every=4, stagger=0, swiglu=False):
    """The same K-tile on v_mfma_f32_16x16x32_f16 (round 6: under the board's power cap a hand-scheduled LDS-fed 128 x 128 wave tile runs 14 %
    faster on the small tile - tools/gen_tile_probe.py, profiles/r06_mfma_tile_energy_lds.txt).

        acc[i][j] (float4) += W fragment FB[s][j] (16 features x 32 k, the instruction's A input) * X fragment FA[i]^T (16 tokens x 32 k),
        i, j in 0..7, s = 0, 1 (32-deep slabs) = 128 MFMAs = 2048 matrix-pipe cycles, MFMA m = 64 s + 8 i + j ("block" c = 8 s + i)

    Registers: the W fragments are double buffered by slab (FB[2][8]); the X fragments live in ONE ring of eight (FA[i], reused by slab 1
    as soon as block i of slab 0 has issued) - 96 fragment registers, where double buffering both operands (128) would not fit beside the parked
    output rows. Reads of one instance, all ds_read_b128, one every `read_every` MFMAs, the most urgent first:
        in front of the barrier   X(2 .. 7) of slab 0 (X(0), X(1) and the eight W fragments of slab 0 were read by the previous instance), the eight
                                  W fragments of slab 1, X(8 + k) into FA[k] once block k is through                      = 22 reads
        barrier in front of MFMA `barrier_at` (>= 72: X(15) can only be read once block 7 is through): lgkmcnt(0), vmcnt(8), s_barrier
        behind it                 of K-tile g + 1: X(0) -> FA[0], W(0 .. 7) of slab 0 -> FB[0], X(1) -> FA[1]                  = 10 reads
    DMA pieces (D1 from MFMA d1_at, D2 behind the barrier), the parked stores behind D2, and the five address additions as in build()."""
    lines = []
    counts = {"mfma": 0, "lds": 0, "vmem": 0, "salu": 0, "wait": 0, "nop": 0, "valu": 0}
    fifo = []

    def emit(kind, text):
        if ("dma" in strip and (kind in ("vmem", "salu", "nop") or "vmcnt" in text)) or ("lds" in strip and kind == "lds") or \
                ("bar" in strip and text == "s_barrier"):
            return
        counts[kind] += 1
        lines.append(text)

    def need(*frags):
        idx = max((fifo.index(f) for f in frags if f in fifo), default=-1)
        if idx >= 0:
            # (the counter has four bits: a wait for "at most 15 outstanding" where more would do is only stricter)
            emit("wait", "s_waitcnt lgkmcnt(%d)" % min(15, len(fifo) - 1 - idx))
            del fifo[:idx + 1]

    def dma(mbase, imm, voff, sbase):
        emit("salu", "s_add_u32 m0, %%[%s], 0x%x" % (mbase, imm))
        emit("nop", "s_nop 0")
        emit("vmem", "global_load_lds_dwordx4 %%[%s], %%[%s]%s" % (voff, sbase, (" " + dma_flags) if dma_flags else ""))

    # (earliest MFMA index behind which the read may issue, MFMA index that needs it, destination, address register, offset)
    pre = [(0, 8 * i, "fa%d" % i, "ra0", i * 2048) for i in range(2, 8)]
    pre += [(0, 64 + j, "fb1_%d" % j, "rb1", j * 2048) for j in range(8)]
    pre += [(8 * k + 7, 64 + 8 * k, "fa%d" % k, "ra1", k * 2048) for k in range(8)]
    post = [(71, 128, "fa0", "rao", 0)] + [(0, 128 + j, "fb0_%d" % j, "rbo", j * 2048) for j in range(8)] + [(79, 136, "fa1", "rao", 2048)]
    # (the addresses of K-tile g + 1 reuse the registers of ra0 / rb1: both are dead by MFMA 64 - asserted where they are overwritten)
    head_at = {0: [("ra0", "sa", "rab0"), ("rb1", "sb", "rbb1")], 1: [("ra1", "sa", "rab1")], 66: [("rao", "san", "rab0")], 67: [("rbo", "sbn", "rbb0")]}
    alias = {"rao": "ra0", "rbo": "rb1"}
    d1 = {d1_at + n * d1_every: ("md1", n * 1024, "vd1_%d" % n, "sd1") for n in range(8)}
    d2 = {barrier_at + n * d2_every: ("md2", n * 1024, "vd2_%d" % n, "sd2") for n in range(8)}
    st = {}
    if stores:
        st = {barrier_at + 7 * d2_every + st_at + n * st_every: n for n in range(stores)}
        assert max(st) < 128 and len(st) == stores
    assert 72 <= barrier_at and max(d2) < 128 and max(d1) < barrier_at

    for m in range(128):
        s, i, j = m >> 6, (m >> 3) & 7, m & 7
        if m == barrier_at:
            assert not pre, "reads of this K-tile left behind the barrier: %r" % (pre,)
            emit("wait", "s_waitcnt lgkmcnt(0)")
            del fifo[:]
            emit("wait", "s_waitcnt vmcnt(8)")
            emit("wait", "s_barrier")
            if stagger:
                # wave w leaves the barrier w * `stagger` cycles late and keeps that phase until the next one: the four waves of the workgroup
                # present their DMA instructions to the CU's one address path in turn instead of at once
                for bit in (0, 1):
                    emit("salu", "s_bitcmp1_b32 %%[wv], %d" % bit)
                    emit("salu", "s_cbranch_scc0 .Lstag%d_%%=" % bit)
                    left = stagger << bit
                    while left > 0:
                        n = min(16, left)
                        emit("nop", "s_nop %d" % (n - 1))
                        left -= n
                    lines.append(".Lstag%d_%%=:" % bit)
        need("fa%d" % i, "fb%d_%d" % (s, j))
        c = "%%[c%d_%d]" % (i, j)
        emit("mfma", "v_mfma_f32_16x16x32_f16 %s, %%[fb%d_%d], %%[fa%d], %s" % (c, s, j, i, "0" if (first and s == 0) else c))
        for dst, sreg, base in head_at.get(m, ()):
            assert dst not in alias or not any(r[3] == alias[dst] for r in pre), "%s still in use at MFMA %d" % (alias[dst], m)
            emit("valu", "v_add_u32_e32 %%[%s], %%[%s], %%[%s]" % (alias.get(dst, dst), sreg, base))
        if m % read_every == read_every - 1:
            queue = pre if m < barrier_at else post
            ready = [r for r in queue if r[0] <= m]
            if ready:
                r = min(ready, key=lambda r: r[1])
                queue.remove(r)
                assert r[1] > m + 4 or r[1] >= 128, "read of %s issued too late (MFMA %d needs it, slot %d)" % (r[2], r[1], m)
                emit("lds", "ds_read_b128 %%[%s], %%[%s] offset:%d" % (r[2], alias.get(r[3], r[3]), r[4]))
                fifo.append(r[2])
        if m in d1:
            dma(*d1[m])
        if m in d2:
            dma(*d2[m])
        if m in st:
            n = st[m]
            emit("salu", "s_mul_i32 %%[stt], %%[rowb], %%[k%d]" % n)
            emit("vmem", "buffer_store_dwordx%d %%[pk%d], %%[stv], %%[srd], %%[stt] offen offset:%%[o%d]" % (4 if wide else 2, n, n))
    assert not pre and not post, (pre, post)
    emit("wait", "s_waitcnt lgkmcnt(0)")

    outs, ins = [], []
    for i in range(8):
        for j in range(8):
            outs.append('[c%d_%d] "%s"(acc[%d][%d])' % (i, j, "=&a" if first else "+a", i, j))
    for i in range(8):
        outs.append('[fa%d] "%s"(fa[%d])' % (i, "+v" if i < 2 else "=&v", i))
    for j in range(8):
        outs.append('[fb0_%d] "+v"(fb[0][%d])' % (j, j))
    for j in range(8):
        outs.append('[fb1_%d] "=&v"(fb[1][%d])' % (j, j))
    outs += ['[ra0] "=&v"(ra[0])', '[ra1] "=&v"(ra[1])', '[rb1] "=&v"(ra[2])']
    ins += ['[rab0] "v"(rab[0])', '[rab1] "v"(rab[1])', '[rbb0] "v"(rbb[0])', '[rbb1] "v"(rbb[1])']
    ins += ['[sa] "s"(sa)', '[sb] "s"(sb)', '[san] "s"(san)', '[sbn] "s"(sbn)']
    for n in range(8):
        ins.append('[vd1_%d] "v"(vd1[%d])' % (n, n))
    for n in range(8):
        ins.append('[vd2_%d] "v"(vd2[%d])' % (n, n))
    ins += ['[sd1] "s"(sd1)', '[sd2] "s"(sd2)', '[md1] "s"(md1)', '[md2] "s"(md2)', '[wv] "s"(wv)']
    if stores:
        outs.append('[stt] "=&s"(stt)')
        for n in range(stores):
            if swiglu:      # gemm.hip, w4_epilogue16g: 16 whole 128-byte output rows per lane and tile, no column offset
                ins.append('[pk%d] "v"(park[(IDX0 + %d) %% W4_NPARKG])' % (n, n))
                ins.append('[k%d] "n"(w4_store_rowg((IDX0 + %d) %% W4_NPARKG))' % (n, n))
                ins.append('[o%d] "n"(0)' % n)
                continue
            ins.append('[pk%d] "v"(park[(IDX0 + %d) %% W4_NPARK16])' % (n, n))
            ins.append('[k%d] "n"(w4_store_row16((IDX0 + %d) %% W4_NPARK16))' % (n, n))
            ins.append('[o%d] "n"(w4_store_col16((IDX0 + %d) %% W4_NPARK16) * %d)' % (n, n, 128 if wide else 64))
        ins += ['[srd] "s"(srd)', '[stv] "v"(stv)', '[rowb] "s"(rowb)']
    stat = " ".join("%s=%d" % kv for kv in sorted(counts.items()))
    body = '"\n        "'.join(l + "\\n\\t" for l in lines[:-1])
    text = []
    text.append("// %s<%s>: barrier_at=%d read_every=%d d1_at=%d d1_every=%d d2_every=%d stagger=%d stores=%d%s : %s" % (
        name, "FIRST" if first else "", barrier_at, read_every, d1_at, d1_every, d2_every, stagger, stores, ("" if not stores else (" x16B" if wide else " x8B")), stat))
    fname = name + ("_first" if first else "") + (("_st%d%s" % (stores, "g" if swiglu else "w" if wide else "n")) if stores else "")
    sig = ("__device__ __forceinline__ void %s(float4_t (&acc)[8][8], half8_t (&fa)[8], half8_t (&fb)[2][8], "
           "const unsigned (&rab)[2], const unsigned (&rbb)[2], unsigned sa, unsigned sb, unsigned san, unsigned sbn, const unsigned (&vd1)[8], "
           "const unsigned (&vd2)[8], const char* sd1, const char* sd2, unsigned md1, unsigned md2, unsigned wv%s) {" % (
               fname, (", const %s (&park)[%s], uint4_t srd, unsigned stv, unsigned rowb" % ("uint4_t" if wide else "uint2_t", "W4_NPARKG" if swiglu else "W4_NPARK16")) if stores else ""))
    if stores:
        text.append("template <int IDX0>")
    text.append(sig)
    text.append("    unsigned ra[3];")
    if stores:
        text.append("    unsigned stt;")
    text.append('    asm volatile("' + body + '"\n        "' + lines[-1] + '"')
    text.append("        : " + ", ".join(outs))
    text.append("        : " + ", ".join(ins))
    text.append('        : "memory", "scc");')
    text.append("    (void)ra;" + (" (void)stt;" if stores else ""))
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
    ap.add_argument("--no-stores", action="store_true", help="omit the variants with store slots")
    ap.add_argument("--dma-flags", default="", help="cache policy bits of the LDS-DMA instructions: nt | sc0 | sc1 | sc0 sc1")
    ap.add_argument("--strip", default="", help="comma list of dma,lds,bar: timing experiments only (wrong results)")
    ap.add_argument("--out", default=None)
    ap.add_argument("--tile16", action="store_true", help="write gemm_ktile16_mfma.inc: the same K-tile on 16x16x32 MFMAs (build16)")
    ap.add_argument("--barrier-at", type=int, default=78)
    ap.add_argument("--read-every", type=int, default=3)
    ap.add_argument("--st-at", type=int, default=1)
    ap.add_argument("--st-every", type=int, default=4)
    ap.add_argument("--stagger", type=int, default=0, help="--tile16: cycles by which wave w leaves the K-tile's barrier late (times w)")
    ap.add_argument("--d2-every16", type=int, default=4, help="--tile16: MFMAs between the DMA pieces behind the barrier")
    ap.add_argument("--nowg", type=int, default=8, help="--tile16: of the 16 stores of a SwiGLU tile, those the epilogue issues itself")
    ap.add_argument("--now16", type=int, default=24, help="--tile16: output rows (of 32 per lane) the epilogue stores itself; the rest is parked")
    a = ap.parse_args()
    if a.tile16:
        kw = dict(barrier_at=a.barrier_at, read_every=a.read_every, d1_at=a.d1_at, d1_every=2 * a.d1_every, d2_every=a.d2_every16, stagger=a.stagger,
                  strip=tuple(x for x in a.strip.split(",") if x), dma_flags=a.dma_flags)
        text = ("// GENERATED by tools/gen_gemmstep.py --tile16 - do not edit. One K-tile (64 deep) of gemm_w4_kernel's 128 x 128 wave tile on\n"
                "// v_mfma_f32_16x16x32_f16: 128 MFMAs, 32 fragment reads, the wave's 16 LDS-DMA pieces of the K-tiles ahead, one barrier. See build16()\n"
                "// in the generator for the schedule.\n"
                "// Deferred epilogue as in gemm_ktile_mfma.inc, with this stream's own split: the first W4_NOW16 of a tile's 32 rows per lane are stored by\n"
                "// the epilogue, W4_NPARK16 are parked (96 fragment registers instead of 64 live through the K loop: with 20 parked rows of\n"
                "// 16 bytes the kernel spilled - profiles/r06_gemm_tile16.txt).\n"
                "#ifndef W4_NOW16_VALUE\n#define W4_NOW16_VALUE %d\n#endif\n"
                "constexpr int W4_NOW16 = W4_NOW16_VALUE, W4_NPARK16 = 32 - W4_NOW16;\n"
                "constexpr int w4_store_row16(int idx) { return 32 * ((idx + W4_NOW16) >> 3) + 8 * (idx & 3); }\n"
                "constexpr int w4_store_col16(int idx) { return ((idx + W4_NOW16) >> 2) & 1; }\n"
                "// SwiGLU outputs (w4_epilogue16g): a tile is 16 stores per lane (unit u = 4 blk + rr: token row 32 blk + 8 rr of the wave, the whole\n"
                "// 128-byte output row); the first W4_NOWG go out in the epilogue, the others are parked.\n"
                "#ifndef W4_NOWG_VALUE\n#define W4_NOWG_VALUE %d\n#endif\n"
                "constexpr int W4_NOWG = W4_NOWG_VALUE, W4_NPARKG = 16 - W4_NOWG;\n"
                "constexpr int w4_store_rowg(int idx) { return 32 * ((idx + W4_NOWG) >> 2) + 8 * ((idx + W4_NOWG) & 3); }\n" % (a.now16, a.nowg))
        text += build16(first=False, **kw) + "\n" + build16(first=True, **kw)
        for wide in (True, False):
            for first in (False, True):
                text += "\n" + build16(first=first, stores=4, wide=wide, st_at=a.st_at, st_every=a.st_every, **kw)
        for first in (False, True):
            text += "\n" + build16(first=first, stores=4, wide=True, swiglu=True, st_at=a.st_at, st_every=a.st_every, **kw)
        path = a.out or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bonito_amd", "csrc", "gemm_ktile16_mfma.inc")
        with open(path, "w") as f:
            f.write(text)
        print(path)
        return
    kw = dict(wait_at=a.wait_at, read_at=a.read_at, d1_at=a.d1_at, d1_every=a.d1_every, d2_every=a.d2_every, name=a.name,
              strip=tuple(x for x in a.strip.split(",") if x), dma_flags=a.dma_flags)
    head = ("// GENERATED by tools/gen_gemmstep.py - do not edit. One K-tile (64 deep) of gemm_w4_kernel's 128 x 128 wave tile as one\n"
            "// instruction stream: 64 v_mfma_f32_32x32x16_f16, the 32 fragment reads of the next k-steps, the wave's 16 LDS-DMA pieces of the\n"
            "// K-tiles ahead, one barrier. See the generator's docstring for the schedule and gemm.hip for the operands.\n")
    text = head + build(first=False, **kw) + "\n" + build(first=True, **kw)
    if not a.no_stores:
        # instances that also issue `stores` parked output rows of the previous tile: row index / feature-pair of parked row idx
        text += ("\n// Deferred epilogue: of a tile's 32 output rows per lane (block b = (token tile i = b >> 1, feature pair P = b & 1), rr < 4: token row\n"
                 "// 32 i + 8 rr (+ lane >> 3) of the wave, byte column 128 P (64 P for SwiGLU outputs) (+ lane & 7 pieces)) the first\n"
                 "// W4_NOW are stored by the epilogue itself, the other W4_NPARK are parked in registers (index idx = 4 b + rr - W4_NOW) and\n"
                 "// issued by the K-tile instances of the next tile. (All 32 parked: 128 + 88 registers of operands leave the compiler ~25 -\n"
                 "// it spilled, and every reload in front of an instance is an s_waitcnt vmcnt(0).)\n"
                 "constexpr int W4_NOW = 12, W4_NPARK = 32 - W4_NOW;\n"
                 "constexpr int w4_store_row(int idx) { return 32 * ((idx + W4_NOW) >> 3) + 8 * (idx & 3); }\n"
                 "constexpr int w4_store_col(int idx) { return ((idx + W4_NOW) >> 2) & 1; }\n")
        for stores in (4,):
            for wide in (True, False):
                for first in (False, True):
                    text += "\n" + build(first=first, stores=stores, wide=wide, **kw)
    path = a.out or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bonito_amd", "csrc", "gemm_ktile_mfma.inc")
    with open(path, "w") as f:
        f.write(text)
    print(path)


if __name__ == "__main__":
    main()
