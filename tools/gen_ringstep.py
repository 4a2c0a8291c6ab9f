#!/usr/bin/env python3
"""Generates bonito_amd/csrc/ringstep3_mfma.inc: ONE time step of a ring of the paired recurrent kernel (H = 384: 12 k-steps, three M
tiles = three cells per lane) as a hand-scheduled instruction stream:

    R[m][ks]:  acc[m] += W_hh[m][ks] * h_{t-1}[ks]       36 MFMAs (recurrent part of step t; acc[m] enters holding bias + W_ih x_t)
    X[m][ks]:  xa[m]  += W_ih[m][ks] * x_{t+1}[ks]       36 MFMAs (input projection of step t+1; xa[m] enters holding the bias)
    gates(m):  lstm_cell() of cell m on acc[m]           35 vector instructions each (7 transcendental)

The recurrent part runs TILE-major: once the twelve R[0][*] have issued, cell 0's gate arithmetic has matrix work to hide behind
(R[1][*] and part of X), cell 1's hides behind R[2][*] and X, cell 2's behind the rest of X - the matrix pipe never waits for the
vector ALU and the vector ALU never runs alone (the round-2 kernel issued the 36 recurrent MFMAs bare, then wove the three cells
into the 36 input-projection MFMAs: 1.1 k + 1.25 k cycles per ring step for 1.15 k cycles of MFMA).

B fragments are read from LDS inside the stream (ds_read_b128 into a few rotating registers, explicit lgkmcnt waits): nothing but the
fragments in flight is live. The accumulation order of every accumulator is that of lstm_layer_wgx_kernel (ks ascending), the gate
arithmetic is lstm_cell()'s operation for operation => the same bits (tests/test_gpu_encoder.py::test_paired_rings_*).

One asm statement (no operand limit on this target), so the compiler cannot place anything inside the stream. A cell's four
pre-activations are the four registers of its accumulator tuple; an asm operand cannot name a sub-register, so the three recurrent
accumulators are PHYSICAL registers (v[PBASE .. PBASE+11], "=&{v[a:b]}" constraints) and the gate arithmetic names their components
literally. They are pure outputs: the first recurrent MFMA of a tile reads the projected input (xacc[m]) as its C operand and writes
the tuple (D != C), the first input-projection MFMA reads the bias as C and overwrites xacc[m] - no register copies at either end.
(A first version cut the stream into four statements and let the compiler rename sub-registers at the cuts: under the register
pressure of the real kernel it moved accumulator tuples that MFMAs were still writing. tools/audit_ringstep.py is the check that
found it; it stays as a guard for any multi-statement stream.)

Hazards placed here (the compiler pads nothing inside an asm statement):
  * MFMA D -> vector ALU read: >= 12 wait states (8-pass XDL) behind the last R[m][*] before gates(m) starts, and behind the last
    MFMA at the end of the stream (the compiler reads the outputs right behind the statement);
  * transcendental result -> consumer: at least one instruction in between (gfx940+ trans forwarding hazard);
  * v_cmp (vcc) -> v_cndmask: at least one instruction in between.

usage: gen_ringstep.py --preset plain|paired|unrolled|rec    (the four includes of the library)
       gen_ringstep.py [--xdist a,b,c,d] [--hdepth n] [--xdepth n] [--hf-live] [--polls-at n --xdma-at n --validate-at n --spread n]
                       [--name fn] [--out path]         (experiments: tools/stream_bench.py, tools/gpu_variants.sh)
"""
import argparse
import os

L_NEG = "0xbfb8aa3b"      # -log2(e)
L_POS = "0x3fb8aa3b"      # +log2(e)
L_M2POS = "0xc038aa3b"    # -2 log2(e)
PACKED = False           # --packed: packed fp32 instructions + fused constants in the gate arithmetic (30 instead of 35 per cell)
NKS = 12


class Ins:
    def __init__(self, kind, text, reads=(), writes=(), **meta):
        self.kind = kind            # mfma | valu | trans | cmp | sel | lds | wait | nop
        self.text = text
        self.reads = tuple(reads)
        self.writes = tuple(writes)
        self.meta = meta

    def states(self):
        if self.kind == "nop":
            return self.meta["n"] + 1
        return 1


def nop(n):
    return Ins("nop", "s_nop %d" % n, n=n)


POLL_FLAGS = "sc0 sc1"  # cache policy of the poll DMAs (experiments: --poll-flags)
IMM = False          # --imm: LDS tile offsets / M0 tile offsets are template constants ("n" operands) added to loop-invariant bases
ALL_AGPR = False     # timing experiments: every weight operand in the accumulator half
PBASE = 232          # v[PBASE .. PBASE+11]: the three recurrent accumulator tuples (physical registers, see the module docstring)


def gates(m, nan_check=False):
    """lstm_cell() of cell m, one dependency-ordered list. g{m}{i}: the four pre-activations = the registers of the accumulator tuple
    (literal names), reused as temporaries; operands e{m}, c{m} (cell state), hi25 / hi12."""
    g0, g1, g2, g3 = ("g%d%d" % (m, i) for i in range(4))
    e, cs = "e%d" % m, "c%d" % m

    def R(n):
        if n[0] == "g":
            return "v%d" % (PBASE + 4 * int(n[1]) + int(n[2]))
        return "%[" + n + "]"

    def v(kind, fmt, dst, *src):
        names = {"d": R(dst)}
        for i, s in enumerate(src):
            names["s%d" % i] = R(s)
        return Ins(kind, fmt.format(**names), reads=src, writes=(dst,), cell=m)

    seq = []
    if nan_check:
        # an element of h_{t-1} that had not arrived is the exchange sentinel 0xFFFF = an fp16 NaN: it makes every pre-activation of
        # its chunk column NaN. The clamps below would swallow that (v_med3 returns a number), so fold the four pre-activations into
        # one running fma first: it is NaN afterwards iff one of them was (finite values cannot overflow it)
        seq.append(v("valu", "v_fma_f32 {d}, {s0}, {s1}, {s2}", "nan", g0, g1, "nan"))
        seq.append(v("valu", "v_fma_f32 {d}, {s0}, {s1}, {s2}", "nan", g2, g3, "nan"))
    for g, hi in ((g0, "hi25"), (g1, "hi25"), (g2, "hi12"), (g3, "hi25")):
        seq.append(v("valu", "v_med3_f32 {d}, {s0}, {s1}, -{s1}", g, g, hi))

    def pk(fmt, lo, hi, const):          # one packed fp32 instruction on the register pair (lo, hi) with a 64-bit scalar constant pair
        pair = "v[%d:%d]" % (PBASE + 4 * m + int(lo[2]), PBASE + 4 * m + int(hi[2]))
        return Ins("valu", fmt % (pair, pair, "%[" + const + "]"), reads=(lo, hi, const), writes=(lo, hi), cell=m)

    if PACKED:
        # v_pk_mul_f32 / v_pk_add_f32 round each half like the scalar instruction; (-2 x) * log2e == x * (-2 log2e) bit for bit (the
        # factor 2 is exact, |x| <= 12.5): five multiplications and two additions become three packed instructions
        seq.append(pk("v_pk_mul_f32 %s, %s, %s", g0, g1, "kneg2"))            # (-log2e, -log2e)
        seq.append(pk("v_pk_mul_f32 %s, %s, %s", g2, g3, "kc2n"))             # (-2 log2e, -log2e)
    else:
        seq.append(v("valu", "v_mul_f32 {d}, " + L_NEG + ", {s0}", g0, g0))
        seq.append(v("valu", "v_mul_f32 {d}, " + L_NEG + ", {s0}", g1, g1))
        seq.append(v("valu", "v_mul_f32 {d}, -2.0, {s0}", g2, g2))
        seq.append(v("valu", "v_mul_f32 {d}, " + L_NEG + ", {s0}", g3, g3))
        seq.append(v("valu", "v_mul_f32 {d}, " + L_POS + ", {s0}", g2, g2))
    for g in (g0, g1, g2, g3):
        seq.append(v("trans", "v_exp_f32 {d}, {s0}", g, g))
    if PACKED:
        seq.append(pk("v_pk_add_f32 %s, %s, %s", g0, g1, "kone2"))            # Di = 1 + ei, Df = 1 + ef
    else:
        seq.append(v("valu", "v_add_f32 {d}, 1.0, {s0}", g0, g0))                 # Di = 1 + ei
        seq.append(v("valu", "v_add_f32 {d}, 1.0, {s0}", g1, g1))                 # Df = 1 + ef
    seq.append(v("valu", "v_sub_f32 {d}, 1.0, {s0}", e, g2))                  # 1 - eg
    seq.append(v("valu", "v_add_f32 {d}, 1.0, {s0}", g2, g2))                 # Dg = 1 + eg
    seq.append(v("valu", "v_mul_f32 {d}, {s0}, {s1}", g0, g0, g2))            # didg
    seq.append(v("valu", "v_mul_f32 {d}, {s0}, {s1}", e, e, g1))              # (1 - eg) df
    seq.append(v("valu", "v_mul_f32 {d}, {s0}, {s1}", g2, g1, g0))            # df didg
    seq.append(v("valu", "v_fma_f32 {d}, {s0}, {s1}, {s2}", e, cs, g0, e))    # num
    seq.append(v("trans", "v_rcp_f32 {d}, {s0}", g2, g2))
    seq.append(v("valu", "v_add_f32 {d}, 1.0, {s0}", g3, g3))                 # Do = 1 + eo (independent: sits behind the rcp)
    seq.append(v("valu", "v_mul_f32 {d}, {s0}, {s1}", cs, e, g2))             # c'
    seq.append(v("valu", "v_med3_f32 {d}, {s0}, {s1}, -{s1}", g0, cs, "hi12"))
    if PACKED:
        seq.append(v("valu", "v_mul_f32 {d}, " + L_M2POS + ", {s0}", g0, g0))
    else:
        seq.append(v("valu", "v_mul_f32 {d}, -2.0, {s0}", g0, g0))
        seq.append(v("valu", "v_mul_f32 {d}, " + L_POS + ", {s0}", g0, g0))
    seq.append(v("trans", "v_exp_f32 {d}, {s0}", g0, g0))                     # ec
    seq.append(v("valu", "v_add_f32 {d}, 1.0, {s0}", g1, g0))
    seq.append(v("valu", "v_sub_f32 {d}, 1.0, {s0}", g0, g0))
    seq.append(v("valu", "v_mul_f32 {d}, {s0}, {s1}", g1, g1, g3))
    seq.append(v("trans", "v_rcp_f32 {d}, {s0}", g1, g1))
    seq.append(v("valu", "v_mul_f32 {d}, {s0}, {s1}", g0, g0, g1))            # hv
    seq.append(Ins("cmp", "v_cmp_le_f32_e64 vcc, |%s|, 1.0" % R(g0), reads=(g0,), writes=("vcc",), cell=m))
    seq.append(Ins("sel", "v_cndmask_b32_e32 %s, 0, %s, vcc" % (R(g0), R(g0)), reads=(g0, "vcc"), writes=(g0,), cell=m))
    assert len(seq) == (30 if PACKED else 35) + (2 if nan_check else 0)
    return seq


def merge_even(a, b):
    """b spread evenly through a (both keep their order)."""
    if not b:
        return list(a)
    if not a:
        return list(b)
    out, j = [], 0
    n, m = len(a), len(b)
    for i, x in enumerate(a):
        out.append(x)
        while j < m and (j + 1) * n <= (i + 1) * m:
            out.append(b[j]); j += 1
    out += b[j:]
    return out


def weave(mf, va, lead_m):
    """va spread evenly over the MFMAs behind the first lead_m of them; leftovers trail."""
    out = list(mf[:lead_m])
    rest = mf[lead_m:]
    if not rest:
        return out + list(va)
    n, m = len(rest), len(va)
    j = 0
    for i, x in enumerate(rest):
        out.append(x)
        want = ((i + 1) * m + n - 1) // n if i + 1 < n else m
        while j < want:
            out.append(va[j]); j += 1
    return out


def publish_ops():
    """h_t of the lane's three cells -> fp16 -> the wave's LDS staging row (2-byte writes) -> read back as the 8 bytes this lane moves (one
    wave: LDS operations complete in issue order) -> exchange slot, re-arm of the slot of h_{t+2}, layer output row. Plain stores: the
    unrolled main loop only runs where the ring sits on one XCD."""
    ops = [Ins("valu", "v_cvt_f16_f32_e32 %%[hh%d], v%d" % (m, PBASE + 4 * m), writes=("hh%d" % m,)) for m in range(3)]
    ops += [Ins("lds", "ds_write_b16 %%[sga], %%[hh%d]%s" % (m, " offset:%d" % (2 * m) if m else ""), reads=("sga", "hh%d" % m), frag=("W", 0, m)) for m in range(3)]
    ops.append(Ins("lds", "ds_read_b64 %[pk], %[rda]", reads=("rda",), writes=("pk",), frag=("P", 0, 0)))
    ops.append(Ins("vmem", "global_store_dwordx2 %[vmy], %[pk], %[exs]", reads=("vmy", "pk", "exs"), needs=[("P", 0, 0)]))
    ops.append(Ins("vmem", "global_store_dwordx2 %[vmy], %[ones], %[exa]", reads=("vmy", "ones", "exa")))
    ops.append(Ins("vmem", "global_store_dwordx2 %[vh], %[pk], %[hrow]", reads=("vh", "pk", "hrow")))
    return ops


VBASE = 220          # v[VBASE .. VBASE+11]: the three fragments of the other ring's h tile read back for validation (physical: the OR tree names their dwords)


def build(xdist, hdepth, xdepth, hpool, xpool, hf_live, lead, wgroup=1, nan_check=False, polls_at=-1, xdma_at=-1, validate_at=-1, spread=0, publish=False, tail=0):
    # xdist all zero: the recurrent half alone (single-ring kernel: the input projection of the next step runs BEHIND the publish there,
    # it is what fills the hand-off's round trip)
    assert (sum(xdist) == 3 * NKS or sum(xdist) == 0) and len(xdist) == 4
    xs = [(ks, m) for ks in range(NKS) for m in range(3)]
    xi = 0
    spine_by_phase = []
    for p in range(4):
        # the first recurrent MFMA of every tile goes out at the very start: it reads xacc[m] (as C) before the input projection
        # of the next step starts to overwrite it
        R = ([("R", m, 0) for m in range(3)] if p == 0 else []) + ([("R", p, ks) for ks in range(1, NKS)] if p < 3 else [])
        X = [("X", m, ks) for ks, m in xs[xi:xi + xdist[p]]]
        xi += xdist[p]
        spine_by_phase.append(merge_even(R, X) if len(R) >= len(X) else merge_even(X, R))

    def mfma_ins(tag):
        kind, m, ks = tag
        if kind == "R":
            acc, w, frag = "P%d" % m, "wh%d_%d" % (m, ks), ("H", 0 if hf_live else m, ks)
        else:
            acc, w, frag = "xa%d" % m, "wi%d_%d" % (m, ks), ("X", 0, ks)
        return Ins("mfma", None, reads=(w,) if kind == "R" else (acc, w), writes=() if kind == "R" else (acc,), acc=acc, w=w, frag=frag, tag=tag)

    # ---- weave the cells into the phases -----------------------------------------------------------------------------
    seq = []
    for p in range(4):
        mf = [mfma_ins(t) for t in spine_by_phase[p]]
        va = gates(p - 1, nan_check) if p >= 1 else []
        if p == 3 and tail:
            # --tail K: cell 2's gate arithmetic over all but the last K MFMAs, the publish of h_t (fp16 -> LDS transpose -> three plain
            # stores) over the last K: the peers see h_t ~150 cycles earlier and the stream's MFMA-less end shrinks to the validation
            assert publish and 6 <= tail < len(mf)
            ops, mb = publish_ops(), mf[-tail:]
            # conversions + transpose right behind the cell, the stores behind the last two MFMAs: the LDS round trip of the transpose
            # (the stores wait for it) has the MFMAs in between to hide behind
            seq += weave(mf[:-tail], va, lead) + weave(mb[:3], ops[:7], 0) + mb[3:-2] + [mb[-2], ops[7], ops[8], mb[-1], ops[9]]
        else:
            seq += weave(mf, va, lead if p >= 1 else 0)

    # ---- fragment reads: `depth` MFMAs ahead of the first use, into rotating registers ---------------------------------
    mf_pos = [i for i, x in enumerate(seq) if x.kind == "mfma"]
    frags, first, last = [], {}, {}
    for n, i in enumerate(mf_pos):
        f = seq[i].meta["frag"]
        if f not in first:
            first[f] = n; frags.append(f)
        last[f] = n
    pools = {"H": [], "X": []}
    reg_of, issue_after = {}, {}
    for f in frags:
        pool = pools[f[0]]
        size = (NKS if hf_live else hpool) if f[0] == "H" else xpool
        depth = hdepth if f[0] == "H" else xdepth
        j = len(pool)
        reg_of[f] = ("ht%d" if f[0] == "H" else "xt%d") % (j % size)
        free_after = last[pool[j - size]] if j >= size else -1
        issue_after[f] = max(first[f] - depth - 1, free_after, -1)
        assert issue_after[f] < first[f]
        pool.append(f)
    reads_at = {}
    for f in frags:
        reads_at.setdefault(issue_after[f], []).append(f)

    def lds_ins(f):
        base, off = ("hb", f[2] * 1024) if f[0] == "H" else ("xb", f[2] * 1024)
        imm = ("%%[%s]+" % ("hoff" if f[0] == "H" else "xoff")) if IMM else ""
        return Ins("lds", "ds_read_b128 %%[%s], %%[%s] offset:%s%d" % (reg_of[f], base, imm, off), reads=(base,) + ((("hoff" if f[0] == "H" else "xoff"),) if IMM else ()),
                   writes=(reg_of[f],), frag=f)

    out = [lds_ins(f) for f in reads_at.get(-1, [])]
    n = -1
    for x in seq:
        out.append(x)
        if x.kind == "mfma":
            n += 1
            out += [lds_ins(f) for f in reads_at.get(n, [])]
    seq = out
    def tup(m):
        return "v[%d:%d]" % (PBASE + 4 * m, PBASE + 4 * m + 3)

    for x in seq:
        if x.kind == "mfma":
            b = reg_of[x.meta["frag"]]
            x.reads = x.reads + (b,)
            kind, m, ks = x.meta["tag"]
            if kind == "R":
                d, c = tup(m), ("%%[xa%d]" % m if ks == 0 else tup(m))
                if ks == 0:
                    x.reads = x.reads + ("xa%d" % m,)
            else:
                d, c = "%%[xa%d]" % m, ("%%[bi%d]" % m if ks == 0 else "%%[xa%d]" % m)
                if ks == 0:
                    x.reads = tuple(r for r in x.reads if r != "xa%d" % m) + ("bi%d" % m,)
            x.text = "v_mfma_f32_16x16x32_f16 %s, %%[%s], %%[%s], %s" % (d, x.meta["w"], b, c)
    # the first input-projection MFMA of a tile overwrites xacc[m]: the recurrent MFMA that reads it must have issued
    pos = {x.meta["tag"]: i for i, x in enumerate(seq) if x.kind == "mfma"}
    for m in range(3):
        assert ("X", m, 0) not in pos or pos[("R", m, 0)] < pos[("X", m, 0)]

    # ---- vector-memory work of the section issued from inside the stream, behind given MFMAs -------------------------------------------
    def after_mfma(seq, at, extra):
        out, n = [], -1
        for x in seq:
            out.append(x)
            if x.kind == "mfma":
                n += 1
                if n == at:
                    out += extra
        assert n >= at
        return out

    # (`spread`: MFMAs between two DMA instructions of a group - a burst of three costs the issuing wave ~150 cycles in one piece and
    #  skews the four waves of the workgroup against each other in front of the next barrier)
    if polls_at >= 0:            # the other ring's first poll round: three LDS-DMA instructions (1 KiB each) into its h tile
        for k in range(3):
            if IMM:
                m0w = Ins("salu", "s_add_u32 m0, %%[pm0], %%[pmo]+0x%x" % (0x1000 * k), reads=("pm0", "pmo"))
            else:
                m0w = Ins("salu", "s_mov_b32 m0, %[pm0]" if k == 0 else "s_add_u32 m0, %%[pm0], 0x%x" % (0x1000 * k), reads=("pm0",))
            ex = [m0w, nop(0),
                  Ins("vmem", "global_load_lds_dwordx4 %%[vp%d], %%[exo] %s" % (k, POLL_FLAGS), reads=("vp%d" % k, "exo"))]
            seq = after_mfma(seq, polls_at + k * spread, ex)
    if xdma_at >= 0:             # this ring's share of x_{t+2}: three LDS-DMA instructions into the x slot the previous step consumed
        assert xdma_at >= polls_at + 2 * spread
        for k in range(3):
            if IMM:
                ex = [Ins("salu", "s_add_u32 m0, %%[xm0], %%[xmo]+0x%x" % (0x1000 * k), reads=("xm0", "xmo"))]
            else:
                ex = [Ins("salu", "s_mov_b32 m0, %[xm0]" if k == 0 else "s_add_u32 m0, %%[xm0], 0x%x" % (0x1000 * k), reads=("xm0",))]
            if k:
                ex.append(Ins("valu", "v_add_u32_e32 %%[xv%d], 0x%x, %%[vx]" % (k, 0x100 * k), reads=("vx",), writes=("xv%d" % k,)))
            else:
                ex.append(nop(0))
            ex.append(Ins("vmem", "global_load_lds_dwordx4 %%[%s], %%[xsrc]" % ("xv%d" % k if k else "vx"), reads=(("xv%d" % k if k else "vx"), "xsrc")))
            seq = after_mfma(seq, xdma_at + k * spread, ex)
    if validate_at >= 0:         # my quarter of the other ring's h tile, read back: behind its polls only the x-stream DMAs above
        assert polls_at >= 0 and xdma_at >= polls_at and validate_at > xdma_at + 2 * spread
        vr = ["v[%d:%d]" % (VBASE + 4 * k, VBASE + 4 * k + 3) for k in range(3)]
        ex = [Ins("wait", "s_waitcnt vmcnt(3)", vmwait=True)]       # (the count is resolved below: the vector-memory operations younger than the polls)
        ex += [Ins("lds", "ds_read_b128 %s, %%[hbo] offset:%s%d" % (vr[k], "%[voff]+" if IMM else "", 4096 * k), reads=("hbo",) + (("voff",) if IMM else ()),
                   frag=("V", 0, k)) for k in range(3)]
        seq = after_mfma(seq, validate_at, ex)
    # ---- waits: LDS returns in order; wait for exactly as many as were issued behind the one needed ----------------------
    out, issued, done = [], [], -1
    for x in seq:
        if x.kind == "lds":
            issued.append(x.meta["frag"])
        elif x.kind == "mfma" or x.meta.get("needs"):
            needs = [x.meta["frag"]] if x.kind == "mfma" else x.meta["needs"]
            r = max(i for i, f in enumerate(issued) if f in needs)
            if r > done:
                if x.kind == "mfma":
                    r = min(r + wgroup - 1, len(issued) - 1)      # wgroup > 1: one wait covers the next fragments too (fewer issue slots)
                out.append(Ins("wait", "s_waitcnt lgkmcnt(%d)" % (len(issued) - 1 - r)))
                done = r
        out.append(x)
    seq = out
    # the validation's vmcnt: everything up to the polls must be home, i.e. all but the vector-memory operations issued behind them
    for i, x in enumerate(seq):
        if x.meta.get("vmwait"):
            last_poll = max(j for j, y in enumerate(seq[:i]) if y.kind == "vmem" and "exo" in y.reads)
            x.text = "s_waitcnt vmcnt(%d)" % sum(1 for y in seq[last_poll + 1:i] if y.kind == "vmem")

    if nan_check:
        seq = [Ins("valu", "v_mov_b32 %[nan], 0", writes=("nan",))] + seq

    # ---- hazards -----------------------------------------------------------------------------------------------------
    out = []
    last_r = {}                      # cell -> index in out of its last R MFMA
    started = set()
    for x in seq:
        cell = x.meta.get("cell")
        if cell is not None and cell not in started:
            started.add(cell)
            gap = sum(y.states() for y in out[last_r[cell] + 1:])
            if gap < 12:
                out.append(nop(12 - gap - 1))
        if out and out[-1].kind == "trans" and set(out[-1].writes) & set(x.reads):
            out.append(nop(0))
        if x.kind == "sel" and out[-1].kind == "cmp":
            out.append(nop(0))
        out.append(x)
        if x.kind == "mfma" and x.meta["tag"][0] == "R":
            last_r[x.meta["tag"][1]] = len(out) - 1
    last_m = max(i for i, y in enumerate(out) if y.kind == "mfma")
    gap = sum(y.states() for y in out[last_m + 1:])
    if gap < 12 and not tail:        # (--tail: the validation below sits behind the last MFMA; checked at the end)
        out.append(nop(12 - gap - 1))
    if nan_check:
        out.append(Ins("cmp", "v_cmp_u_f32_e64 %[bad], %[nan], %[nan]", reads=("nan",), writes=("bad",)))
    if tail:
        # the publish is inside the stream already; what is left behind the last MFMA is the validation
        out.append(Ins("wait", "s_waitcnt lgkmcnt(0)"))
    elif publish:
        # h_t of the lane's three cells -> fp16 -> the wave's LDS staging row (2-byte writes) -> read back as the 8 bytes this lane
        # moves (one wave: LDS operations complete in issue order, no wait between the writes and the read). The OR tree of the
        # validation runs while that read is in flight.
        assert validate_at >= 0
        for m in range(3):
            out.append(Ins("valu", "v_cvt_f16_f32_e32 %%[hh%d], v%d" % (m, PBASE + 4 * m), writes=("hh%d" % m,)))
        for m in range(3):
            out.append(Ins("lds", "ds_write_b16 %%[sga], %%[hh%d]%s" % (m, " offset:%d" % (2 * m) if m else ""), reads=("sga", "hh%d" % m)))
        out.append(Ins("lds", "ds_read_b64 %[pk], %[rda]", reads=("rda",), writes=("pk",)))
        out.append(Ins("wait", "s_waitcnt lgkmcnt(4)"))          # the three validation reads are home (three writes + one read younger)
    else:
        out.append(Ins("wait", "s_waitcnt lgkmcnt(0)"))
    if validate_at >= 0:         # any of the twelve dwords still carrying the sentinel bit? (the stream's own LDS reads are all home by now)
        d = ["v%d" % (VBASE + i) for i in range(12)]
        out.append(Ins("valu", "v_or3_b32 %s, %s, %s, %s" % (d[0], d[0], d[1], d[2])))
        out.append(Ins("valu", "v_or3_b32 %s, %s, %s, %s" % (d[4], d[3], d[4], d[5])))
        out.append(Ins("valu", "v_or3_b32 %s, %s, %s, %s" % (d[8], d[6], d[7], d[8])))
        out.append(Ins("valu", "v_or3_b32 %s, %s, %s, %s" % (d[0], d[0], d[9], d[10])))
        out.append(Ins("valu", "v_or3_b32 %s, %s, %s, %s" % (d[4], d[4], d[8], d[11])))
        out.append(Ins("valu", "v_or_b32_e32 %s, %s, %s" % (d[0], d[0], d[4])))
        out.append(Ins("valu", "v_and_b32_e32 %s, 0x40004000, %s" % (d[0], d[0])))
        out.append(Ins("cmp", "v_cmp_ne_u32_e64 %%[bad], 0, %s" % d[0], writes=("bad",)))
    if publish and not tail:
        # publish h_t into its exchange slot, re-arm the slot of h_{t+2}, write the layer output row: plain stores when the ring sits on
        # one XCD (`fast`), write-through otherwise - the policy of the C++ sections
        out.append(Ins("wait", "s_waitcnt lgkmcnt(0)"))
        out.append(Ins("salu", "s_cmp_lg_u32 %[fast], 0", reads=("fast",)))
        out.append(Ins("salu", "s_cbranch_scc1 1f"))
        out.append(Ins("vmem", "global_store_dwordx2 %[vmy], %[pk], %[exs] sc1", reads=("vmy", "pk", "exs")))
        out.append(Ins("vmem", "global_store_dwordx2 %[vmy], %[ones], %[exa] sc1", reads=("vmy", "ones", "exa")))
        out.append(Ins("salu", "s_branch 2f"))
        out.append(Ins("label", "1:"))
        out.append(Ins("vmem", "global_store_dwordx2 %[vmy], %[pk], %[exs]", reads=("vmy", "pk", "exs")))
        out.append(Ins("vmem", "global_store_dwordx2 %[vmy], %[ones], %[exa]", reads=("vmy", "ones", "exa")))
        out.append(Ins("label", "2:"))
        out.append(Ins("vmem", "global_store_dwordx2 %[vh], %[pk], %[hrow]", reads=("vh", "pk", "hrow")))
    if tail:
        gap = sum(y.states() for y in out[last_m + 1:])
        if gap < 12:
            out.append(nop(12 - gap - 1))
    return out


def operand(name):
    """name -> (register class, c++ expression)"""
    if name.startswith("wh"):
        m, ks = name[2:].split("_")
        return "a", "whh[%s][%s]" % (m, ks)
    if name.startswith("wi"):
        m, ks = name[2:].split("_")
        return ("a" if int(m) < 2 or ALL_AGPR else "v"), "wih[%s][%s]" % (m, ks)
    if name.startswith("xa"):
        return "v", "xacc[%s]" % name[2:]
    if name.startswith("bi"):
        return "v", "bias[%s]" % name[2:]
    if name.startswith("ht") or name.startswith("xt"):
        return "v", name
    if name[0] == "e" and name[1:].isdigit():
        return "v", "e[%s]" % name[1]
    if name[0] == "c" and name[1:].isdigit():
        return "v", "cst[%s]" % name[1]
    if name in ("hb", "xb"):
        return "v", name
    if name == "kneg2":
        return "s", "0xbfb8aa3bbfb8aa3bull"
    if name == "kc2n":
        return "s", "0xbfb8aa3bc038aa3bull"          # low half: -2 log2e (lane pair element 0 = the g gate), high half: -log2e (o gate)
    if name == "kone2":
        return "s", "0x3f8000003f800000ull"
    if name in ("hoff", "xoff", "voff", "pmo", "xmo"):
        return "n", name.upper()
    if name == "nan":
        return "v", "nanacc"
    if name == "bad":
        return "s", "bad"
    if name == "pm0":
        return "s", "pm0"
    if name.startswith("vp"):
        return "v", "vp[%s]" % name[2:]
    if name == "exo":
        return "s", "exo"
    if name == "xsrc":
        return "s", "xsrc"
    if name in ("xm0", "vx", "hbo"):
        return ("s" if name == "xm0" else "v"), name
    if name in ("xv1", "xv2"):
        return "v", name
    if name in ("hh0", "hh1", "hh2"):
        return "v", "hh[%s]" % name[2]
    if name == "pk":
        return "v", "pk"
    if name in ("sga", "rda", "vmy", "vh"):
        return "v", name
    if name == "ones":
        return "v", "~0ull"
    if name in ("exs", "exa", "hrow"):
        return "s", name
    if name == "fast":
        return "s", "fast"
    if name == "hi25":
        return "v", "25.0f"
    if name == "hi12":
        return "v", "12.5f"
    raise KeyError(name)


def render(seq, fn, hf_live, header):
    names = {n for x in seq for n in x.reads + x.writes}
    extra_args = ""
    if "bad" in names:
        extra_args += ", unsigned long long& bad"
    if "exo" in names:
        extra_args += ", unsigned pm0, const char* exo, const unsigned (&vp)[3]"
    if "xsrc" in names:
        extra_args += ", unsigned xm0, const char* xsrc, unsigned vx"
    if "hbo" in names:
        extra_args += ", unsigned hbo"
    if "pk" in names:
        extra_args += ", unsigned sga, unsigned rda, unsigned vmy, unsigned vh, const char* exs, const char* exa, const char* hrow, unsigned fast"
    lines = ["// GENERATED by tools/gen_ringstep.py - do not edit. " + header,
             "// in: xacc[m] = bias + W_ih x_t (tile m), cst; h_{t-1} fragments at LDS address hb, x_{t+1} fragments at xb (+ lane * 16 each)",
             "// out: hv[m] = h_t of the lane's three cells, cst, xacc[m] = bias + W_ih x_{t+1}",
             *(["// --imm: hb / xb / hbo / pm0 / xm0 are loop-invariant bases, the tile they address is the template constant added to them (HOFF: this ring's h",
                "// tile, XOFF: its x slot of step t+1, VOFF / PMO: the other ring's h tile that is validated / polled into, XMO: the x slot the DMA fills)"] if IMM else []),
             ("template <int HOFF, int XOFF, int VOFF, int PMO, int XMO>\n" if IMM else "") +
             "__device__ __forceinline__ void %s(float4_t (&xacc)[3], float (&cst)[3], float (&hv)[3], const half8_t (&whh)[3][12]," % fn,
             "        const half8_t (&wih)[3][12], const float4_t (&bias)[3], unsigned hb, unsigned xb%s) {" % extra_args,
             "    float e[3], nanacc;",
             "    unsigned xv1, xv2;",
             "    uint4_t vt[3];",
             "    unsigned hh[3];",
             "    unsigned long long pk;",
             "    float4_t acc[3];",
             "    half8_t " + ", ".join(["ht%d" % i for i in range(NKS if hf_live else 8)] + ["xt%d" % i for i in range(4)]) + ";"]
    first_access, written = {}, set()
    for x in seq:
        for r in x.reads:
            if r != "vcc" and r[0] not in "gP":
                first_access.setdefault(r, "r")
        for w in x.writes:
            if w != "vcc" and w[0] not in "gP":
                first_access.setdefault(w, "w"); written.add(w)
    outs = ['"=&{v[%d:%d]}"(acc[%d])' % (PBASE + 4 * m, PBASE + 4 * m + 3, m) for m in range(3)]
    if "hbo" in names:
        outs += ['"=&{v[%d:%d]}"(vt[%d])' % (VBASE + 4 * k, VBASE + 4 * k + 3, k) for k in range(3)]
    ins = []
    for name in first_access:
        cls, expr = operand(name)
        if name in written:
            outs.append('[%s] "%s"(%s)' % (name, ("+" if first_access[name] == "r" else "=&") + cls, expr))
        else:
            ins.append('[%s] "%s"(%s)' % (name, cls, expr))
    body = '\\n\\t"\n        "'.join(x.text for x in seq)
    lines.append('    asm volatile("' + body + '"\n        : ' + ", ".join(outs) + "\n        : " + ", ".join(ins) + ('\n        : "vcc", "scc", "memory");' if "exo" in names else '\n        : "vcc");'))
    if any(x.meta.get("cell") is not None for x in seq):
        lines += ["    hv[0] = acc[0][0]; hv[1] = acc[1][0]; hv[2] = acc[2][0]; (void)nanacc; (void)xv1; (void)xv2; (void)vt; (void)hh; (void)pk;" + (" (void)fast;" if "pk" in names else ""), "}", ""]
    else:
        lines += ["    hv[0] = acc[0][0]; hv[1] = acc[1][0]; hv[2] = acc[2][0]; (void)e; (void)nanacc; (void)xv1; (void)xv2; (void)vt; (void)hh; (void)pk;" + (" (void)fast;" if "pk" in names else ""), "}", ""]
    return "\n".join(lines)


# The two streams the library is built from (python tools/gen_ringstep.py --preset plain|paired); the parameters are the best of the
# sweeps of round 3 (tools/stream_bench.py for the arithmetic, tools/gpu_variants.sh + tools/lstm_stats2.py for the in-stream DMA)
PRESETS = {
    "plain": ["--hf-live", "--xdist", "0,10,10,16", "--xdepth", "3"],
    # (--pbase: the single-ring kernel otherwise needs ~190 VGPRs; with the accumulators at v[232..] its 244 + 240 registers left no room
    #  for a decode wave beside it, and the one-batch-per-call pipeline lost the overlap it lives on: 18.0 -> 19.4 ms per batch)
    "rec": ["--hf-live", "--xdist", "0,0,0,0", "--pbase", "144", "--name", "ringstep3r_mfma", "--out",
            os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bonito_amd", "csrc", "ringstep3r_mfma.inc")],
    "paired": ["--hf-live", "--xdist", "0,10,10,16", "--xdepth", "3", "--polls-at", "26", "--xdma-at", "42", "--spread", "2",
               "--validate-at", "71", "--publish", "--name", "ringstep3p_mfma", "--out",
               os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bonito_amd", "csrc", "ringstep3p_mfma.inc")],
}
# the same stream for the main loop of the paired kernel, unrolled over four steps: every tile / slot choice is a compile-time constant
# (--tail 8: the publish inside the last eight MFMAs, plain stores - the main loop runs only where the ring sits on one XCD; polls four MFMAs
#  later than in the generic stream: the section is ~350 cycles shorter, the peers' publishes are not visible any earlier)
PRESETS["unrolled"] = PRESETS["paired"][:-4] + ["--imm", "--tail", "8", "--polls-at", "30", "--xdma-at", "46", "--name", "ringstep3u_mfma", "--out",
                                                PRESETS["paired"][-1].replace("ringstep3p", "ringstep3u")]


def main(argv=None):
    import sys
    argv = list(sys.argv[1:] if argv is None else argv)
    if len(argv) >= 2 and argv[0] == "--preset":
        argv = PRESETS[argv[1]] + argv[2:]
    ap = argparse.ArgumentParser()
    ap.add_argument("--xdist", default="0,12,12,12")
    ap.add_argument("--hdepth", type=int, default=5)
    ap.add_argument("--xdepth", type=int, default=6)
    ap.add_argument("--hpool", type=int, default=6)
    ap.add_argument("--xpool", type=int, default=3)
    ap.add_argument("--lead", type=int, default=2, help="MFMAs of a phase in front of its first vector instruction")
    ap.add_argument("--wgroup", type=int, default=1, help="fragments covered by one lgkmcnt wait")
    ap.add_argument("--nan-check", action="store_true", help="fold the pre-activations into a NaN probe: `bad` = lanes that saw an element of h that had not arrived")
    ap.add_argument("--polls-at", type=int, default=-1, help="issue the other ring's three poll DMAs behind this MFMA (0-based)")
    ap.add_argument("--xdma-at", type=int, default=-1, help="issue this ring's three x-stream DMAs behind this MFMA")
    ap.add_argument("--spread", type=int, default=0, help="MFMAs between two DMA instructions of the poll / x-stream groups")
    ap.add_argument("--publish", action="store_true", help="LDS transpose of h_t and the section's three stores at the end of the stream")
    ap.add_argument("--validate-at", type=int, default=-1, help="read back my quarter of the other ring's h tile behind this MFMA; `bad` = sentinel found")
    ap.add_argument("--pbase", type=int, default=232, help="first of the twelve physical accumulator registers (the kernel's VGPR count is at least this + 12)")
    ap.add_argument("--vbase", type=int, default=220, help="first of the twelve physical registers of the validation read-back")
    ap.add_argument("--imm", action="store_true", help="tile offsets as template constants on loop-invariant bases (the unrolled main loop of the paired kernel)")
    ap.add_argument("--tail", type=int, default=0, help="weave the publish (plain stores) into the last K MFMAs; cell 2's gate arithmetic over the ones before")
    ap.add_argument("--packed", action="store_true", help="experiment: gate arithmetic with v_pk_mul_f32 / v_pk_add_f32 and fused constants (same bits, 30 instead of 35 instructions per cell; measured SLOWER: 1965 vs 1870 cycles per ring step)")
    ap.add_argument("--poll-flags", default="sc0 sc1", help="timing experiments: cache policy bits of the poll DMAs")
    ap.add_argument("--all-agpr", action="store_true", help="timing experiments: W_ih tile 2 as AGPR operands too")
    ap.add_argument("--strip", default="", help="timing experiments only (wrong results): 'valu' drops the gate arithmetic, 'mfma' drops MFMAs + LDS reads")
    ap.add_argument("--hf-live", action="store_true", help="keep the twelve h fragments in registers instead of re-reading them per tile")
    ap.add_argument("--name", default="ringstep3_mfma")
    ap.add_argument("--out", default=os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bonito_amd", "csrc", "ringstep3_mfma.inc"))
    a = ap.parse_args(argv)
    global PBASE, VBASE, ALL_AGPR, IMM, POLL_FLAGS, PACKED
    PBASE, VBASE, ALL_AGPR, IMM, POLL_FLAGS, PACKED = a.pbase, a.vbase, a.all_agpr, a.imm, a.poll_flags, a.packed
    xdist = tuple(int(v) for v in a.xdist.split(","))
    seq = build(xdist, a.hdepth, a.xdepth, a.hpool, a.xpool, a.hf_live, a.lead, a.wgroup, a.nan_check, a.polls_at, a.xdma_at, a.validate_at, a.spread, a.publish, a.tail)
    if a.strip == "valu":
        seq = [x for x in seq if x.kind not in ("valu", "trans", "cmp", "sel")]
    elif a.strip == "mfma":
        seq = [x for x in seq if x.kind not in ("mfma", "lds", "wait")]
    counts = {}
    for x in seq:
        counts[x.kind] = counts.get(x.kind, 0) + 1
    header = "xdist=%s hdepth=%d xdepth=%d hpool=%d xpool=%d lead=%d wgroup=%d hf_live=%d nan_check=%d polls_at=%d xdma_at=%d validate_at=%d spread=%d publish=%d%s : %s" % (
        a.xdist, a.hdepth, a.xdepth, a.hpool, a.xpool, a.lead, a.wgroup, a.hf_live, a.nan_check, a.polls_at, a.xdma_at, a.validate_at, a.spread, a.publish,
        (" imm=1" if a.imm else "") + (" packed=1" if a.packed else "") + (" tail=%d" % a.tail if a.tail else "") + (" poll_flags=%s" % a.poll_flags.replace(" ", "+") if a.poll_flags != "sc0 sc1" else ""),
        " ".join("%s=%d" % kv for kv in sorted(counts.items())))
    with open(a.out, "w") as fh:
        fh.write(render(seq, a.name, a.hf_live, header))
    print("wrote", a.out, header)


if __name__ == "__main__":
    main()
