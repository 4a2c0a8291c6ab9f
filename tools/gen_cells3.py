#!/usr/bin/env python3
"""Generates tools/cells3_mfma.inc: lstm_cell() of THREE units, their dependency chains interleaved round-robin, with the
36 MFMAs of the next step's input projection (H = 384: 12 k-steps x 3 M tiles) threaded through at about one MFMA to three vector
instructions. Five asm blocks of at most 30 distinct operands each, named operands. The arithmetic is lstm_cell()'s
(bonito_amd/csrc/lstm.hip), operation for operation:

    ei = exp2(med3(ai, +-25) * -log2e), ef, eo likewise;  eg = exp2((med3(ag, +-12.5) * -2) * log2e)
    didg = (1 + ei)(1 + eg); df = 1 + ef; num = fma(c, didg, (1 - eg) df); c' = num * rcp(df * didg)
    ec = exp2((med3(c', +-12.5) * -2) * log2e); hv = (1 - ec) * rcp((1 + ec)(1 + eo)); h = |hv| <= 1 ? hv : 0

Hand-placed hazards: a transcendental result is never used by the next instruction (round-robin over the cells + MFMAs in between),
v_cmp -> v_cndmask on vcc with an MFMA in between, MFMA operands come from outside, the last block ends with the wait states the
accumulators need before the vector ALU may read them.
"""
import os
L_NEG = "0xbfb8aa3b"      # -log2(e)
L_POS = "0x3fb8aa3b"      # +log2(e)


class Block:
    def __init__(self):
        self.ops = {}          # name -> (constraint, c++ expr), insertion ordered
        self.lines = []

    def reg(self, name, cons, expr):
        if name not in self.ops:
            self.ops[name] = (cons, expr)
        return "%[" + name + "]"

    def emit(self, text):
        self.lines.append(text)

    def render(self, clobber_vcc=False, tail=""):
        outs = [(n, c, e) for n, (c, e) in self.ops.items() if c.startswith(("+", "="))]
        ins = [(n, c, e) for n, (c, e) in self.ops.items() if not c.startswith(("+", "="))]
        assert len(outs) + len(ins) <= 30, (len(outs), len(ins))
        body = "\\n\\t\"\n        \"".join(self.lines)
        s = '    asm("' + body + tail + '"\n'
        s += "        : " + ", ".join('[%s] "%s"(%s)' % (n, c, e) for n, c, e in outs) + "\n"
        s += "        : " + ", ".join('[%s] "%s"(%s)' % (n, c, e) for n, c, e in ins)
        if clobber_vcc:
            s += '\n        : "vcc"'
        s += ");\n"
        return s


def weave(valu, mfmas, lead=2, tail=4):
    """The vector instructions with the MFMAs spread evenly between them, at least `lead` vector instructions ahead of the first MFMA
    and `tail` behind the last one: the compiler knows nothing of the MFMAs inside a block and may copy an accumulator right behind
    it (it did), and an MFMA result needs 11 cycles before the vector ALU may read it - `tail` instructions of 4 cycles each."""
    n, m = len(valu), len(mfmas)
    assert n >= lead + tail and m >= 1
    span = n - lead - tail                      # MFMA j goes behind vector instruction lead + round(j * span / (m - 1))
    at = [lead + (round(j * span / (m - 1)) if m > 1 else 0) for j in range(m)]
    out, j = [], 0
    for i, v in enumerate(valu):
        while j < m and at[j] == i:
            out.append(mfmas[j]); j += 1
        out.append(v)
    assert j == m
    return out


def main():
    blocks = []
    ks_of = [(0, 1, 2), (3, 4, 5), (6, 7), (8, 9), (10, 11)]

    def mf(b, ks_list):
        res = []
        for ks in ks_list:
            bb = b.reg("b%d" % ks, "v", "bf[%d]" % ks)
            for t in range(3):
                w = b.reg("w%d_%d" % (t, ks), "a" if t < 2 else "v", "wih[%d][%d]" % (t, ks))
                x = b.reg("x%d" % t, "+v", "xa[%d]" % t)
                res.append("v_mfma_f32_16x16x32_f16 %s, %s, %s, %s" % (x, w, bb, x))
        return res

    def G(b, cell, gate, cons="+v"):
        return b.reg("g%d%d" % (cell, gate), cons, "g[%d][%d]" % (cell, gate))

    def E(b, cell, cons="+v"):
        return b.reg("e%d" % cell, cons, "e[%d]" % cell)

    def C(b, cell, cons="+v"):
        return b.reg("c%d" % cell, cons, "cst[%d]" % cell)

    def rr(per_cell):
        """round-robin interleave of three per-cell instruction lists of equal length"""
        out = []
        for i in range(len(per_cell[0])):
            for c in range(3):
                out.append(per_cell[c][i])
        return out

    # block 1: clamp + scale the four pre-activations
    b = Block()
    hi25 = b.reg("hi25", "v", "25.0f"); hi12 = b.reg("hi12", "v", "12.5f")
    per = []
    for c in range(3):
        g0, g1, g2, g3 = (G(b, c, i) for i in range(4))
        per.append(["v_med3_f32 %s, %s, %s, -%s" % (g0, g0, hi25, hi25), "v_med3_f32 %s, %s, %s, -%s" % (g1, g1, hi25, hi25),
                    "v_med3_f32 %s, %s, %s, -%s" % (g2, g2, hi12, hi12), "v_med3_f32 %s, %s, %s, -%s" % (g3, g3, hi25, hi25),
                    "v_mul_f32 %s, %s, %s" % (g0, L_NEG, g0), "v_mul_f32 %s, %s, %s" % (g1, L_NEG, g1), "v_mul_f32 %s, -2.0, %s" % (g2, g2),
                    "v_mul_f32 %s, %s, %s" % (g3, L_NEG, g3), "v_mul_f32 %s, %s, %s" % (g2, L_POS, g2)])
    for l in weave(rr(per), mf(b, ks_of[0])): b.emit(l)
    blocks.append(b.render())
    # block 2: the four exponentials, 1 + e terms
    b = Block()
    per = []
    for c in range(3):
        g0, g1, g2, g3 = (G(b, c, i) for i in range(4))
        e = E(b, c, "=&v")
        per.append(["v_exp_f32 %s, %s" % (g0, g0), "v_exp_f32 %s, %s" % (g1, g1), "v_exp_f32 %s, %s" % (g2, g2), "v_exp_f32 %s, %s" % (g3, g3),
                    "v_add_f32 %s, 1.0, %s" % (g0, g0), "v_add_f32 %s, 1.0, %s" % (g1, g1), "v_sub_f32 %s, 1.0, %s" % (e, g2),
                    "v_add_f32 %s, 1.0, %s" % (g2, g2)])
    for l in weave(rr(per), mf(b, ks_of[1])): b.emit(l)
    blocks.append(b.render())
    # block 3: the cell state
    b = Block()
    per = []
    for c in range(3):
        g0, g1, g2 = (G(b, c, i) for i in range(3))
        e, cs = E(b, c), C(b, c)
        per.append(["v_mul_f32 %s, %s, %s" % (g0, g0, g2), "v_mul_f32 %s, %s, %s" % (e, e, g1), "v_mul_f32 %s, %s, %s" % (g2, g1, g0),
                    "v_fma_f32 %s, %s, %s, %s" % (e, cs, g0, e), "v_rcp_f32 %s, %s" % (g2, g2), "v_mul_f32 %s, %s, %s" % (cs, e, g2)])
    for l in weave(rr(per), mf(b, ks_of[2])): b.emit(l)
    blocks.append(b.render())
    # block 4: exp(-2 c'), 1 + eo
    b = Block()
    hi12 = b.reg("hi12", "v", "12.5f")
    per = []
    for c in range(3):
        g0, g3 = G(b, c, 0), G(b, c, 3)
        cs = C(b, c, "v")
        per.append(["v_med3_f32 %s, %s, %s, -%s" % (g0, cs, hi12, hi12), "v_mul_f32 %s, -2.0, %s" % (g0, g0), "v_mul_f32 %s, %s, %s" % (g0, L_POS, g0),
                    "v_exp_f32 %s, %s" % (g0, g0), "v_add_f32 %s, 1.0, %s" % (g3, g3)])
    for l in weave(rr(per), mf(b, ks_of[3])): b.emit(l)
    blocks.append(b.render())
    # block 5: h
    b = Block()
    per = []
    for c in range(3):
        g0, g1, g3 = G(b, c, 0), G(b, c, 1), G(b, c, 3)
        per.append(["v_add_f32 %s, 1.0, %s" % (g1, g0), "v_sub_f32 %s, 1.0, %s" % (g0, g0), "v_mul_f32 %s, %s, %s" % (g1, g1, g3),
                    "v_rcp_f32 %s, %s" % (g1, g1), "v_mul_f32 %s, %s, %s" % (g0, g0, g1)])
    m5 = mf(b, ks_of[4])
    seq = weave(rr(per), m5[:3], lead=2, tail=1)
    # compare / select through vcc, one cell at a time, an MFMA (>= 2 wait states) between the compare and the select
    for c in range(3):
        g0 = G(b, c, 0)
        seq += ["v_cmp_le_f32_e64 vcc, |%s|, 1.0" % g0, m5[3 + c], "s_nop 0", "v_cndmask_b32_e32 %s, 0, %s, vcc" % (g0, g0)]
    for l in seq: b.emit(l)
    blocks.append(b.render(clobber_vcc=True, tail="\\n\\ts_nop 15\\n\\ts_nop 7"))

    out = ["// GENERATED by tools/gen_cells3.py - do not edit. See the generator for what this is.",
           "__device__ __forceinline__ void cells3_mfma(const float4_t (&acc)[3], float (&cst)[3], float (&hv)[3], const half8_t (&wih)[3][12],",
           "                                            const half8_t (&bf)[12], float4_t (&xa)[3]) {",
           "    float g[3][4], e[3];",
           "#pragma unroll",
           "    for (int c = 0; c < 3; ++c)",
           "#pragma unroll",
           "        for (int i = 0; i < 4; ++i) g[c][i] = acc[c][i];"]
    out += [blk for blk in blocks]
    out += ["#pragma unroll", "    for (int c = 0; c < 3; ++c) hv[c] = g[c][0];", "}", ""]
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "cells3_mfma.inc")
    with open(path, "w") as fh:
        fh.write("\n".join(out))
    print("wrote", path)


if __name__ == "__main__":
    main()
