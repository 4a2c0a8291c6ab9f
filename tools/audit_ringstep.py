#!/usr/bin/env python3
"""Audit of the compiled ring-step stream (tools/gen_ringstep.py): the stream is several asm statements with MFMAs and LDS reads in
flight across the cuts, and the compiler is free to place its own instructions between two statements (it does: sub-register copies
when a cell's pre-activations leave the accumulator tuple). This checks, on the compiler's assembly output (-S), that whatever sits
between two statements of one stream
  * is a scalar instruction, an s_nop or a v_mov_b32 / v_accvgpr move, and
  * touches no register that an MFMA of the statement before wrote within the last 12 wait states (8-pass XDL result not readable yet),
  * and no destination of an LDS read the statement before has not waited for.
usage: audit_ringstep.py file.s   (exit status 1 and a report on a violation)
"""
import re
import sys

REG = re.compile(r"\b([va])(\d+)\b|\b([va])\[(\d+):(\d+)\]")


def regs(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(1):
            out.add((m.group(1), int(m.group(2))))
        else:
            out |= {(m.group(3), i) for i in range(int(m.group(4)), int(m.group(5)) + 1)}
    return out


def states(line):
    m = re.match(r"s_nop (\d+)", line)
    return int(m.group(1)) + 1 if m else 1


def unsafe_after(stmt):
    """registers that must not be touched right behind this statement"""
    pending = []                                     # destinations of LDS reads not yet waited for, oldest first
    for line in stmt:
        if line.startswith("ds_read"):
            pending.append(regs(line.split(",")[0]))
        m = re.match(r"s_waitcnt .*lgkmcnt\((\d+)\)", line)
        if m:
            n = int(m.group(1))
            pending = pending[len(pending) - n:] if n else []
    bad = set()
    for p in pending:
        bad |= p
    st = 0
    for line in reversed(stmt):
        if st >= 12:
            break
        if line.startswith("v_mfma"):
            bad |= regs(line.split(",")[0])
        st += states(line)
    return bad


def audit(path):
    text = open(path).read()
    problems, checked = [], 0
    for km in re.finditer(r"^(\w+):[^\n]*\n(.*?)\n\s*s_endpgm", text, re.S | re.M):
        name, body = km.group(1), [l.strip() for l in km.group(2).split("\n")]
        marks = [i for i, l in enumerate(body) if l.startswith("; RS3 statement")]
        for a, b in zip(marks[:-1], marks[1:]):
            ka, kb = int(body[a].split()[-1]), int(body[b].split()[-1])
            if kb != ka + 1:
                continue                              # the next stream instance
            end = next(i for i in range(a, b) if body[i].startswith(";;#ASMEND"))
            start = max(i for i in range(a, b) if body[i].startswith(";;#ASMSTART"))
            stmt = [l for l in body[a + 1:end] if l and not l.startswith(";")]
            between = [l for l in body[end + 1:start] if l and not l.startswith(";") and not l.endswith(":")]
            bad = unsafe_after(stmt)
            checked += 1
            for l in between:
                op = l.split()[0]
                if op.startswith("s_"):
                    continue
                if not re.match(r"v_mov_b(32|64)|v_accvgpr_(read|write|mov)", op):
                    problems.append("%s: statement %d -> %d: unexpected instruction between the statements: %s" % (name, ka, kb, l))
                    continue
                hit = regs(l) & bad
                if hit:
                    problems.append("%s: statement %d -> %d: `%s` touches in-flight register(s) %s" % (name, ka, kb, l, sorted(hit)))
    return checked, problems


if __name__ == "__main__":
    n, probs = audit(sys.argv[1])
    print("%d statement boundaries checked, %d problem(s)" % (n, len(probs)))
    for p in probs:
        print("  " + p)
    sys.exit(1 if probs else 0)
