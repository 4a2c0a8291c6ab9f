#!/usr/bin/env python3
"""profiles/pmc_traffic.json and profiles/<tag>_pmc_hbm_bytes.txt from the PMC passes of tools/prof_round.sh - generated, never edited.

    python tools/pmc_traffic.py gpurun_out/prof_<tag> <tag> "<workload, e.g. hac 1024x10000>"

reads pmc_rd.txt / pmc_wr.txt (tools/pmc_summary.py output of the separate `--pmc FETCH_SIZE` and `--pmc WRITE_SIZE` passes), writes
profiles/<tag>_pmc_hbm_bytes.txt (both lists under one header) and, for every recurrent kernel found, the entry of
profiles/pmc_traffic.json that bench.py copies into roofline.traffic (key "<kernel>|<workload>"): bytes per launch = 1024 * (2 * FETCH_SIZE + WRITE_SIZE) - the
counters are KiB per dispatch, and on gfx950 FETCH_SIZE reports half the bytes of 16-byte coalesced streams (MI355X_MICROARCH.md, HBM)."""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def parse(path):
    rows = []
    if not os.path.exists(path):
        return rows
    for line in open(path):
        m = re.match(r"(.+?)\s+(FETCH_SIZE|WRITE_SIZE)\s+n=\s*(\d+)\s+avg\s+([0-9.e+]+)", line.rstrip())
        if m:
            rows.append((m.group(1).strip(), m.group(2), int(m.group(3)), float(m.group(4)), line.rstrip()))
    return rows


def main():
    src, tag, workload = sys.argv[1], sys.argv[2], sys.argv[3]
    rd, wr = parse(os.path.join(src, "pmc_rd.txt")), parse(os.path.join(src, "pmc_wr.txt"))
    if not rd or not wr:
        sys.exit("no PMC rows under " + src)
    out = os.path.join(ROOT, "profiles", "%s_pmc_hbm_bytes.txt" % tag)
    with open(out, "w") as fh:
        fh.write("# rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes, tools/prof_round.sh %s), tools/profile_step.py --steps 1, workload %s\n" % (tag, workload))
        fh.write("# values are KiB per dispatch as reported; gfx950: double FETCH_SIZE of 16-B coalesced streams (MI355X_MICROARCH.md)\n")
        for r in rd + wr:
            fh.write(r[4] + "\n")
    table_path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    table = json.load(open(table_path))
    fetch = {r[0]: (r[2], r[3]) for r in rd}
    write = {r[0]: (r[2], r[3]) for r in wr}
    # the kernels a bench roofline can name: the recurrent kernels, fc1 (+ SwiGLU) of the transformer, the attention kernel. Several template
    # instances of one kernel (the last 8-bit recurrent layer also writes fp16 rows) are averaged over their launches.
    wanted = (r"(lstm_layer_\w+_kernel)", r"(gemm_w4_kernel)<0, true, 4, 0(?:, true)?>", r"(attention_ring2?_kernel)")
    acc = {}
    for name in fetch:
        m = None
        for pat in wanted:
            m = m or re.search(pat, name)
        if not m or name not in write:
            continue
        n = fetch[name][0]
        ent = acc.setdefault(m.group(1), {"n": 0, "bytes": 0.0, "kernels": []})
        ent["n"] += n
        ent["bytes"] += n * 1024 * (2 * fetch[name][1] + write[name][1])
        ent["kernels"].append(name)
    for key, ent in acc.items():
        out_ent = {"workload": workload, "bytes_per_launch": int(round(ent["bytes"] / ent["n"], -6)), "launches": ent["n"],
                   "kernel": "; ".join(ent["kernels"]), "source": "profiles/%s_pmc_hbm_bytes.txt" % tag}
        table["%s|%s" % (key, workload)] = out_ent
        print(key, out_ent)
    with open(table_path, "w") as fh:
        json.dump(table, fh, indent=1)
        fh.write("\n")
    print("wrote", out, "and", table_path)


if __name__ == "__main__":
    main()
