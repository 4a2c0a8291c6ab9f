#!/usr/bin/env python3
"""Summarise a rocprofv3 (ROCm 7.2, rocpd sqlite) --kernel-trace run into a per-kernel CSV.

    python tools/rocprof_summary.py gpurun_out/prof1/r01_hac_beam_results.db profiles/r01_hac_beam_kernel_stats.csv

AverageNs is over EVERY launch of the trace. In a trace of bench.py that includes the launches of the software-pipelined timed region, where a
kernel of one call can start while the decode kernels of the call before still hold the CUs (its duration then contains the wait for its last
workgroups to become resident: MaxNs well above MinNs). MedianNs and Last30AverageNs (the last 30 launches of the kernel: bench.py's roofline leg,
three forwards run one kernel at a time after the timed regions - the launches its `roofline.avg_launch_ms` is measured on) are the figures to
hold against the bench line.
"""
import csv
import sqlite3
import sys


def main(db, out):
    con = sqlite3.connect(db)
    rows = list(con.execute(
        "select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start), "
        "max(vgpr_count), max(accum_vgpr_count), max(sgpr_count), max(lds_size), max(grid_x), max(workgroup_x) "
        "from kernels group by name order by 3 desc"))
    total = sum(r[2] for r in rows) or 1
    extra = {}
    for name, in con.execute("select distinct name from kernels"):
        d = [r[0] for r in con.execute("select end-start from kernels where name = ? order by start", (name,))]
        tail = d[-30:]
        extra[name] = (sorted(d)[len(d) // 2], sum(tail) / len(tail))
    with open(out, "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs", "Percentage", "VGPRs",
                    "AccumVGPRs", "SGPRs", "LDSBytes", "GridX", "WorkgroupX", "MedianNs", "Last30AverageNs"])
        for r in rows:
            w.writerow([r[0], r[1], int(r[2]), round(r[3], 1), int(r[4]), int(r[5]), round(100.0 * r[2] / total, 2),
                        r[6], r[7], r[8], r[9], r[10], r[11], int(extra[r[0]][0]), round(extra[r[0]][1], 1)])
    for r in rows[:12]:
        print("%-64s n=%4d avg %10.1f us (median %10.1f, last 30: %10.1f) %5.1f%%" % (r[0][:64], r[1], r[3] / 1e3, extra[r[0]][0] / 1e3,
                                                                                    extra[r[0]][1] / 1e3, 100.0 * r[2] / total))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
