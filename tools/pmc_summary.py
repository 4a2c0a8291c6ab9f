#!/usr/bin/env python3
"""Per-kernel PMC averages from a rocprofv3 --pmc run (rocpd sqlite).  Usage: pmc_summary.py run.db"""
import sqlite3
import sys

con = sqlite3.connect(sys.argv[1])
cols = [d[1] for d in con.execute("pragma table_info(counters_collection)")]
print(cols)
q = ("select kernel_name, counter_name, count(*), avg(value), sum(value) from counters_collection "
     "group by kernel_name, counter_name order by 5 desc")
try:
    for r in con.execute(q):
        print("%-60s %-14s n=%4d avg %.6g" % (r[0][:60], r[1], r[2], r[3]))
except Exception as exc:
    print("query failed:", exc)
