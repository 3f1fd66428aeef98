#!/usr/bin/env python3
"""Turn a rocprofv3 rocpd (.db) kernel trace into the per-kernel summary table committed under profiles/.
usage: python tools/rocpd_summary.py gpurun_out/prof1/r01_results.db > profiles/r01_kernel_stats.csv"""
import sqlite3
import sys

con = sqlite3.connect(sys.argv[1])
rows = con.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels "
                   "group by name order by sum(duration) desc").fetchall()
total = sum(r[2] for r in rows)
print("kernel,calls,total_us,avg_us,min_us,max_us,percent")
for name, n, tot, avg, mn, mx in rows:
    short = name.split("(")[0].replace("void ", "")
    print(f"\"{short}\",{n},{tot / 1e3:.1f},{avg / 1e3:.2f},{mn / 1e3:.2f},{mx / 1e3:.2f},{100 * tot / total:.3f}")
