#!/usr/bin/env python3
"""Dump the per-kernel statistics of a rocprofv3 (rocpd sqlite) result as text: tools/rocpd_summary.py results.db"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
print(f"# rocprofv3 --kernel-trace --stats summary of {sys.argv[1]} (durations in microseconds)")
print(f"{'kernel':60s} {'calls':>6s} {'total_us':>12s} {'avg_us':>10s} {'pct':>6s}")
for name, calls, total, avg, pct in cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels"):
    print(f"{name[:60]:60s} {calls:6d} {total:12.1f} {avg:10.1f} {pct:6.2f}")
