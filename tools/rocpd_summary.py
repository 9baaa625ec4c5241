#!/usr/bin/env python3
"""Per-kernel statistics of a rocprofv3 (rocpd sqlite) result as text: tools/rocpd_summary.py results.db

Two blocks: rocprofv3's own `--stats` table (every launch), and the same for the full-batch launches only (grid >= half
of that kernel's largest grid) -- a bench run also issues one single-scan set-up launch of every extract kernel (the
carry scan), which would dilute the average of 5 + 1 launches.  The second block is what bench.py's HIP-event figures
(kernel_ms_per_step, roofline.avg_launch_ms) are to be compared with."""
import sqlite3
import sys
from collections import defaultdict

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
print(f"# rocprofv3 --kernel-trace --stats summary of {sys.argv[1]} (durations in microseconds)")
print(f"{'kernel':60s} {'calls':>6s} {'total_us':>12s} {'avg_us':>10s} {'pct':>6s}")
for name, calls, total, avg, pct in cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels"):
    print(f"{name[:60]:60s} {calls:6d} {total:12.1f} {avg:10.1f} {pct:6.2f}")
rows = defaultdict(list)
for name, gx, gy, gz, dur in cur.execute("select name, grid_x, grid_y, grid_z, duration from kernels"):
    rows[name].append((int(gx) * int(gy) * int(gz), float(dur) / 1000.0))
print("\n# full-batch launches only (grid >= half of the kernel's largest grid)")
print(f"{'kernel':60s} {'calls':>6s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s}")
for name in sorted(rows, key=lambda n: -sum(d for _, d in rows[n])):
    gmax = max(g for g, _ in rows[name])
    big = [d for g, d in rows[name] if g >= 0.5 * gmax]
    print(f"{name[:60]:60s} {len(big):6d} {sum(big) / len(big):10.1f} {min(big):10.1f} {max(big):10.1f}")
