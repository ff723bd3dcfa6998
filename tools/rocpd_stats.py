#!/usr/bin/env python3
"""Per-kernel statistics (the rocprofv3 --stats CSV columns) from a rocprofv3 rocpd database:
    rocpd_stats.py results.db [out.csv] [steps]        steps: timed+warm-up steps in the trace, for a per-step column"""
import csv, re, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
rows = db.execute("select name, (end - start) from kernels").fetchall()
agg = {}
for name, d in rows:
    a = agg.setdefault(name, [0, 0, 1 << 62, 0])
    a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
total = sum(a[1] for a in agg.values())
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 0
out = sorted(agg.items(), key=lambda kv: -kv[1][1])
w = csv.writer(open(sys.argv[2], "w", newline="")) if len(sys.argv) > 2 else None
hdr = ["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"] + (["CallsPerStep", "UsPerStep"] if steps else [])
if w: w.writerow(hdr)
for name, (n, t, mn, mx) in out:
    row = [name, n, t, round(t / n, 1), round(100.0 * t / total, 3), mn, mx] + ([round(n / steps, 2), round(t / steps / 1e3, 2)] if steps else [])
    if w: w.writerow(row)
short = lambda s: re.sub(r"\(anonymous namespace\)::", "", s)[:110]
print(f"{len(rows)} dispatches, {total/1e6:.2f} ms of kernel time" + (f", {total/steps/1e6:.3f} ms and {len(rows)/steps:.0f} launches per step" if steps else ""))
for name, (n, t, mn, mx) in out[:28]:
    print(f"{100.0*t/total:6.2f}%  {n:6d} x {t/n/1e3:8.1f} us  {short(name)}")
