#!/usr/bin/env python3
"""The kernel sequence of the LAST training step in a rocprofv3 rocpd database (between the last two sgd_flat_kernel dispatches):
    rocpd_last_step.py results.db [out.txt]
Prints launches per step, the per-kernel counts of that one step and (to out.txt) the ordered list with start offsets and durations."""
import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, start, end from kernels order by start").fetchall()
short = lambda s: re.sub(r"\(anonymous namespace\)::", "", s)
sgd = [i for i, r in enumerate(rows) if "sgd_flat_kernel" in r[0]]
a, b = sgd[-2] + 1, sgd[-1] + 1
step = rows[a:b]
t0 = step[0][1]
print(f"last step: {len(step)} launches, {(step[-1][2] - t0) / 1e6:.3f} ms from first start to last end, "
      f"{sum(e - s for _, s, e in step) / 1e6:.3f} ms of kernel time")
cnt = {}
for n, s, e in step:
    k = re.sub(r"\(.*", "", short(n))[:100]
    c = cnt.setdefault(k, [0, 0])
    c[0] += 1; c[1] += e - s
for k, (n, t) in sorted(cnt.items(), key=lambda kv: -kv[1][1]):
    print(f"{n:4d} x {t / n / 1e3:8.1f} us  {k}")
if len(sys.argv) > 2:
    with open(sys.argv[2], "w") as f:
        prev_end = t0
        for n, s, e in step:
            f.write(f"{(s - t0) / 1e3:10.1f} us  +{(e - s) / 1e3:7.1f}  gap {(s - prev_end) / 1e3:6.1f}  {re.sub(r'[(].*', '', short(n))[:110]}\n")
            prev_end = max(prev_end, e)
