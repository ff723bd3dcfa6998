#!/usr/bin/env python3
"""Whole-step HBM-side traffic from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over bench.py (--no-graph, a few steps):
    step_pmc_parse.py <dir with FETCH_SIZE/ and WRITE_SIZE/ outputs> <out.json> <label>
Per step = total over all dispatches / number of sgd_flat_kernel dispatches.  FETCH_SIZE is doubled (gfx950 tallies 128-B read
requests at 64 B: MI355X_MICROARCH.md, HBM section); counters are in KB.  Calibrated for 16-B-per-lane accesses; the 8-B bf16
loads of the register paths are not (ratios between the two modes are what this is for)."""
import csv, glob, json, os, re, sys
src, out, label = sys.argv[1:4]
res = {"label": label, "note": "rocprofv3 --pmc, separate passes; FETCH_SIZE x2; bytes per training step"}
top = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    files = glob.glob(os.path.join(src, c, "**", "*counter_collection.csv"), recursive=True)
    assert files, f"no counter csv for {c}"
    tot, steps, per = 0.0, 0, {}
    with open(files[0]) as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] != c:
                continue
            v = float(row["Counter_Value"]) * 1024 * (2 if c == "FETCH_SIZE" else 1)
            k = re.sub(r"\(anonymous namespace\)::|void ", "", row["Kernel_Name"]); k = re.sub(r"[(<].*", "", k)
            tot += v; per[k] = per.get(k, 0.0) + v
            steps += "sgd_flat_kernel" in row["Kernel_Name"]
    res[c.lower() + "_bytes_per_step"] = int(tot / max(steps, 1))
    res["steps_in_trace"] = steps
    for k, v in per.items():
        top.setdefault(k, {})[c] = int(v / max(steps, 1))
res["total_bytes_per_step"] = res["fetch_size_bytes_per_step"] + res["write_size_bytes_per_step"]
res["per_kernel_bytes_per_step"] = dict(sorted(top.items(), key=lambda kv: -sum(kv[1].values()))[:14])
json.dump(res, open(out, "w"), indent=1)
print(label, {k: v for k, v in res.items() if k.endswith("per_step") and not k.startswith("per_kernel")})
