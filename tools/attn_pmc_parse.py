#!/usr/bin/env python3
"""rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of tools/attn_pmc.sh -> bytes per launch of the attention kernels.
FETCH_SIZE is doubled (gfx950 tallies 128-B read requests at 64 B: MI355X_MICROARCH.md, HBM section); the counters are in KB."""
import csv, glob, json, os, sys
src, out = sys.argv[1], sys.argv[2]
B, E = 24, 64
res = {"note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes (tools/attn_pmc.sh -> tools/attn_one.py, B=24); "
               "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half of wide coalesced reads); KB*1024",
       "per_launch": {}}
for s in (1, 2, 3, 4):
    L, C = (56 >> (s - 1)) ** 2, E << (s - 1)
    raw = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        files = glob.glob(os.path.join(src, f"s{s}_{c}", "**", "*counter_collection.csv"), recursive=True)
        assert files, f"no counter csv for stage {s} {c}"
        acc = {}
        with open(files[0]) as f:
            for row in csv.DictReader(f):
                if row["Counter_Name"] != c:
                    continue
                k = row["Kernel_Name"]
                name = "fwd" if "attn_fwd" in k else "bwd" if ("attn_bwd" in k or "attn_delta" in k or "lepe_wgrad" in k) else \
                       "slab_reduce" if "rows_sum" in k else None
                if name:
                    acc.setdefault(name, []).append(float(row["Counter_Value"]))
        for name, v in acc.items():
            # per launch of the C-ABI entry point: kernels of one entry point are summed, repetitions averaged (3 reps)
            raw[f"{name}.{c}"] = round(sum(v) / 3.0, 1)
    fwd = 2 * raw.get("fwd.FETCH_SIZE", 0) + raw.get("fwd.WRITE_SIZE", 0)
    bwd = 2 * (raw.get("bwd.FETCH_SIZE", 0) + raw.get("slab_reduce.FETCH_SIZE", 0)) + raw.get("bwd.WRITE_SIZE", 0) + raw.get("slab_reduce.WRITE_SIZE", 0)
    res["per_launch"][f"stage{s}"] = {"fwd_bytes": int(fwd * 1024), "bwd_bytes": int(bwd * 1024),
                                      "algorithmic_fwd_bytes": 16 * L * C * B, "algorithmic_bwd_bytes": 28 * L * C * B, "raw_kb": raw}
with open(out, "w") as f:
    json.dump(res, f, indent=1)
for k, v in res["per_launch"].items():
    print(k, "fwd %.2fx" % (v["fwd_bytes"] / v["algorithmic_fwd_bytes"]), "bwd %.2fx" % (v["bwd_bytes"] / v["algorithmic_bwd_bytes"]))
