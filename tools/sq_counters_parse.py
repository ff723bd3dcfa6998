#!/usr/bin/env python3
"""Per-kernel matrix-pipe utilisation and wave-state split from the counter pass of tools/sq_counters.sh:
    sq_counters_parse.py <dir with out_counter_collection.csv> <out.txt>
mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (kernel duration x 1024 SIMDs x 2.4 GHz): the counter is in shader cycles summed over the
chip's SIMDs (MI355X_MICROARCH.md, per-instruction table); durations are the dispatch timestamps; 2.4 GHz is the clock behind the
peaks bench.py prices against (under fp32 MFMA load the chip runs at ~2.05 GHz, so a pipe that never idles would read ~0.85 here).
GRBM_GUI_ACTIVE is collected too but not used as the time base: for kernels of tens of microseconds it includes ~5 us of dispatch
outside the timestamps.
wait / stall / active = SQ_WAIT_ANY (parked on s_waitcnt or a barrier), SQ_WAIT_INST_ANY (issue stall), SQ_ACTIVE_INST_ANY as
fractions of SQ_WAVE_CYCLES."""
import collections, csv, glob, os, re, sys
src, out = sys.argv[1:3]
f = glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True)[0]
per = collections.defaultdict(lambda: collections.defaultdict(float)); dur = collections.defaultdict(float); n = collections.defaultdict(int)
seen = set()
for r in csv.DictReader(open(f)):
    k = re.sub(r"\(anonymous namespace\)::|void ", "", r["Kernel_Name"]); k = re.sub(r"\(.*", "", k)
    per[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Dispatch_Id"] not in seen:
        seen.add(r["Dispatch_Id"]); dur[k] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); n[k] += 1
ghz = 2.4
lines = [f"# {os.path.basename(src)}: eager training steps under rocprofv3 --pmc (SQ + GRBM pass); mfma_busy against 1024 SIMDs x {ghz} GHz x dispatch duration",
         f"# {'kernel':78s} {'launches':>8s} {'avg us':>8s} {'mfma_busy':>9s} {'wait':>6s} {'stall':>6s} {'active':>6s}"]
tot_d = sum(dur.values())
for k in sorted(per, key=lambda k: -dur[k])[:24]:
    c = per[k]; wc = c["SQ_WAVE_CYCLES"] or 1.0
    lines.append(f"  {k[:78]:78s} {n[k]:8d} {dur[k] / n[k] / 1e3:8.1f} {c['SQ_VALU_MFMA_BUSY_CYCLES'] / (dur[k] * ghz * 1024):9.3f} "
                 f"{c['SQ_WAIT_ANY'] / wc:6.2f} {c['SQ_WAIT_INST_ANY'] / wc:6.2f} {c['SQ_ACTIVE_INST_ANY'] / wc:6.2f}")
fam = collections.defaultdict(lambda: [0.0, 0.0])
for k in per:
    g = "attention" if "attn" in k else ("GEMM family" if ("gemm" in k or "wgrad" in k) else "other")
    fam[g][0] += per[k]["SQ_VALU_MFMA_BUSY_CYCLES"]; fam[g][1] += dur[k]
lines.append("# by family: " + "; ".join(f"{g}: {100 * d / tot_d:.0f} % of kernel time, matrix pipe busy {b / (d * ghz * 1024):.3f}" for g, (b, d) in fam.items()))
lines.append(f"# whole trace: matrix pipe busy {sum(b for b, _ in fam.values()) / (tot_d * ghz * 1024):.3f} of kernel time")
open(out, "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
