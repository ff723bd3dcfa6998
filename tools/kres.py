#!/usr/bin/env python3
"""Per-kernel resource table of one .hip source (hipcc -Rpass-analysis=kernel-resource-usage): VGPRs, AGPRs, spills, scratch,
LDS, occupancy.  usage: tools/kres.py csrc/attn.hip [name-filter]"""
import re, subprocess, sys, os
src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
inc = os.path.dirname(os.path.abspath(src))
out = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=fast", "-I", inc,
                      "-c", src, "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True).stderr
rows, cur = [], None
for line in out.splitlines():
    m = re.search(r"remark:\s+(Function Name|[A-Za-z ]+?):\s+(.*?) \[-Rpass", line)
    if not m:
        continue
    k, v = m.group(1).strip(), m.group(2).strip()
    if k == "Function Name":
        cur = {"name": v}
        rows.append(cur)
    elif cur is not None:
        cur[k] = v
def demangle(n):
    m = re.match(r"_ZN12_GLOBAL__N_1\d+([A-Za-z0-9_]+?)I(.*?)EEvNS", n)
    if not m:
        return n
    args = re.findall(r"L[ib](\d+)E", m.group(2))
    return m.group(1) + "<" + ",".join(args) + ">"
print(f"{'VGPR':>5} {'AGPR':>5} {'SGPR':>5} {'vspill':>6} {'sspill':>6} {'scratch':>8} {'LDS':>7} {'occ':>4}  kernel")
for r in rows:
    name = demangle(r["name"])
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"\(.*$", "", name)
    if flt and flt not in name:
        continue
    print(f"{r.get('VGPRs','?'):>5} {r.get('AGPRs','?'):>5} {r.get('TotalSGPRs','?'):>5} {r.get('VGPRs Spill','?'):>6} {r.get('SGPRs Spill','?'):>6} "
          f"{r.get('ScratchSize [bytes/lane]','?'):>8} {r.get('LDS Size [bytes/block]','?'):>7} {r.get('Occupancy [waves/SIMD]','?'):>4}  {name}")
