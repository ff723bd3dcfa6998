#!/bin/bash
# HBM-side traffic of the stripe-attention kernels, one stage and one counter per rocprofv3 pass (FETCH_SIZE and WRITE_SIZE
# cannot share a pass: MI355X_MICROARCH.md, counter slots).  Run from the repo root on the GPU box:
#   bash tools/attn_pmc.sh && python3 tools/attn_pmc_parse.py gpurun_out/pmc profiles/round3_attn_pmc.json
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/pmc
for s in 1 2 3 4; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc/s${s}_$c -o out -- python3 tools/attn_one.py $s 3 > gpurun_out/pmc/s${s}_$c.log 2>&1
    echo "stage $s $c done"
  done
done
