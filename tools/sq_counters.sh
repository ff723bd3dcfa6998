#!/bin/bash
# Matrix-pipe busy time and wave-state split per kernel over a few eager training steps (one rocprofv3 --pmc pass of SQ / GRBM
# counters; no HBM counters in the same pass).  Run from the repo root on the GPU box:
#   bash tools/sq_counters.sh fp32|bf16 && python3 tools/sq_counters_parse.py gpurun_out/sq/<mode> profiles/round3_sq_counters_<mode>.txt
set -e
mode=${1:-fp32}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/sq
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/sq/$mode -o out -- python3 bench.py --no-graph --steps 2 --warmup 1 --matmul $mode --skip-cpu --skip-bf16 --skip-roofline > gpurun_out/sq/$mode.log 2>&1
echo "collected gpurun_out/sq/$mode"
