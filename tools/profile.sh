#!/bin/bash
# Dev tool: rocprofv3 kernel-trace stats + PMC passes of bench.py on the GPU box.
# Usage (via gpurun): bash tools/profile.sh <tag>   -> gpurun_out/prof_<tag>/...
set -u
TAG=${1:-r01}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $REPO/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- $CMD > $OUT/bench_under_trace.json 2> $OUT/trace.err
CMDS="python $REPO/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras"
i=0
for PMC in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM"; do
  i=$((i+1))
  rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d $OUT/pmc$i -o pmc -- $CMDS > $OUT/pmc$i.json 2> $OUT/pmc$i.err
done
find $OUT -name "*.csv" | head -30
