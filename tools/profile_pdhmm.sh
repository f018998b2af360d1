#!/bin/bash
# Dev tool: rocprofv3 kernel stats + PMC passes of the PDHMM measurement (tests/perf_pdhmm.py) on the GPU box.
# Usage (via gpurun): bash tools/profile_pdhmm.sh <tag> [fixture-x]  -> gpurun_out/prof_pd_<tag>/...
set -u
TAG=${1:-r02}
FX=${2:-32}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_pd_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $REPO/tests/perf_pdhmm.py --reps 10 --fixture-x $FX"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- $CMD > $OUT/run.txt 2> $OUT/trace.err
i=0
for PMC in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_VMEM_WR" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d $OUT/pmc$i -o pmc -- $CMD > $OUT/pmc$i.txt 2> $OUT/pmc$i.err
done
python - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob("$OUT/pmc*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open("$OUT/pmc_summary.csv", "w") as f:
    f.write("kernel,counter,dispatches,max_per_dispatch,mean_per_dispatch\n")
    for k, cs in agg.items():
        for c, v in sorted(cs.items()):
            f.write('"%s",%s,%d,%.6g,%.6g\n' % (k[:100], c, len(v), max(v), sum(v) / len(v)))
print(open("$OUT/pmc_summary.csv").read())
PY
grep "cross entry\|fixture" $OUT/run.txt | cut -c1-200
