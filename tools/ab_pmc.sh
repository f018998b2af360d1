#!/bin/bash
# Dev tool: PMC A/B of the forward kernels under an environment switch.
# Usage (via gpurun): bash tools/ab_pmc.sh VAR "0 1" [kernel-substring] [extra bench args]
set -u
VAR=${1:-GKLHIP_ASM_GENERAL}; VALS=${2:-"0 1"}; KSUB=${3:-fwd_stream}; EXTRA=${4:-}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/ab_pmc
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMDS="python $REPO/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras $EXTRA"
for V in $VALS; do
  i=0
  for PMC in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
             "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU"; do
    i=$((i+1))
    env $VAR=$V rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d $OUT/v${V}_p$i -o pmc -- $CMDS > $OUT/v${V}_p$i.json 2> $OUT/v${V}_p$i.err
  done
done
python3 - "$OUT" "$KSUB" $VALS <<'PY'
import csv, glob, sys, os
from collections import defaultdict
out, ksub, vals = sys.argv[1], sys.argv[2], sys.argv[3:]
res = {}
for v in vals:
    agg = defaultdict(list)
    for p in glob.glob(os.path.join(out, f"v{v}_p*", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(p)):
            if ksub in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    res[v] = {k: sum(x) / len(x) for k, x in agg.items()}
names = sorted(set().union(*[set(r) for r in res.values()]))
print("counter".ljust(24) + "".join(f"{v:>16}" for v in vals))
for n in names:
    print(n.ljust(24) + "".join(f"{res[v].get(n, float('nan')):16.5g}" for v in vals))
PY
