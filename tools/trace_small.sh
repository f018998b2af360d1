#!/bin/bash
# Dev tool: kernel timeline (rocprofv3 --kernel-trace) of small host-buffer calls.
# Usage (via gpurun): bash tools/trace_small.sh <tag> [reads haps kind]
set -u
TAG=${1:-small}
NR=${2:-100}; NH=${3:-10}; KIND=${4:-hc}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/trace_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cat > /tmp/small_calls.py <<PY
import sys, time, numpy as np
sys.path.insert(0, "$REPO")
from gkl_amd import native
from gkl_amd.synth import make_batch
b = make_batch("$KIND", $NR, $NH)
out = np.empty(b.n_pairs)
with native.PairHmmContext() as c:
    for _ in range(10): c.compute(b, out)
    ts = []
    for _ in range(30):
        t = time.perf_counter(); c.compute(b, out); ts.append(time.perf_counter() - t)
    print("host-buffer call median %.1f us min %.1f us" % (np.median(ts) * 1e6, np.min(ts) * 1e6))
PY
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT -o t -- python /tmp/small_calls.py > $OUT/run.txt 2>&1
python - <<PY
import csv, glob
kt = glob.glob("$OUT/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(kt)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
mc = glob.glob("$OUT/**/*memory_copy_trace.csv", recursive=True)
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60]) for r in rows]
if mc:
    for r in csv.DictReader(open(mc[0])):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "") + " " + r.get("Bytes", r.get("Size", ""))))
ev.sort()
# last call = the events after the last prep_kernel start
idx = max(i for i, e in enumerate(ev) if "prep_kernel" in e[2])
# include the copy right before it
start = idx - 1 if idx > 0 and ev[idx - 1][2].startswith("COPY") else idx
t0 = ev[start][0]
for s, e, n in ev[start:]:
    print("%8.1f us  +%7.1f us  %s" % ((s - t0) / 1e3, (e - s) / 1e3, n))
PY
cat $OUT/run.txt | tail -3
