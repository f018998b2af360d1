#!/bin/bash
# Dev tool: kernel timeline of N concurrent C-ABI callers (100 x 10 regions): how many kernels overlap on the device.
N=${1:-16}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/trace_conc$N
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT -o t -- python $REPO/tools/cabi_concurrency.py $N > $OUT/run.txt 2>&1
python - <<PY
import csv, glob
rows = list(csv.DictReader(open(glob.glob("$OUT/**/*kernel_trace.csv", recursive=True)[0])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:40], r.get("Queue_Id", "")) for r in rows)
ev = ev[len(ev) // 2: len(ev) // 2 + 3000]      # a steady-state window
t0, t1 = ev[0][0], ev[-1][1]
busy = sum(e - s for s, e, _, _ in ev)
print("window %.1f us, %d kernels, sum of kernel times %.1f us -> mean concurrency %.2f" % ((t1 - t0) / 1e3, len(ev), busy / 1e3, busy / (t1 - t0)))
import collections
d = collections.defaultdict(list)
for s, e, n, q in ev: d[n].append((e - s) / 1e3)
for n, v in d.items(): print("  %-42s n=%5d mean %.1f us" % (n, len(v), sum(v) / len(v)))
print("queues:", collections.Counter(q for _, _, _, q in ev))
PY
tail -2 $OUT/run.txt
