#!/bin/bash
# Dev tool: is the GPU kept busy by a pipelined computeLikelihoodsNative?  Kernel timeline (rocprofv3 --kernel-trace) of C2 calls
# through the mock JVM: busy fraction (union of kernel intervals / span), kernel time per call by kernel, for the pipelined
# default and for the same call in one shot.   Usage (via gpurun): bash tools/jni_timeline.sh <tag> [max_threads] [extra env...]
set -u
TAG=${1:-jt}; MT=${2:-1}; shift 2 || true
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/jni_timeline_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cat > /tmp/jni_calls.py <<PY
import sys, os
sys.path.insert(0, "$REPO")
from gkl_amd.synth import make_batch
from tests import mockjni
b = make_batch("hc", 10000, 128)
t, calls = [], []
rc, _, cls, msg, wall = mockjni.run_concurrent(b, 1, iters=12, warm=6, max_threads=$MT, timing=t, calls=calls)
assert rc == 0, (cls, msg)
import numpy as np
print("calls median %.3f ms; marshal %.3f wait %.3f wb %.3f" % (np.median([c[0] for c in calls]), t[0] / t[4] / 1e6, t[1] / t[4] / 1e6, t[2] / t[4] / 1e6))
PY
for MODE in pipelined oneshot; do
  if [ $MODE = oneshot ]; then export GKL_HIP_JNI_PIPELINE_PAIRS=2000000000; else unset GKL_HIP_JNI_PIPELINE_PAIRS; fi
  env "$@" rocprofv3 --kernel-trace --output-format csv -d $OUT/$MODE -o t -- python /tmp/jni_calls.py > $OUT/$MODE.txt 2>&1
  tail -1 $OUT/$MODE.txt
  python - <<PY
import csv, glob, collections
kt = glob.glob("$OUT/$MODE/**/*kernel_trace.csv", recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:48]) for r in csv.DictReader(open(kt))]
rows.sort()
# the timed calls = the last 12 of 18: cut at the biggest gaps is fragile, so take the last two thirds of the main forward kernels' launches
fwd = [i for i, r in enumerate(rows) if "fwd_stream" in r[2] and "float" in r[2]]
first = fwd[len(fwd) // 3]
ev = rows[first:]
span = ev[-1][1] - ev[0][0]
busy, cur_s, cur_e = 0, ev[0][0], ev[0][1]
for s, e, _ in ev[1:]:
    if s > cur_e:
        busy += cur_e - cur_s; cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
by = collections.Counter(); n = collections.Counter()
for s, e, k in ev:
    by[k] += e - s; n[k] += 1
print("$MODE: span %.2f ms for 12 calls = %.3f ms per call; GPU busy %.3f of it; kernel time per call %.3f ms" % (span / 1e6, span / 12e6, busy / span, sum(by.values()) / 12e6))
for k, v in by.most_common(8):
    print("   %-48s %8.3f ms per call in %5.1f launches" % (k, v / 12e6, n[k] / 12))
PY
done
