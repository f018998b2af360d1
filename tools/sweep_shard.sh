#!/bin/bash
# Dev tool: the eighth-shard (1250 x 128) pipelined device-resident step for planner knobs; one line per setting.
cd ${GRAFT_REPO_ROOT:-.}
run() { # label, env...
  local label="$1"; shift
  local out=$(env "$@" python bench.py --reads 1250 --no-extras --no-cpu-baseline --steps 300 --warmup 30 $EXTRA 2>/dev/null)
  python - "$label" "$out" <<'PY'
import json, sys
d = json.loads(sys.argv[2])
k = d["kernels_ms"]
print(f"{sys.argv[1]:44s} step {d['ms_per_step']:.3f} ms  main {k['fwd_main']:.3f}  fb {k['fwd_fp64_fallback']:.3f}  fixed {d.get('fixed_cost_ms')}  plan {d.get('plan')}")
PY
}
for ov in "" "--overlap"; do
  EXTRA="$ov"
  echo "== $ov"
  run "default" A=1
  for tc in 1024 1536 3072; do run "TARGET_COLS=$tc" GKLHIP_TARGET_COLS=$tc; done
  for wj in 2048 8192 16384; do run "WANTED_JOBS=$wj" GKLHIP_WANTED_JOBS=$wj; done
  for fj in 4096 6144 24576; do run "FB_WANTED_JOBS=$fj" GKLHIP_FB_WANTED_JOBS=$fj; done
  for pb in 16 32 128; do run "PLAN_BLOCKS=$pb" GKLHIP_PLAN_BLOCKS=$pb; done
done
