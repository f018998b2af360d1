#!/bin/bash
# Dev tool: build timing-ablation variants of the library (WRONG results, timing only).
cd "$(dirname "$0")/../gkl_amd/csrc"
FL="--offload-arch=gfx950 -fgpu-flush-denormals-to-zero -Xarch_device -fdenormal-fp-math=preserve-sign -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -shared"
for v in "$@"; do
  /opt/rocm/bin/hipcc $FL -DGKL_ABL=$v pairhmm_api.hip pairhmm_tables.cpp pairhmm_plan.cpp -o ../lib/var_abl$v.so -lpthread &
done
wait
ls ../lib/var_*
