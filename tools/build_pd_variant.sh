#!/bin/bash
# Dev tool: a variant build of the PDHMM library for A/B runs (tools/ab_pd_variants.sh):
#   bash tools/build_pd_variant.sh <name> [generator knobs, e.g. allplain,nosmem] [extra compiler flags]
# -> gkl_amd/lib/libgklhip_pdhmm_<name>.so (git-ignored).  Knobs are timing experiments: wrong results.
set -e
cd "$(dirname "$0")/.."
N=$1; K=${2:-}; F=${3:-}
D=/tmp/pdv_$N; rm -rf $D; mkdir -p $D/gkl_amd/csrc $D/include
cp gkl_amd/csrc/*.h gkl_amd/csrc/*.hip $D/gkl_amd/csrc/; cp include/*.h $D/include/
PD_ASM_KNOBS=$K python3 tools/gen_pdhmm_asm.py $D/gkl_amd/csrc/pdhmm_plain_asm.h
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize $F -c $D/gkl_amd/csrc/pdhmm_api.hip -o $D/pdhmm_api.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o gkl_amd/lib/libgklhip_pdhmm_$N.so $D/pdhmm_api.o gkl_amd/lib/pairhmm_plan.o -lpthread
echo gkl_amd/lib/libgklhip_pdhmm_$N.so
