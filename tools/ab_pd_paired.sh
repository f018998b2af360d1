#!/bin/bash
# Dev tool: PDHMM paired entry point (computePDHMMNative's layout, fixture x32) with several variant libraries
# (tools/build_pd_variant.sh), alternating on one box.  Usage (via gpurun): bash tools/ab_pd_paired.sh "<suffix> ..."  ("built" = the built library)
cd ${GRAFT_REPO_ROOT:-.}
for rep in 1 2 3; do for v in $1; do
  [ "$v" = "built" ] && lib=gkl_amd/lib/libgklhip_pdhmm.so || lib=gkl_amd/lib/libgklhip_pdhmm_$v.so
  echo -n "$v: "; GKL_AMD_PDHMM_LIB=$lib python tests/perf_pdhmm.py --reps 5 --fixture-x ${2:-32} 2>&1 | grep "fixture pdhmm_new x" | grep -o "kernel [0-9.]* ms.*host-to-host [0-9.]* ms"
done; done
