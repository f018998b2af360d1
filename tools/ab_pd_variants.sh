#!/bin/bash
# Dev tool: PDHMM cross entry point (fixture x32) with several variant libraries, alternating on one box.
# Usage (via gpurun): bash tools/ab_pd_variants.sh "<suffix> <suffix> ..."   ("" = the built library)
cd ${GRAFT_REPO_ROOT:-.}
for rep in 1 2 3; do for v in $1; do
  [ "$v" = "built" ] && lib=gkl_amd/lib/libgklhip_pdhmm.so || lib=gkl_amd/lib/libgklhip_pdhmm_$v.so
  echo -n "$v: "; GKL_AMD_PDHMM_LIB=$lib python tests/perf_pdhmm.py --reps 5 --fixture-x ${2:-32} 2>&1 | grep "cross entry" | grep -o "kernel [0-9.]* ms"
done; done
