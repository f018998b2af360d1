for pad in 0 8192 14336; do echo "== pad $pad"; for i in 1 2 3; do GKLHIP_POLICY_LDS_PAD=$pad python tools/jni_concurrency.py 16 2>&1 | grep callers | cut -c1-60; done; done
for pad in 0 8192; do echo "== trace pad $pad"; GKLHIP_POLICY_LDS_PAD=$pad GKL_HIP_COMBINE_FLIGHTS=1 bash tools/trace_concurrency.sh 16 2>&1 | grep -E "multi|window"; done
