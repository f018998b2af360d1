for f in 2 3 4; do echo "flights $f"; GKL_HIP_COMBINE_FLIGHTS=$f python tools/fused_ab.py jni 2>&1 | grep callers; done
