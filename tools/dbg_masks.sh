for m in 0 1 2 4; do echo "== MASK_DEBUG=$m"; UOB_RT_MASK_DEBUG=$m python tools/fuzz_debug.py 1519 2>&1 | grep "default"; done
