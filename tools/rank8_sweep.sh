#!/bin/bash
# One rank of eight (512 rows of the headline frame), render alone on two contexts: job size, grid and list threshold knobs.
cd $GRAFT_REPO_ROOT
run() { echo "--- $*"; env "$@" CONTEXTS="2 2" timeout -k 10 120 python tools/band_pipeline.py 8 2>&1 | grep ranks; }
run X=0
run UOB_RT_JOB_TASKS=2
run UOB_RT_JOB_TASKS=8
run UOB_RT_GRID_PER_CU=3
run UOB_RT_GRID_PER_CU=5
run UOB_RT_GRID_PER_CU=5 UOB_RT_JOB_TASKS=2
run UOB_RT_HEAVY_FACTOR4=6
run UOB_RT_HEAVY_FACTOR4=12
run UOB_RT_SPLIT_LISTED=1
run UOB_RT_PLAIN_ORDER=1
