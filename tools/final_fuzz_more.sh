#!/bin/bash
# A second set of seeds on the round's last build: gpurun_out/final/fuzz_more.txt
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/final; mkdir -p $OUT
{
  echo "== fuzz_oracle 300 20000";        timeout -k 10 900 python tools/fuzz_oracle.py 300 20000 2>&1 | grep -v amdgpu.ids | tail -1
  echo "== fuzz_paths 500 9000 --wide";   timeout -k 10 900 python tools/fuzz_paths.py 500 9000 --wide 2>&1 | grep -v amdgpu.ids | tail -2
  echo "== fuzz_thresholds 1000 9000";    timeout -k 10 600 python tools/fuzz_thresholds.py 1000 9000 2>&1 | grep -v amdgpu.ids | tail -1
} > $OUT/fuzz_more.txt 2>&1
cat $OUT/fuzz_more.txt
