#!/bin/bash
# Long fuzz runs of the round's last build (appended to profiles/r03_fuzz.txt): gpurun_out/final/fuzz_long.txt
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/final; mkdir -p $OUT
{
  echo "== fuzz_oracle 400 9000";        timeout -k 10 900 python tools/fuzz_oracle.py 400 9000 2>&1 | grep -v amdgpu.ids | tail -2
  echo "== fuzz_thresholds 2200 6000";   timeout -k 10 900 python tools/fuzz_thresholds.py 2200 6000 2>&1 | grep -v amdgpu.ids | tail -2
  echo "== fuzz_paths 600 50000";        timeout -k 10 600 python tools/fuzz_paths.py 600 50000 2>&1 | grep -v amdgpu.ids | tail -2
  echo "== fuzz_paths 700 8000 --wide";  timeout -k 10 900 python tools/fuzz_paths.py 700 8000 --wide 2>&1 | grep -v amdgpu.ids | tail -3
} > $OUT/fuzz_long.txt 2>&1
cat $OUT/fuzz_long.txt
