#!/bin/bash
# Fuzz runs of the round's last build (profiles/r03_fuzz.txt): gpurun_out/final/fuzz.txt
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/final; mkdir -p $OUT
{
  echo "== fuzz_oracle 150 5000";      timeout -k 10 400 python tools/fuzz_oracle.py 150 5000 2>&1 | grep -v amdgpu.ids | tail -4
  echo "== fuzz_thresholds 600 3000";  timeout -k 10 300 python tools/fuzz_thresholds.py 600 3000 2>&1 | grep -v amdgpu.ids | tail -3
  echo "== fuzz_paths 200 30000";      timeout -k 10 300 python tools/fuzz_paths.py 200 30000 2>&1 | grep -v amdgpu.ids | tail -3
  echo "== fuzz_paths 250 4000 --wide"; timeout -k 10 300 python tools/fuzz_paths.py 250 4000 --wide 2>&1 | grep -v amdgpu.ids | tail -3
  echo "== rt_selftest_normalize, every significand pair"; timeout -k 10 200 python -c "
import sys; sys.path.insert(0, '.')
from uob_raytracer_amd import runtime as rt
print(rt.selftest_normalize(1))" 2>&1 | grep -v amdgpu.ids
} > $OUT/fuzz.txt 2>&1
cat $OUT/fuzz.txt
