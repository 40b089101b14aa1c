#!/bin/bash
# The round's final measurements on the GPU box, everything that ends up under profiles/r03_*:
#   bash tools/final_r03.sh        (run from the repo root through gpurun; writes gpurun_out/final/)
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/final; mkdir -p $OUT
export TMPDIR=/tmp
step() { echo "=== $1"; shift; timeout -k 10 "$@" ; echo "=== exit $?"; }
step tests 500 python -m pytest tests -x -q -m gpu > $OUT/tests.log 2>&1; tail -3 $OUT/tests.log
# counters of every workload on THIS build, then the bench lines that price them
for w in headline cfg2 cfg3 cfg5 reference; do
  extra=""; [ $w != headline ] && extra="--workload $w"
  timeout -k 10 400 bash tools/pmc_passes.sh final/pmc_$w $extra > $OUT/pmc_$w.log 2>&1
  python tools/pmc_to_json.py $OUT/pmc_$w $w profiles/r03_pmc.json > $OUT/pmc_json_$w.log 2>&1 || echo "pmc_to_json $w failed"
  find $OUT/pmc_$w -name "*.db" -delete
done
cp profiles/r03_pmc.json profiles/r03_kernel_stats_*.csv $OUT/
timeout -k 10 400 python bench.py > $OUT/bench_n1.json 2> $OUT/bench_n1.err; tail -c 300 $OUT/bench_n1.json; echo
for w in cfg2 cfg3 cfg5 reference; do timeout -k 10 300 python bench.py --workload $w > $OUT/bench_$w.json 2> $OUT/bench_$w.err; done
timeout -k 10 200 python bench.py --force-collective --no-cpu-baseline > $OUT/bench_rccl_gather_1rank.json 2> $OUT/bench_rccl.err
timeout -k 10 300 python tools/emulate_ranks.py 16 > $OUT/emulated_ranks.txt 2>&1; tail -4 $OUT/emulated_ranks.txt
CONTEXTS="2 2" timeout -k 10 200 python tools/band_pipeline.py 8 4 2 1 > $OUT/band_time.txt 2>&1
CONTEXTS="1 1" timeout -k 10 200 python tools/band_pipeline.py 8 4 2 1 >> $OUT/band_time.txt 2>&1
timeout -k 10 200 python tools/light_sweep.py > $OUT/light_sweep.txt 2>&1
timeout -k 10 200 python tools/wave_timeline.py > $OUT/wave_timeline.txt 2>&1
timeout -k 10 300 bash tools/host_anim.sh > $OUT/host_animation.txt 2>&1
timeout -k 10 200 python tools/mesh_first_frame.py > $OUT/mesh.txt 2>&1; NOSPH=0 timeout -k 10 200 python tools/mesh_first_frame.py >> $OUT/mesh.txt 2>&1
echo "final run done"; ls $OUT | head -50
