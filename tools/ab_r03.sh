# A/B of the round-3 kernel changes on the GPU box: bash tools/ab_r03.sh
cd $GRAFT_REPO_ROOT
run() { python bench.py --steps 300 --warmup 80 --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4f ms/step kernel %.4f checksum %d' % (d['ms_per_step'], d['kernel_ms_per_launch'], d['frame_checksum']))"; }
echo "N=1 specialised:"; run --timed-only
echo "N=1 generic:"; UOB_RT_NO_SPECIALISE=1 run --timed-only
echo "N=1 specialised again:"; run --timed-only
for n in 4 8; do for br in 8 16 32; do echo "N=$n band $br rank 0:"; run --emulate-rank 0/$n --band-rows $br; echo "N=$n band $br rank 1:"; run --emulate-rank 1/$n --band-rows $br; done; done
for w in cfg2 cfg3 reference; do echo "$w:"; run --timed-only --workload $w; echo "$w generic:"; UOB_RT_NO_SPECIALISE=1 run --timed-only --workload $w; done
