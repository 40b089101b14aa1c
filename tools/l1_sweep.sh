# headline frame time against the width of level 1's point set (UOB_RT_L1_INFLATE), round-3 experiment
cd $GRAFT_REPO_ROOT
for v in 1 2.5 3.5 4.5 5.5 7.5 9; do echo -n "l1_inflate $v: "; UOB_RT_L1_INFLATE=$v python tools/ab_time.py uob_raytracer_amd/libuob_rt.so 2>&1 | tail -1 | cut -d' ' -f2-; done
