# one rank's bands of an N-rank job: share of the jobs taken statically (UOB_RT_STATIC_PCT)
for n in 8 1; do for pct in 0 50 75 90 100; do
  echo -n "bands $n static_pct $pct: "; UOB_RT_STATIC_PCT=$pct AB_BANDS=$n python tools/ab_time.py uob_raytracer_amd/libuob_rt.so | tail -1 | cut -d' ' -f2-
done; done
echo -n "bands 8 static 75 full grid: "; UOB_RT_FULL_GRID=1 UOB_RT_STATIC_PCT=75 AB_BANDS=8 python tools/ab_time.py uob_raytracer_amd/libuob_rt.so | tail -1 | cut -d' ' -f2-
echo -n "bands 8 static 75 jt 8: "; UOB_RT_JOB_TASKS=8 UOB_RT_STATIC_PCT=75 AB_BANDS=8 python tools/ab_time.py uob_raytracer_amd/libuob_rt.so | tail -1 | cut -d' ' -f2-
