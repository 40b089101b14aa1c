for jt in 4 8; do echo "== UOB_RT_JOB_TASKS=$jt"; UOB_RT_JOB_TASKS=$jt python tools/band_pipeline.py 8 2; done
echo "== default"; python tools/band_pipeline.py 8 4 2 1
