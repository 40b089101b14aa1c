#!/bin/bash
# The C++ application loop (uob_raytracer) over animated frames: what the reference prints per frame.
cd $GRAFT_REPO_ROOT
run() { echo "./uob_raytracer $*"; ./uob_raytracer_amd/uob_raytracer "$@" --out /tmp/shot.bmp | grep "Rendertime" | awk '{print $4}' | sort -n | awk '{a[NR]=$1} END {printf "  %d frames: median %d us, min %d, max %d\n", NR, a[int((NR+1)/2)], a[1], a[NR]}'; }
run --size 1024 --frames 200
run --size 4096 --aa 4 2 --shadows 64 --frames 100
run --size 1024 --frames 200 --copy-back
run --size 4096 --aa 4 2 --shadows 64 --frames 100 --copy-back
run --size 4096 --aa 4 2 --shadows 64 --frames 100 --devices 0,0
run --size 4096 --aa 4 2 --shadows 64 --frames 100 --devices 0,0,0,0
run --size 4096 --aa 4 2 --shadows 64 --frames 100 --devices 0,0 --copy-back
run --size 4096 --aa 4 2 --shadows 64 --frames 100 --devices 0,0,0,0 --copy-back
