cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r4a
rocprofv3 -L > gpurun_out/r4a/counters.txt 2>&1
timeout 400 scripts/bin/arena_place 6 > gpurun_out/r4a/arena_place.txt 2>&1
tail -30 gpurun_out/r4a/arena_place.txt
TAG=base timeout 200 python scripts/quick_all.py 2>/dev/null | tee gpurun_out/r4a/quick_all.txt
