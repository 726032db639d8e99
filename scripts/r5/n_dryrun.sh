cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r5n
mkdir -p $O
timeout 1400 python -m pytest tests/test_gpu_bench_dryrun.py -q -m gpu > $O/tests.txt 2>&1; tail -30 $O/tests.txt | cut -c1-400
