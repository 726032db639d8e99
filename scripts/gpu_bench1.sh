set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; tail -3 gpurun_out/smoke.log
python bench.py --steps 6 --warmup 2 > gpurun_out/bench1.json 2> gpurun_out/bench1.err; cat gpurun_out/bench1.json; tail -5 gpurun_out/bench1.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof1 -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 1 --no-cpu > $GRAFT_REPO_ROOT/gpurun_out/prof1.log 2>&1
ls -R $GRAFT_REPO_ROOT/gpurun_out/prof1 | head -20
