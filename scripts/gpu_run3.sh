set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -X faulthandler -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; tail -5 gpurun_out/smoke.log
timeout 900 python -m pytest tests -m gpu -q --tb=short -x > gpurun_out/pytest2.log 2>&1; tail -15 gpurun_out/pytest2.log
python bench.py --steps 10 --warmup 2 > gpurun_out/bench2.json 2> gpurun_out/bench2.err; cat gpurun_out/bench2.json; tail -5 gpurun_out/bench2.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof2 -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 1 --no-cpu > $GRAFT_REPO_ROOT/gpurun_out/prof2.log 2>&1
ls -R $GRAFT_REPO_ROOT/gpurun_out/prof2 | head -20
