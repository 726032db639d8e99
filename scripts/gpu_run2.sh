set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -X faulthandler -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; tail -25 gpurun_out/smoke.log
./scripts/bin/bb_microbench > gpurun_out/microbench1.log 2>&1; cat gpurun_out/microbench1.log
lscpu | head -20 > gpurun_out/lscpu.log; free -g >> gpurun_out/lscpu.log
python scripts/cpu_scaling.py > gpurun_out/cpu_scaling.log 2>&1; cat gpurun_out/cpu_scaling.log
