# iteration pass: parity subset, realdata timings, kernel timelines of the batches named in "$@" (name:op:dataset)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/iter gpurun_out/prof_r2
timeout 400 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --tb=short -k "${PYTEST_K:-probe or synth_every or edge or randomized or union_boundaries or extremes or weather}" > gpurun_out/iter/pytest.log 2>&1; tail -3 gpurun_out/iter/pytest.log | head -2
timeout 200 python scripts/quick_c3.py > gpurun_out/iter/quick_c3.jsonl 2> gpurun_out/iter/quick.err; cat gpurun_out/iter/quick_c3.jsonl
for spec in "$@"; do
  name=${spec%%:*}; rest=${spec#*:}; op=${rest%%:*}; ds=${rest#*:}
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r2/$name -o p -- python scripts/prof_weather.py $op $ds > gpurun_out/prof_r2/$name.log 2>&1
  grep "min ms" gpurun_out/prof_r2/$name.log
done
