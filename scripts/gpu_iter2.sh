cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/iter gpurun_out/prof_r2
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --tb=short -k "probe or synth_every or edge or randomized or union_boundaries or extremes or bitset_only" > gpurun_out/iter/pytest.log 2>&1; tail -3 gpurun_out/iter/pytest.log | head -2
timeout 200 python scripts/quick_c3.py > gpurun_out/iter/quick_c3.jsonl 2> gpurun_out/iter/quick.err; cat gpurun_out/iter/quick_c3.jsonl
timeout 300 python bench.py --no-cpu --no-secondary --steps 10 > gpurun_out/iter/bench.json 2> gpurun_out/iter/bench.err; python -c "
import json; d=json.loads(open('gpurun_out/iter/bench.json').read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_ms'])"
timeout 200 python scripts/bench_c2_ops.py > gpurun_out/iter/c2_ops.jsonl 2>/dev/null; cat gpurun_out/iter/c2_ops.jsonl
for spec in "$@"; do
  name=${spec%%:*}; rest=${spec#*:}; op=${rest%%:*}; ds=${rest#*:}
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r2/$name -o p -- python scripts/prof_weather.py $op $ds > gpurun_out/prof_r2/$name.log 2>&1
  grep "min ms" gpurun_out/prof_r2/$name.log
done
