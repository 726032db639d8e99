cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/prof_r2
for m in 1 0; do TAG="merge=$m" RHIP_MERGE_CLASSES=$m MULTI=1 timeout 200 python scripts/quick_all.py 2>/dev/null; done
for spec in c1_and:and:census1881 wk_or:or:wikileaks-noquotes c5_and:and:c5; do
  name=${spec%%:*}; rest=${spec#*:}; op=${rest%%:*}; ds=${rest#*:}
  rm -rf gpurun_out/prof_r2/$name
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_r2/$name -o p -- python scripts/prof_weather.py $op $ds > gpurun_out/prof_r2/$name.log 2>&1
  python scripts/show_trace.py $name
done
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --tb=short -k "realdata_all_pairs or explicit_unit or synth_every or multi or edge" 2>&1 | tail -2
