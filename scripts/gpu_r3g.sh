cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/prof_r2
for occ in 6 5; do TAG="usmall_occ=$occ" RHIP_USMALL_OCC=$occ timeout 200 python scripts/quick_all.py weather_sept_85 census-income census1881 2>/dev/null; done
for occ in 6 5; do
  for spec in w_or:or:weather_sept_85; do
    name=${spec%%:*}; rest=${spec#*:}; op=${rest%%:*}; ds=${rest#*:}
    rm -rf gpurun_out/prof_r2/$name
    RHIP_USMALL_OCC=$occ RHIP_NO_OVERLAP=1 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_r2/$name -o p -- python scripts/prof_weather.py $op $ds > gpurun_out/prof_r2/$name.log 2>&1
    echo "occ=$occ standalone"; python scripts/show_trace.py $name | grep -E "usmall|wave|k_ba|period"
  done
done
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --tb=short -k "realdata_all_pairs or union_boundaries or synth_every" 2>&1 | tail -2
