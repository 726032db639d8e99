cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/prof_r2
TAG=staged timeout 200 python scripts/quick_all.py 2>/dev/null
rm -rf gpurun_out/prof_r2/w_or
RHIP_NO_OVERLAP=1 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_r2/w_or -o p -- python scripts/prof_weather.py or weather_sept_85 > gpurun_out/prof_r2/w_or.log 2>&1
python scripts/show_trace.py w_or | grep -E "usmall|wave|k_ba|period"
timeout 400 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --tb=short -k "realdata_all_pairs or union_boundaries or synth_every or randomized" 2>&1 | tail -2
