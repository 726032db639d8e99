cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/prof_r2
for abl in 0 1; do
  rm -rf gpurun_out/prof_r2/w_or
  if [ $abl = 1 ]; then export RHIP_ABL=1; fi
  RHIP_NO_OVERLAP=1 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_r2/w_or -o p -- python scripts/prof_weather.py or weather_sept_85 > gpurun_out/prof_r2/w_or.log 2>&1
  echo "abl=$abl"; python scripts/show_trace.py w_or | grep -E "usmall"
done
