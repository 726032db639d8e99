# host-phase clocks of all-pairs batches, with the polled completion word and with a blocking stream wait
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/iter
for spec in "$@"; do
  op=${spec%%:*}; ds=${spec#*:}
  for sw in 1 0; do
    echo "RHIP_SPIN_WAIT=$sw" >> gpurun_out/iter/hostclk.log
    RHIP_SPIN_WAIT=$sw python scripts/prof_weather.py $op $ds 2>&1 | grep "min ms" | tee -a gpurun_out/iter/hostclk.log
  done
done
