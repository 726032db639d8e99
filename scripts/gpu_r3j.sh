cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/prof_r2
for ds in census1881 weather_sept_85; do
  rm -rf gpurun_out/prof_r2/m_$ds
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_r2/m_$ds -o p -- python scripts/prof_multi.py $ds > gpurun_out/prof_r2/m_$ds.log 2>&1
  grep "min ms" gpurun_out/prof_r2/m_$ds.log
  python scripts/show_trace.py m_$ds
done
