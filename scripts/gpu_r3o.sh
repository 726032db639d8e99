cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/ivl
timeout 300 python -m pytest tests/test_gpu_parity.py -q -x -k "long_interval or tiny_interval or synth_every or realdata_all_pairs" 2>&1 | tail -2
PAIRS="R100 x R100,R100 x A874" timeout 200 python scripts/quick_classes.py 2>/dev/null | tail -2 | tee gpurun_out/ivl/classes.jsonl
MULTI=0 timeout 200 python scripts/quick_all.py 2>/dev/null
for ds in c5 weather_sept_85; do
  rm -rf gpurun_out/ivl/$ds
  RHIP_NO_OVERLAP=1 timeout 200 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/ivl/$ds -o p -- python scripts/per_kernel_c3.py $ds > gpurun_out/ivl/$ds.out 2> gpurun_out/ivl/$ds.err
  python scripts/join_per_kernel.py gpurun_out/ivl/$ds.out gpurun_out/ivl/$ds | grep "genw\|ivl<64" | tee gpurun_out/ivl/$ds.jsonl
done
