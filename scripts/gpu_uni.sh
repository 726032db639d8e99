cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/ivl
for i in 1 2; do MULTI=0 timeout 200 python scripts/quick_all.py 2>/dev/null; done
for ds in weather_sept_85; do
  rm -rf gpurun_out/ivl/$ds
  RHIP_NO_OVERLAP=1 timeout 200 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/ivl/$ds -o p -- python scripts/per_kernel_c3.py $ds > gpurun_out/ivl/$ds.out 2> gpurun_out/ivl/$ds.err
  python scripts/join_per_kernel.py gpurun_out/ivl/$ds.out gpurun_out/ivl/$ds | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['op'], d['kernel'], d['items'], d['us_standalone'])"
done
