# A/B of the pool slot granule (RHIP_POOL_ALIGN = 16 / 128) on the many-way path: kernel timelines of C4, whole-call
# timings of C4 and C4 x 10, realdata timings, and the layout test.   gpurun --timeout 900 -- 'bash scripts/gpu_align.sh'
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/align
mkdir -p $O
for a in 16 128; do
  RHIP_POOL_ALIGN=$a timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$a -o p -- python scripts/prof_c4.py 100000 > $O/prof_$a.log 2>&1
  tail -1 $O/prof_$a.log | cut -c1-200
  python scripts/trace_many.py $O/prof_$a "align $a"; rm -f $(find $O/prof_$a -name "*kernel_trace.csv")
done
for a in 16 128; do echo "== x10 align $a"; RHIP_POOL_ALIGN=$a timeout 300 python scripts/prof_c4.py 1000000 2>&1 | tail -1 | cut -c1-200; done
for a in 16 128; do echo "== realdata align $a"; RHIP_POOL_ALIGN=$a LIST=1 MULTI=0 timeout 200 python scripts/quick_all.py 2>/dev/null | tail -30; done
timeout 600 python -m pytest tests -q -m gpu -k "payload_layout or sparse or c4" > $O/tests.txt 2>&1; grep -E "passed|failed|^E  " $O/tests.txt | tail -12 | cut -c1-300
