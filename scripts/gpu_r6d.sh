cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r6d
mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q --tb=short -x -k "64bit or c5 or many or distributed or join or alloc or compat or sharded or dropin_harness" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
for d in 0 1; do RHIP_MANY_DICT=$d timeout 120 python scripts/prof_c5_union.py 2>/dev/null | tail -1; done
timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c5u -o p -- python scripts/prof_c5_union.py > $O/prof_c5u.log 2>&1
python scripts/trace_many.py $O/prof_c5u "c5 union, key dictionary" 2>/dev/null | tail -14
rm -f $(find $O -name "*kernel_trace.csv")
echo done
