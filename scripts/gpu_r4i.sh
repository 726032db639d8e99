cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4i
mkdir -p $O
# L2 behaviour of the class kernels on weather (stand-alone, prepared list): hit / miss / requests to memory
for op in or and; do
  for grp in "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"; do
    tag=$(echo $grp | cut -d' ' -f1)
    rm -rf $O/pmc_${op}_$tag
    (cd /tmp && LIST=1 RHIP_NO_OVERLAP=1 timeout 200 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc_${op}_$tag -o p -- python $GRAFT_REPO_ROOT/scripts/prof_weather.py $op weather_sept_85 > $GRAFT_REPO_ROOT/$O/pmc_${op}_$tag.log 2>&1)
    echo "== weather $op: $grp"; python scripts/pmc_l2.py $O/pmc_${op}_$tag
    rm -f $O/pmc_${op}_$tag/*/*kernel_trace.csv
  done
done 2>&1 | tee $O/pmc_l2.txt
# host clock with every slot warm (prof_weather resets the clock after 2 calls: 12 calls rotate over 4 slots)
for spec in and:weather_sept_85 and:census1881; do
  op=${spec%%:*}; ds=${spec#*:}
  LIST=1 python scripts/prof_weather.py $op $ds 2>&1 | grep "min ms" | cut -c1-250 | tee -a $O/hostclk.txt
done
# the bench line (new layout) + its detail file
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; wc -c $O/bench.json; python scripts/bench_line.py < $O/bench.json | cut -c1-400
cp gpurun_out/bench_detail.json $O/ 2>/dev/null
grep -v BENCH_DETAIL $O/bench.err | tail -5
# more malloc'ed result arenas: how often does each k_bb mode come up
timeout 200 scripts/bin/arena_place 10 malloc > $O/place_malloc10.txt 2>&1; cat $O/place_malloc10.txt | tail -12
