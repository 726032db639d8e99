# round-6 measurement pass: everything profiles/r06_* cites, from ONE box.  gpurun --timeout 1500 -- 'bash scripts/gpu_final_r6.sh'
# (every rocprofv3 line under `timeout`: a counter run serialises kernels)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/final_r6
mkdir -p $O
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; wc -c $O/bench.json; python scripts/bench_line.py $O/bench.json | cut -c1-300
cp gpurun_out/bench_detail.json $O/bench_detail.json
timeout 200 python scripts/bench_c2_ops.py 2>/dev/null > $O/c2_ops.jsonl
TAG="prepared" LIST=1 MULTI=1 timeout 200 python scripts/quick_all.py > $O/quick_all.txt 2>/dev/null
TAG="adhoc" LIST=0 MULTI=0 timeout 200 python scripts/quick_all.py >> $O/quick_all.txt 2>/dev/null
timeout 200 python scripts/quick_classes.py > $O/class_throughput.jsonl 2>/dev/null
timeout 300 python scripts/bench_poolops.py > $O/poolops.jsonl 2>/dev/null
# kernel-trace stats + timelines
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -o bench -- python bench.py --steps 6 --warmup 2 --no-cpu --no-secondary > $O/prof_bench.log 2>&1
for spec in w_and:and:weather_sept_85 w_or:or:weather_sept_85 w_xor:xor:weather_sept_85 w_andnot:andnot:weather_sept_85 c1_and:and:census1881 c1_or:or:census1881 c5_and:and:c5 c5_or:or:c5; do
  name=${spec%%:*}; rest=${spec#*:}; op=${rest%%:*}; ds=${rest#*:}
  LIST=1 timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$name -o p -- python scripts/prof_weather.py $op $ds > $O/prof_$name.log 2>&1
  grep "min ms" $O/prof_$name.log | cut -c1-120
done
timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c4 -o p -- python scripts/prof_c4.py 100000 > $O/prof_c4.log 2>&1; tail -1 $O/prof_c4.log | cut -c1-200
python scripts/trace_many.py $O/prof_c4 "c4 or_many, 100 000 sparse bitmaps" > $O/timelines_many.txt
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c4x10 -o p -- python scripts/prof_c4.py 1000000 > $O/prof_c4x10.log 2>&1; tail -1 $O/prof_c4x10.log | cut -c1-200
python scripts/trace_many.py $O/prof_c4x10 "c4 x 10 or_many, 10^6 sparse bitmaps" >> $O/timelines_many.txt
timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_loader -o p -- python scripts/prof_loader.py 100000 > $O/prof_loader.log 2>&1; grep "^loader" $O/prof_loader.log | cut -c1-200
# six fresh processes: the headline with the address placement, then two with the candidate search of round 4
for i in 1 2 3 4 5 6; do timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu --no-secondary --no-live-traffic 2>/dev/null | python scripts/bench_line.py | head -2 | tr "\n" " "; echo; done > $O/placement_runs.txt
for i in 1 2; do RHIP_ARENA_VMM=0 timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu --no-secondary --no-live-traffic 2>/dev/null | python scripts/bench_line.py | head -2 | tr "\n" " "; echo "(RHIP_ARENA_VMM=0)"; done >> $O/placement_runs.txt
cat $O/placement_runs.txt | cut -c1-400
cat $O/timelines_many.txt
# per-kernel algorithmic GB/s on weather: class statistics + stand-alone durations (one stream)
RHIP_NO_OVERLAP=1 timeout 200 rocprofv3 --kernel-trace --output-format csv -d $O/pk_c3 -o p -- python scripts/per_kernel_c3.py weather_sept_85 > $O/pk_c3.out 2>/dev/null
python scripts/join_per_kernel.py $O/pk_c3.out $O/pk_c3 > $O/per_kernel_c3.jsonl; head -2 $O/per_kernel_c3.jsonl | cut -c1-200
# PMC passes, each counter set in its own run, kernel-trace only
timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o b -- python bench.py --steps 2 --warmup 1 --rounds 1 --no-cpu --no-secondary > $O/pmc_fetch.log 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o b -- python bench.py --steps 2 --warmup 1 --rounds 1 --no-cpu --no-secondary > $O/pmc_write.log 2>&1
SQSET="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
for op in and or; do
  LIST=1 timeout 200 rocprofv3 --pmc $SQSET --kernel-trace --output-format csv -d $O/pmc_w_$op -o w -- python scripts/prof_weather.py $op > $O/pmc_w_$op.log 2>&1
done
timeout 200 rocprofv3 --pmc $SQSET --kernel-trace --output-format csv -d $O/pmc_c4 -o w -- python scripts/prof_c4.py 100000 > $O/pmc_c4.log 2>&1
python scripts/summarize_sq_c4.py $O/pmc_c4 > $O/pmc_c4_sq.md
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"; do
  tag=$(echo $set | cut -d' ' -f1)
  timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/pmc_many_$tag -o w -- python scripts/prof_c4.py 100000 > $O/pmc_many_$tag.log 2>&1
done
# realdata: HBM traffic and issue shares per batch (bench.py's hbm_traffic_frac column)
RTSQ="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
for spec in c3_and:and:weather_sept_85 c3_or:or:weather_sept_85 c3_xor:xor:weather_sept_85 c3_andnot:andnot:weather_sept_85 c1_and:and:census1881 c1_or:or:census1881 c5_and:and:c5 c5_or:or:c5; do
  name=${spec%%:*}; rest=${spec#*:}; op=${rest%%:*}; ds=${rest#*:}
  LIST=1 timeout 120 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/rt_${name}_fetch -o w -- python scripts/prof_weather.py $op $ds > /dev/null 2>&1
  LIST=1 timeout 120 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/rt_${name}_write -o w -- python scripts/prof_weather.py $op $ds > /dev/null 2>&1
  LIST=1 timeout 120 rocprofv3 --pmc $RTSQ --kernel-trace --output-format csv -d $O/rt_${name}_sq -o w -- python scripts/prof_weather.py $op $ds > /dev/null 2>&1
done
rm -f $(find $O/prof_bench $O/pk_c3 $O/pmc_* $O/rt_* $O/prof_c4x10 -name "*kernel_trace.csv")
du -sh $O
echo done
