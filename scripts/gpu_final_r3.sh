# round-3 measurement pass: everything profiles/r03_* cites, from ONE box.  gpurun -- 'bash scripts/gpu_final_r3.sh'
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/final_r3
mkdir -p $O
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; cut -c1-200 $O/bench.json
for i in 1 2 3; do timeout 200 python scripts/bench_c2_ops.py 2>/dev/null; done > $O/c2_ops.jsonl
timeout 200 python scripts/quick_c3.py > $O/quick_c3.jsonl 2>/dev/null
timeout 200 python scripts/quick_all.py > $O/multi.txt 2>/dev/null
timeout 200 python scripts/quick_classes.py > $O/class_throughput.jsonl 2>/dev/null
timeout 200 python scripts/bench_poolops.py > $O/poolops.jsonl 2>/dev/null
# kernel-trace stats: the bench command itself (C2 only), then one realdata batch type per run
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -o bench -- python bench.py --steps 6 --warmup 2 --no-cpu --no-secondary > $O/prof_bench.log 2>&1
for spec in w_and:and:weather_sept_85 w_or:or:weather_sept_85 w_xor:xor:weather_sept_85 w_andnot:andnot:weather_sept_85 c1_and:and:census1881 c1_or:or:census1881 wk_and:and:wikileaks-noquotes c5_and:and:c5 c5_or:or:c5; do
  name=${spec%%:*}; rest=${spec#*:}; op=${rest%%:*}; ds=${rest#*:}
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$name -o p -- python scripts/prof_weather.py $op $ds > $O/prof_$name.log 2>&1
  grep "min ms" $O/prof_$name.log | cut -c1-120
done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c4 -o p -- python scripts/prof_c4.py 100000 > $O/prof_c4.log 2>&1; tail -1 $O/prof_c4.log | cut -c1-200
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_multi_c1 -o p -- python scripts/prof_multi.py census1881 > $O/prof_multi_c1.log 2>&1; tail -1 $O/prof_multi_c1.log | cut -c1-200
# per-kernel algorithmic GB/s on weather: class statistics + stand-alone durations (one stream)
RHIP_NO_OVERLAP=1 rocprofv3 --kernel-trace --output-format csv -d $O/pk_c3 -o p -- python scripts/per_kernel_c3.py weather_sept_85 > $O/pk_c3.out 2>/dev/null
python scripts/join_per_kernel.py $O/pk_c3.out $O/pk_c3 > $O/per_kernel_c3.jsonl; head -3 $O/per_kernel_c3.jsonl
# PMC passes, each counter set in its own run, kernel-trace only (no other trace domains)
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o b -- python bench.py --steps 2 --warmup 1 --rounds 1 --no-cpu --no-secondary > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o b -- python bench.py --steps 2 --warmup 1 --rounds 1 --no-cpu --no-secondary > $O/pmc_write.log 2>&1
for op in and or; do
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $O/pmc_w_$op -o w -- python scripts/prof_weather.py $op > $O/pmc_w_$op.log 2>&1
done
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $O/pmc_c4 -o w -- python scripts/prof_c4.py 100000 > $O/pmc_c4.log 2>&1
python - <<'P' > $O/pmc_c4_sq.md
import collections, csv, glob
fs = glob.glob('gpurun_out/final_r3/pmc_c4/*counter_collection.csv')
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(fs[0])) if fs else []:
    agg[r["Kernel_Name"].split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("# SQ counters of the many-way kernels, C4 or_many over 100 000 sparse bitmaps (r03)\n")
print("`rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace` on `scripts/prof_c4.py 100000` (6 calls)\n")
print("| kernel | launches | active | wait_any | wait_inst | VALU instr / launch | LDS instr / launch | LDS bank-conflict cycles / LDS-active |\n|---|---|---|---|---|---|---|---|")
for k, c in sorted(agg.items(), key=lambda kv: -sum(kv[1].get("SQ_WAVE_CYCLES", [0]))):
    if not k.startswith("k_many") or "SQ_WAVE_CYCLES" not in c: continue
    n = len(c["SQ_WAVE_CYCLES"]); wc = sum(c["SQ_WAVE_CYCLES"]); f = lambda x: sum(c.get(x, [0]))
    print("| `%s` | %d | %.0f %% | %.0f %% | %.0f %% | %.3g | %.3g | %.0f %% |" % (k, n, 100 * f('SQ_ACTIVE_INST_ANY') / wc, 100 * f('SQ_WAIT_ANY') / wc,
          100 * f('SQ_WAIT_INST_ANY') / wc, f('SQ_INSTS_VALU') / n, f('SQ_INSTS_LDS') / n, 100 * f('SQ_LDS_BANK_CONFLICT') / max(1, f('SQ_LDS_IDX_ACTIVE'))))
P
# raw traces are large: keep the stats and counter tables, drop the per-dispatch traces except the realdata ones (small)
rm -f $O/prof_bench/*kernel_trace.csv $O/prof_c4/*kernel_trace.csv $O/pk_c3/*kernel_trace.csv $O/pmc_c4/*kernel_trace.csv
du -sh $O
echo done
