cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r5m
mkdir -p $O
timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c4 -o p -- python scripts/prof_c4.py 100000 > $O/prof_c4.log 2>&1
python scripts/trace_many.py $O/prof_c4 "c4 or_many 100000" | tee -a $O/timeline_c4.txt
rm -f $(find $O/prof_c4 -name "*kernel_trace.csv")
for pf in 4; do echo "== PF $pf"; RHIP_MANY_PF=$pf timeout 90 python scripts/prof_c4.py 100000 2>&1 | tail -1 | cut -c1-200; done
for set in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $set | cut -d' ' -f1)
  timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/pmc_$tag -o w -- python scripts/prof_c4.py 100000 > $O/pmc_$tag.log 2>&1
  python - "$O/pmc_$tag" <<'P'
import csv, glob, sys, collections, os
fs = glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True)
if not fs: print("no csv", sys.argv[1]); sys.exit()
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(fs[0])):
    agg[r["Kernel_Name"].split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in agg.items():
    if "k_many_l1" not in k: continue
    print(k, {n: f"{sum(v)/len(v):.5g}" for n, v in c.items()})
P
  rm -f $(find $O/pmc_$tag -name "*kernel_trace.csv")
done
timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -x -q -m gpu -k "many or c4 or full_container or sharded or dense" > $O/tests.txt 2>&1; grep -E "passed|failed|error" $O/tests.txt | tail -3
