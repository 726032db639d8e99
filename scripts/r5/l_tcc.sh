# round 5, call l: memory-side counters of the many-way kernels on C4 (separate passes), PF 2 default timeline
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r5l
mkdir -p $O
timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c4 -o p -- python scripts/prof_c4.py 100000 > $O/prof_c4.log 2>&1
python scripts/trace_many.py $O/prof_c4 "c4 or_many 100000" | tee -a $O/timeline_c4.txt
rm -f $(find $O/prof_c4 -name "*kernel_trace.csv")
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_REQ_sum TCC_READ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
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
    if "k_many_l1" not in k and "k_many_scatter" not in k: continue
    print(k, {n: f"{sum(v)/len(v):.5g}" for n, v in c.items()})
P
  rm -f $(find $O/pmc_$tag -name "*kernel_trace.csv")
done
