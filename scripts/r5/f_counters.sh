cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r5f
mkdir -p $O
timeout 60 rocprofv3 -L > $O/avail.txt 2>&1
grep -o "SQ_[A-Z0-9_]*\|TCP_[A-Z0-9_]*\|TCC_[A-Z0-9_]*\|LDS[A-Za-z0-9_]*\|GRBM_[A-Z0-9_]*" $O/avail.txt | sort -u > $O/avail_names.txt
wc -l $O/avail_names.txt
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_INST_CYCLES_VMEM" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_ATOMIC_RETURN SQ_LDS_MEM_VIOLATIONS SQ_WAIT_ANY SQ_WAIT_INST_ANY"; do
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
    if "k_many" not in k: continue
    print(k, {n: f"{sum(v)/len(v):.4g}" for n, v in c.items()})
P
  rm -f $(find $O/pmc_$tag -name "*kernel_trace.csv")
done
