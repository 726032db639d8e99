cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r6m
mkdir -p $O
for v in desabl4 desabl5; do
  RHIP_LIB_VARIANT=$v timeout 120 rocprofv3 --kernel-trace --output-format csv -d $O/prof_ld_$v -o p -- python scripts/prof_loader.py 100000 > $O/prof_ld_$v.log 2>&1; grep "^loader" $O/prof_ld_$v.log | cut -c1-200
  echo "variant '$v' k_des_payload us:"; python - $O/prof_ld_$v <<'P'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_des_payload" in r["Kernel_Name"]:
            print("  ", (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
P
done
