cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r6m
mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q --tb=short -x -k "deser or blob or frozen or robust or c4 or portable or load or layout or 64bit" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
timeout 120 rocprofv3 --kernel-trace --output-format csv -d $O/prof_ld -o p -- python scripts/prof_loader.py 100000 > $O/prof_ld.log 2>&1; grep "^loader" $O/prof_ld.log | cut -c1-200
python - $O/prof_ld <<'P'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_des_payload" in r["Kernel_Name"]:
            print("  k_des_payload us", (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
P
timeout 300 python scripts/bench_poolops.py 2>/dev/null | grep '"load"\|frozen' | cut -c1-260
