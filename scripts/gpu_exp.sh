# One parameterised launcher for the short GPU experiments of a round (round 5 made ~20 calls of 25-60 s each; rounds 2-4
# kept one two-line script per call -- those are gone, `git log -- scripts/` has them).
#   gpurun --timeout 900 -- 'bash scripts/gpu_exp.sh <what> [args]'
#     c4 [n_bitmaps]            timing + kernel timeline of one or_many call (scripts/prof_c4.py, scripts/trace_many.py)
#     c4-variants v1 v2 ...     the same for library variants built beside the product (RHIP_BUILD_VARIANT=v RHIP_EXTRA_FLAGS=-D...
#                               python -m croaring_amd.build on the CPU side first; "" = the product); e.g. the ablations of
#                               profiles/r05_many_l1_notes.md: l11 = -DRHIP_ABL_L1=1 ... sc1 = -DRHIP_ABL_SC=1
#     c4-env VAR v1 v2 ...      timing of prof_c4.py under VAR=v (RHIP_MANY_T, RHIP_MANY_PF, RHIP_MANY_SLOTS ...)
#     c4-pmc "SET" ...          one rocprofv3 --pmc run per counter set on prof_c4.py, many-way kernels printed
#     realdata [datasets]       scripts/quick_all.py over prepared pair lists
#     tests <pytest -k expr>    a slice of the -m gpu suite
# Every rocprofv3 line runs under `timeout`: a counter run serialises kernels.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/exp
mkdir -p $O
what=$1; shift
case $what in
  c4)
    n=${1:-100000}
    timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c4 -o p -- python scripts/prof_c4.py $n > $O/prof_c4.log 2>&1
    tail -1 $O/prof_c4.log | cut -c1-200; python scripts/trace_many.py $O/prof_c4 "or_many $n"; rm -f $(find $O/prof_c4 -name "*kernel_trace.csv") ;;
  c4-variants)
    for v in "$@"; do
      RHIP_LIB_VARIANT=$v timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$v -o p -- python scripts/prof_c4.py 100000 > $O/prof_$v.log 2>&1
      python scripts/trace_many.py $O/prof_$v "variant '${v:-product}'"; rm -f $(find $O/prof_$v -name "*kernel_trace.csv")
    done ;;
  c4-env)
    var=$1; shift
    for v in "$@"; do echo "== $var=$v"; env $var=$v timeout 200 python scripts/prof_c4.py ${N:-100000} 2>&1 | tail -1 | cut -c1-200; done ;;
  c4-pmc)
    for set in "$@"; do
      tag=$(echo $set | cut -d' ' -f1)
      timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/pmc_$tag -o w -- python scripts/prof_c4.py 100000 > $O/pmc_$tag.log 2>&1
      python - "$O/pmc_$tag" <<'P'
import csv, glob, sys, collections, os
fs = glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(fs[0])) if fs else []:
    agg[r["Kernel_Name"].split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in agg.items():
    if "k_many" in k: print(k, {n: f"{sum(v)/len(v):.5g}" for n, v in c.items()})
P
      rm -f $(find $O/pmc_$tag -name "*kernel_trace.csv")
    done ;;
  realdata)
    LIST=1 MULTI=0 timeout 200 python scripts/quick_all.py "$@" 2>/dev/null ;;
  tests)
    timeout 1500 python -m pytest tests -q -m gpu -k "$1" > $O/tests.txt 2>&1; grep -E "passed|failed|^E  " $O/tests.txt | tail -12 | cut -c1-300 ;;
  *) echo "unknown experiment $what" ;;
esac
