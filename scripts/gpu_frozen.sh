# Frozen format + payload layout on the GPU: the two new test bodies, then the (de)serialization timings.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/frozen
mkdir -p $O
timeout 600 python -m pytest tests -q -m gpu -k "frozen or payload_layout or deserialization or bulk_serialization" > $O/tests.txt 2>&1; grep -E "passed|failed|^E  " $O/tests.txt | tail -12 | cut -c1-300
timeout 300 python scripts/bench_poolops.py > $O/poolops.jsonl 2> $O/poolops.err; grep -E '"frozen"|"load"|"download"' $O/poolops.jsonl | cut -c1-400; tail -3 $O/poolops.err
for pf in 2 4; do echo "== RHIP_MANY_PF=$pf"; RHIP_MANY_PF=$pf timeout 200 python scripts/prof_c4.py 100000 2>&1 | tail -1 | cut -c1-200; done
