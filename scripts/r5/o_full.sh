# round 5, call o: the whole -m gpu suite, then the driver's bench command
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r5o
mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu > $O/tests.txt 2>&1; grep -E "passed|failed|error" $O/tests.txt | tail -5; grep -E "^FAILED|^ERROR|Error" $O/tests.txt | head -10 | cut -c1-300
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; python scripts/bench_line.py $O/bench.json 2>&1 | cut -c1-400
