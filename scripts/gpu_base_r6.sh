# round-6 baseline / checkpoint: the -m gpu suite, smoke, the corpus timings and the driver's bench line from ONE box
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/base_r6${TAG:+_$TAG}
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q --tb=short > $O/pytest.log 2>&1; tail -5 $O/pytest.log | head -4
timeout 60 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | cut -c1-200
TAG="prepared" LIST=1 MULTI=1 timeout 200 python scripts/quick_all.py > $O/quick_all.txt 2>/dev/null; cat $O/quick_all.txt | cut -c1-250
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; python scripts/bench_line.py $O/bench.json | cut -c1-1500
cp gpurun_out/bench_detail.json $O/bench_detail.json 2>/dev/null
echo done
