cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_gpu_parity.py -q -x -k "long_interval or tiny_interval or synth_every or explicit_unit or class_stats" 2>&1 | tail -2
for i in 1 2; do MULTI=0 timeout 200 python scripts/quick_all.py 2>/dev/null; done
