cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/prof_r2
TAG=multi timeout 300 python scripts/quick_all.py 2>&1 | grep -v amdgpu.ids
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --tb=short -k "multi or realdata_all_pairs or synth_every or batches_in_flight or class_stats" 2>&1 | tail -2
