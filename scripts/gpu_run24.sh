cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_compat.py tests/test_gpu_dropin_harness.py -m gpu -q --tb=short -x > gpurun_out/pytest12.log 2>&1; grep -E "passed|failed|Error|assert" gpurun_out/pytest12.log | tail -8
RHIP_COMPAT_STATS=1 ./oracle/_ref/toplevel_unit_dropin > gpurun_out/dropin.out 2> gpurun_out/dropin.err; tail -1 gpurun_out/dropin.out; grep -E "FAILED|compat" gpurun_out/dropin.err | head
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -x -k "randomized" > gpurun_out/pytest13.log 2>&1; grep -E "passed|failed|Error" gpurun_out/pytest13.log | tail -3
