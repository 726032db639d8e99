cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
RHIP_COMPAT_STATS=1 timeout 900 ./oracle/_ref/cpp_random_unit_dropin > gpurun_out/dropin_cpp.out 2> gpurun_out/dropin_cpp.err; echo rc=$?; tail -2 gpurun_out/dropin_cpp.out | cut -c1-300; grep -E "FAILED|ERROR|compat" gpurun_out/dropin_cpp.err | head -10 | cut -c1-300
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_64bit.py -m gpu -q --tb=short -x > gpurun_out/pytest17.log 2>&1; grep -E "passed|failed|Error" gpurun_out/pytest17.log | tail -3
