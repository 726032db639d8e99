cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
RHIP_COMPAT_STATS=1 timeout 900 ./oracle/_ref/cpp_unit_dropin > gpurun_out/dropin_cppunit.out 2> gpurun_out/dropin_cppunit.err; echo rc=$?; grep -E "tests, " gpurun_out/dropin_cppunit.out | cut -c1-200; grep -E "FAILED|ERROR|compat" gpurun_out/dropin_cppunit.err | head -10 | cut -c1-300
