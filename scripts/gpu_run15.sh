cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
RHIP_COMPAT_STATS=1 timeout 900 ./oracle/_ref/toplevel_unit_dropin > gpurun_out/dropin.out 2> gpurun_out/dropin.err; echo rc=$?; cat gpurun_out/dropin.out; grep -E "FAILED|ERROR|compat" gpurun_out/dropin.err | head -40
