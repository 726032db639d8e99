cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r6o
mkdir -p $O
RHIP_LIB_VARIANT=phases timeout 200 python scripts/phases_weather.py weather_sept_85 or xor 2>&1 | grep -v "^$" | tee $O/phases.txt
