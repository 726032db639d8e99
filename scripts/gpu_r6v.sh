cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r6v
mkdir -p $O
for sl in 1280 1920 2560 3840 5120; do
  for i in 1 2; do echo "slots $sl: $(RHIP_MANY_SLOTS=$sl timeout 120 python scripts/prof_c4.py 100000 2>/dev/null | tail -1 | cut -c1-120)"; done
done | tee $O/slots.txt
