set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -c "
import croaring_amd, numpy as np
e = croaring_amd.Engine(); print('engine ok (no torch)')
p = e.pool_synth_bitset(4, 8, 123); print(p.type_counts(), p.cardinalities())
" > gpurun_out/first.log 2>&1
python -c "
import torch; print(torch.cuda.is_available(), torch.cuda.get_device_name(0))
import croaring_amd
e = croaring_amd.Engine(); print('engine ok (torch first)')
p = e.pool_synth_bitset(4, 8, 123); print(p.type_counts(), p.cardinalities())
r = e.pairwise('and', p, [0,1],p,[1,2]); print(r.cardinalities())
torch.cuda.synchronize()
" >> gpurun_out/first.log 2>&1
timeout 900 python -m pytest tests -m gpu -q --tb=short -x > gpurun_out/pytest1.log 2>&1
tail -30 gpurun_out/pytest1.log
cat gpurun_out/first.log
