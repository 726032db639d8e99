cd $GRAFT_REPO_ROOT
O=gpurun_out/r4n
mkdir -p $O
SLAB_GIB=128 timeout 200 scripts/bin/arena_place 2 slab:malloc > $O/slab128_1g.txt 2>&1
SLAB_GIB=128 SLAB_FROM_GIB=18 SLAB_TO_GIB=22 SLAB_STEP_MIB=64 timeout 200 scripts/bin/arena_place 1 slab:malloc > $O/slab128_fine.txt 2>&1
SLAB_GIB=64 timeout 200 scripts/bin/arena_place 2 slab:contig > $O/slab64_contig_1g.txt 2>&1
python - <<'P'
import re
for f in ("slab128_1g", "slab128_fine", "slab64_contig_1g"):
    for line in open(f"gpurun_out/r4n/{f}.txt"):
        if not line.startswith("slab"): continue
        pts = [(float(a), float(b)) for a, b in re.findall(r"([\d.]+)G:([\d.]+)", line)]
        print(f, line.split(":")[0], "fast(<4.15):", [a for a, b in pts if b < 4.15], "slow(>4.5):", [a for a, b in pts if b > 4.5][:40], "n", len(pts))
P
