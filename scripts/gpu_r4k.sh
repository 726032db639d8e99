cd $GRAFT_REPO_ROOT
O=gpurun_out/r4k
mkdir -p $O
G=$((1<<30))
P0=$((8*G)); PB=$((8*G + G + G/8 + 256)); AX=$((250*4096*8192 + 4096)); AB=$((AX + AX/8 + 256)); A8=$((8*G)); A16=$((16*G))
for cfg in "A $P0 $AX" "B $PB $AX" "C $P0 $AB" "D $P0 $A8" "E $P0 $A16" "F $PB $AB"; do
  set -- $cfg
  echo "== $1: POOL_BYTES=$2 ARENA_BYTES=$3"
  POOL_BYTES=$2 ARENA_BYTES=$3 timeout 100 scripts/bin/arena_place 8 malloc 2>&1 | grep -E "pool|trial" | awk '{print $1, $2, $3, $5, $6, $(NF-4), $(NF-3)}' | tr '\n' ';'
  echo
done 2>&1 | tee $O/sizes.txt
