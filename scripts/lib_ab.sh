# A/B of experimental library builds against the shipped one: scripts/lib_ab.sh <batch> <lib.so>...
cd $GRAFT_REPO_ROOT
B=$1; shift
for rep in 1 2 3; do
  for lib in shipped "$@"; do
    if [ "$lib" = shipped ]; then unset CORA_HIP_LIB; else export CORA_HIP_LIB=$GRAFT_REPO_ROOT/$lib; fi
    line=$(timeout -k 10 120 python3 bench.py --batch $B --no-extra --no-cpu-baseline --steps 50 --warmup 10) || exit 1
    echo "B=$B $lib $(echo "$line" | python3 -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"])')"
  done
done
