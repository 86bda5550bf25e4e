# cfg 3 (12block): deferred update against CORA_NO_LAZY=1, alternating
cd $GRAFT_REPO_ROOT
for rep in 1 2; do for mode in lazy nolazy; do
  if [ $mode = nolazy ]; then export CORA_NO_LAZY=1; else unset CORA_NO_LAZY; fi
  line=$(timeout -k 10 200 python3 bench.py --config 12block --no-extra --no-cpu-baseline --steps 10 --warmup 3) || exit 1
  echo "cfg3 $mode $(echo "$line" | python3 -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"])')"
done; done
