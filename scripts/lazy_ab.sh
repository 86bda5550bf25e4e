# A/B of the deferred vode update (default) against CORA_NO_LAZY=1, alternating runs in one session
# usage: scripts/lazy_ab.sh [batch]  ->  gpurun_out/lazy_ab_B<batch>.txt
cd $GRAFT_REPO_ROOT
B=${1:-200}
out=gpurun_out/lazy_ab_B$B.txt; : > $out
for rep in 1 2 3; do
  for mode in lazy nolazy; do
    if [ $mode = nolazy ]; then export CORA_NO_LAZY=1; else unset CORA_NO_LAZY; fi
    line=$(timeout -k 10 120 python3 bench.py --batch $B --no-extra --no-cpu-baseline --steps 50 --warmup 10) || exit 1
    echo "B=$B $mode $(echo "$line" | python3 -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"])')" | tee -a $out
  done
done
