# 8-wave (CORA_PAIR=1) against 4-wave (CORA_PAIR=0) form of the D = 64 inference kernel over the batch size
# usage: scripts/pair_sweep.sh [batches...]  ->  gpurun_out/pair_sweep.txt
cd $GRAFT_REPO_ROOT
out=gpurun_out/pair_sweep.txt; : > $out
for B in ${@:-25 42 50 64 85 100 128 200}; do
  for P in 0 1; do
    line=$(CORA_PAIR=$P timeout -k 10 120 python3 bench.py --batch $B --no-extra --no-cpu-baseline --steps 30 --warmup 5) || exit 1
    echo "B=$B pair=$P $(echo "$line" | python3 -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"], d["roofline"]["achieved"])')" | tee -a $out
  done
done
