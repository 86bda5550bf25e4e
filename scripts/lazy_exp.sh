# timing experiments on the deferred vode update (scripts/bin/libcora_exp<k>.so built with -DEXP_LAZY=k; wrong numerics)
cd $GRAFT_REPO_ROOT
out=gpurun_out/lazy_exp.txt; : > $out
run() {
  line=$(env "$@" timeout -k 10 120 python3 bench.py --batch 200 --no-extra --no-cpu-baseline --steps 30 --warmup 5) || exit 1
  echo "$* $(echo "$line" | python3 -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"], d["kernel_ms_per_inference_step"])')" | tee -a $out
}
run CORA_NO_LAZY=1
run CORA_X=0
for k in 1 2 3; do run CORA_HIP_LIB=$GRAFT_REPO_ROOT/scripts/bin/libcora_exp$k.so; done
