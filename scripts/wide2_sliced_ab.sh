# A/B of the two-workgroup kernel (pc_block_wide2, sliced heads) against the one-workgroup kernel (CORA_WIDE1=1) on the
# grid points the former took over in round 3; results in gpurun_out/w2ab_D<D>_H<H>_w<0|1>.json
for DH in "128 1" "128 2" "96 1" "96 2" "96 4" "128 4"; do
  set -- $DH
  for W in 0 1; do
    if [ $W = 1 ]; then export CORA_WIDE1=1; else unset CORA_WIDE1; fi
    timeout -k 10 120 python bench.py --config 12block --hidden-size $1 --num-heads $2 --no-cpu-baseline --no-extra --steps 5 --warmup 2 > gpurun_out/w2ab_D$1_H$2_w$W.json 2> gpurun_out/w2ab_D$1_H$2_w$W.err || exit 1
  done
done
