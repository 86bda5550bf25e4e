# the 12-block sweep grid (sweeps/sweep_12block_comprehensive.yaml: hidden 64 / 96 / 128, heads 1 / 2 / 4) at B=200, T=20
mkdir -p gpurun_out
for D in 64 96 128; do for H in 1 2 4; do
  timeout -k 10 300 python bench.py --config 12block --hidden-size $D --num-heads $H --steps 6 --warmup 2 --no-cpu-baseline --no-extra > gpurun_out/grid_${D}_${H}.json 2>/dev/null || exit 1
  python3 -c "
import json; r=json.load(open('gpurun_out/grid_${D}_${H}.json')); print('D=$D H=$H', round(r['value']), 'img-steps/s', round(r['ms_per_step'],2), 'ms/step  whole-step frac', round(r['whole_step_mfma_frac'],3), ' dominant', r['roofline']['kernel'], round(r['roofline']['frac'],3))"
done; done
