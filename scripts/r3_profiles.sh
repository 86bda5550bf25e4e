# Round-3 profile set (run on the GPU box; copy gpurun_out/r3_* into profiles/ afterwards):
# kernel-trace stats and HBM-traffic PMC passes for every benchmarked configuration (cfg 2 fused D=64 kernel,
# cfg 3 wide kernel, cfg 4 per-op fp32 / bf16, cfg 5 row-tile kernels fp32 / bf16), SQ counters, the parity
# table, the GEMM timing experiments, the default bench line.
set -e
mkdir -p gpurun_out
bash scripts/prof_stats.sh r3cfg2 --no-extra > gpurun_out/r3_stats_cfg2.txt
cp gpurun_out/stats_r3cfg2.csv gpurun_out/r3_fused_path_kernel_stats.csv
bash scripts/prof_stats.sh r3w12 --config 12block --steps 8 --warmup 3 --no-extra > gpurun_out/r3_stats_12block.txt
cp gpurun_out/stats_r3w12.csv gpurun_out/r3_wide_path_kernel_stats.csv
bash scripts/pmc_traffic.sh cfg2 > gpurun_out/r3_traffic_cfg2.txt
cp gpurun_out/pmc_traffic_cfg2.json gpurun_out/r3_pmc_traffic.json
bash scripts/pmc_traffic.sh 12block --config 12block > gpurun_out/r3_traffic_12block.txt
cp gpurun_out/pmc_traffic_12block.json gpurun_out/r3_pmc_traffic_12block.json
{ echo "# cfg 2 (6block, B=200)"; bash scripts/pmc_sq.sh cfg2; echo "# 12block (D=128, H=4, L=12, B=200)"; bash scripts/pmc_sq.sh 12block --config 12block;
  echo "# video16 fp32 (row-tile kernels)"; bash scripts/pmc_sq.sh video16 --config video16;
  echo "# imagenet224 bf16 (per-op)"; bash scripts/pmc_sq.sh i224bf16 --config imagenet224 --dtype bf16; } > gpurun_out/r3_sq_counters.txt
# cfg 4 (ImageNet preset, B=64) and cfg 5 (video16, B=16): kernel stats, FETCH_SIZE / WRITE_SIZE passes, bench lines
for cfg in imagenet224 video16; do
  for dt in fp32 bf16; do
    bash scripts/prof_stats.sh r3${cfg}$dt --config $cfg --dtype $dt --steps 2 --warmup 1 --no-extra > gpurun_out/r3_stats_${cfg}_$dt.txt
    cp gpurun_out/stats_r3${cfg}$dt.csv gpurun_out/r3_${cfg}_${dt}_kernel_stats.csv
    bash scripts/pmc_traffic.sh ${cfg}_$dt --config $cfg --dtype $dt > gpurun_out/r3_traffic_${cfg}_$dt.txt
    cp gpurun_out/pmc_traffic_${cfg}_$dt.json gpurun_out/r3_pmc_traffic_${cfg}_$dt.json
    timeout -k 10 300 python3 bench.py --config $cfg --steps 5 --warmup 2 --no-cpu-baseline --no-extra --dtype $dt > gpurun_out/r3_${cfg}_${dt}_bench_line.json 2>/dev/null
  done
done
# (the GEMM timing experiments need scripts/bin/gb_* built from scripts/gemm_bf16_bench.hip; skipped when they are absent)
if [ -x scripts/bin/gb_x ]; then
{ for b in gb_x gb_xDEXP_NOCVT gb_xDEXP_NOCVTDEXP_NOGELUGRAD; do echo "== scripts/gemm_bf16_bench.hip built with: ${b#gb_x}"; timeout -k 5 60 scripts/bin/$b pre; done; } > gpurun_out/r3_gemm_bf16_experiments.txt
fi
# the sweep-grid points that moved to the two-workgroup kernel in round 3: kernel stats, traffic and SQ counters of two of them
bash scripts/prof_stats.sh r3d128h1 --config 12block --hidden-size 128 --num-heads 1 --steps 8 --warmup 3 --no-extra > gpurun_out/r3_stats_12block_D128_H1.txt
bash scripts/prof_stats.sh r3d96h4 --config 12block --hidden-size 96 --num-heads 4 --steps 8 --warmup 3 --no-extra > gpurun_out/r3_stats_12block_D96_H4.txt
bash scripts/pmc_traffic.sh 12block_D128_H1 --config 12block --hidden-size 128 --num-heads 1 > gpurun_out/r3_traffic_12block_D128_H1.txt
cp gpurun_out/pmc_traffic_12block_D128_H1.json gpurun_out/r3_pmc_traffic_12block_D128_H1.json
{ echo "# 12block D=128 H=1 (sliced heads)"; bash scripts/pmc_sq.sh d128h1 --config 12block --hidden-size 128 --num-heads 1;
  echo "# 12block D=96 H=4 (heads of 24)"; bash scripts/pmc_sq.sh d96h4 --config 12block --hidden-size 96 --num-heads 4; } > gpurun_out/r3_sq_counters_grid.txt
bash scripts/wide2_sliced_ab.sh
python3 - <<'PY' > gpurun_out/r3_wide2_ab.txt
import json
print("two-workgroup kernel (pc_block_wide2) against the one-workgroup kernel (CORA_WIDE1=1), 12block shape, B=200, T=20")
for D, H in ((128, 1), (128, 2), (128, 4), (96, 1), (96, 2), (96, 4)):
    for W in (0, 1):
        d = json.loads(open(f"gpurun_out/w2ab_D{D}_H{H}_w{W}.json").read().strip().splitlines()[-1])
        print(f"D={D} H={H} {'wide  (1 wg/CU)' if W else 'wide2 (2 wg/CU)'} {d['value']:9.0f} images*T-steps/s  whole-step {d['whole_step_mfma_frac']:.3f}  "
              f"launch {d['roofline']['avg_launch_us']:7.1f} us  {d['roofline']['frac']:.3f} of the fp32 matrix peak")
PY
timeout -k 10 900 python3 -u scripts/parity_table.py gpurun_out/r3_parity_table.json > gpurun_out/r3_parity_table.log 2>&1
timeout -k 10 400 python3 bench.py > gpurun_out/r3_bench_line.json 2> gpurun_out/r3_bench.err
tail -c 1500 gpurun_out/r3_bench_line.json
