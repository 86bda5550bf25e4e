# Round-2 profile set (run on the GPU box; copy gpurun_out/r2_* into profiles/ afterwards):
# kernel-trace stats of the bench workload (cfg 2, fused D=64 kernel) and of the 12-block config (wide
# kernel), HBM traffic PMC passes, SQ counters, the parity table, the default bench line.
set -e
mkdir -p gpurun_out
bash scripts/prof_stats.sh r2cfg2 --no-extra > gpurun_out/r2_stats_cfg2.txt
cp gpurun_out/stats_r2cfg2.csv gpurun_out/r2_fused_path_kernel_stats.csv
bash scripts/prof_stats.sh r2w12 --config 12block --no-extra > gpurun_out/r2_stats_12block.txt
cp gpurun_out/stats_r2w12.csv gpurun_out/r2_wide_path_kernel_stats.csv
bash scripts/pmc_traffic.sh cfg2 > gpurun_out/r2_traffic_cfg2.txt
cp gpurun_out/pmc_traffic_cfg2.json gpurun_out/r2_pmc_traffic.json
bash scripts/pmc_traffic.sh 12block --config 12block > gpurun_out/r2_traffic_12block.txt
cp gpurun_out/pmc_traffic_12block.json gpurun_out/r2_pmc_traffic_12block.json
{ echo "# cfg 2 (6block, B=200)"; bash scripts/pmc_sq.sh cfg2; echo "# 12block (D=128, H=4, L=12, B=200)"; bash scripts/pmc_sq.sh 12block --config 12block; } > gpurun_out/r2_sq_counters.txt
# cfg 4's shape at B=64 on the per-op path, fp32 against the bf16 mode (kernel stats + bench lines), video16 lines
timeout -k 10 300 python3 bench.py --config imagenet224 --steps 3 --warmup 1 --no-cpu-baseline --no-extra --dtype bf16x3 > gpurun_out/r2_imagenet224_bf16x3_bench_line.json 2>/dev/null
timeout -k 10 60 scripts/bin/mfma_rate > gpurun_out/r2_mfma_rate.txt
for dt in fp32 bf16; do
  bash scripts/prof_stats.sh r2i224$dt --config imagenet224 --dtype $dt --steps 2 --warmup 1 --no-extra > gpurun_out/r2_stats_i224_$dt.txt
  cp gpurun_out/stats_r2i224$dt.csv gpurun_out/r2_imagenet224_${dt}_kernel_stats.csv
  timeout -k 10 300 python3 bench.py --config imagenet224 --steps 3 --warmup 1 --no-cpu-baseline --no-extra --dtype $dt > gpurun_out/r2_imagenet224_${dt}_bench_line.json 2>/dev/null
  timeout -k 10 300 python3 bench.py --config video16 --steps 3 --warmup 1 --no-cpu-baseline --no-extra --dtype $dt > gpurun_out/r2_video16_${dt}_bench_line.json 2>/dev/null
done
timeout -k 10 600 python3 -u scripts/parity_table.py gpurun_out/r2_parity_table.json > gpurun_out/r2_parity_table.log 2>&1
timeout -k 10 400 python3 bench.py > gpurun_out/r2_bench_line.json 2> gpurun_out/r2_bench.err
tail -c 1800 gpurun_out/r2_bench_line.json
