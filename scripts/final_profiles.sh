# Round profile set for profiles/: kernel-trace stats, HBM traffic PMC passes, SQ MFMA counters, bench line
set -e
bash scripts/prof_stats.sh final > gpurun_out/final_stats.txt
bash scripts/pmc_traffic.sh > gpurun_out/final_traffic.txt
bash scripts/pmc_sq.sh > gpurun_out/final_sq.txt
timeout -k 10 400 python bench.py > gpurun_out/final_bench.json 2> gpurun_out/final_bench.err
tail -c 1500 gpurun_out/final_bench.json
