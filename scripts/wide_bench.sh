timeout -k 10 300 python bench.py --config 12block --steps 6 --warmup 2 --no-cpu-baseline --no-extra "$@" > gpurun_out/w_bench.json 2> gpurun_out/w_bench.err
rc=$?; tail -c 400 gpurun_out/w_bench.err; python3 -c "
import json; r=json.load(open('gpurun_out/w_bench.json')); print(r['value'], r['ms_per_step'], r['roofline']['avg_launch_us'], r['roofline']['frac'], r['kernel_ms_per_inference_step'])"; exit $rc
