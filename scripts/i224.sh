for dt in fp32 bf16; do
timeout -k 10 300 python bench.py --config imagenet224 --steps 3 --warmup 1 --no-cpu-baseline --no-extra --dtype $dt > gpurun_out/i224_$dt.json 2>/dev/null || exit 1
python3 -c "
import json; r=json.load(open('gpurun_out/i224_$dt.json')); print('$dt', r['value'], r['ms_per_step'], r['whole_step_mfma_frac']); print(r['kernel_ms_per_inference_step'])"
done
