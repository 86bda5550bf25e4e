mkdir -p gpurun_out
for dt in fp32 bf16x3 bf16; do
  CORA_NO_FUSED=1 timeout -k 10 300 python bench.py --config 12block --steps 3 --warmup 1 --no-cpu-baseline --no-extra --dtype $dt > gpurun_out/x3.json 2>/dev/null || exit 1
  python3 -c "
import json; r=json.load(open('gpurun_out/x3.json')); print('$dt 12block per-op', round(r['value']), r['ms_per_step'], r.get('whole_step_mfma_frac')); print(r['kernel_ms_per_inference_step'])"
done
