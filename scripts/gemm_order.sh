# fp32 per-op GEMM with the XCD-aware column-fastest tile order: parity (whole parity files), then timings
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize_oracle.py tests/test_gpu_fullsize.py -x -q --timeout 300 --timeout-method thread -p no:cacheprovider > gpurun_out/go_test.log 2>&1
rc=$?; tail -3 gpurun_out/go_test.log; [ $rc -eq 0 ] || exit $rc
for cfg in imagenet224 video16 12block; do
  timeout -k 10 300 python bench.py --config $cfg --steps 4 --warmup 2 --no-cpu-baseline --no-extra > gpurun_out/go.json 2>/dev/null || exit 1
  python3 -c "
import json; r=json.load(open('gpurun_out/go.json')); print('$cfg', round(r['value']), r['ms_per_step']); print(r['kernel_ms_per_inference_step'])"
done
CORA_NO_FUSED=1 timeout -k 10 300 python bench.py --config 12block --steps 4 --warmup 2 --no-cpu-baseline --no-extra > gpurun_out/go.json 2>/dev/null || exit 1
python3 -c "
import json; r=json.load(open('gpurun_out/go.json')); print('12block per-op', round(r['value']), r['ms_per_step']); print(r['kernel_ms_per_inference_step'])"
