# dtype_mode 2 (three-term bf16 split GEMMs): parity at the fp32 bars, then cfg 4 / video16 / 12-block per-op timings against fp32
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py -k "bf16x3" -x -q --timeout 300 --timeout-method thread -p no:cacheprovider > gpurun_out/bf16x3_test.log 2>&1
rc=$?; tail -4 gpurun_out/bf16x3_test.log | cut -c1-1500; [ $rc -eq 0 ] || exit $rc
for dt in fp32 bf16x3; do
  for cfg in imagenet224 video16; do
    timeout -k 10 300 python bench.py --config $cfg --steps 3 --warmup 1 --no-cpu-baseline --no-extra --dtype $dt > gpurun_out/x3.json 2>/dev/null || exit 1
    python3 -c "
import json; r=json.load(open('gpurun_out/x3.json')); print('$dt $cfg', round(r['value']), r['ms_per_step'], r.get('whole_step_mfma_frac')); print(r['kernel_ms_per_inference_step'])"
  done
done
