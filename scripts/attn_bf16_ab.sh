# bf16 mode attention (backward on bf16 MFMA, forward on split bf16): parity first, then timings against the fp32 forward
mkdir -p gpurun_out
for v in "CORA_X=1" "CORA_BF16_FP32_ATTN_FWD=1"; do
echo "== $v"
env $v timeout -k 10 400 python -u -m pytest tests/test_gpu_parity.py -k bf16 -x -q -s --timeout 300 --timeout-method thread -p no:cacheprovider > gpurun_out/bf16_test.log 2>&1
rc=$?; grep -E "bf16 max rel err|passed|failed" gpurun_out/bf16_test.log | tail -12; [ $rc -eq 0 ] || exit $rc
done
for v in "CORA_X=1" "CORA_ATTN_BF16_ROWS=64" "CORA_BF16_FP32_ATTN_FWD=1"; do
  echo "== $v"
  for cfg in imagenet224 video16; do
    env $v timeout -k 10 300 python bench.py --config $cfg --steps 3 --warmup 1 --no-cpu-baseline --no-extra --dtype bf16 > gpurun_out/ab.json 2>/dev/null || exit 1
    python3 -c "
import json; r=json.load(open('gpurun_out/ab.json')); print('$cfg', round(r['value']), r['ms_per_step']); print(r['kernel_ms_per_inference_step'])"
  done
done
