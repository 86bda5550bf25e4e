# fp32 attention with 4 / 8 waves per workgroup (8 waves: next chunk prefetched): parity, then cfg4 and video16 timings
mkdir -p gpurun_out
CORA_ATTN_WAVES=8 timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize_oracle.py -x -q --timeout 300 --timeout-method thread -p no:cacheprovider -k "(n36 or cfg4 or cfg5 or d96 or n144) and not bf16" > gpurun_out/attn_test.log 2>&1
rc=$?; tail -3 gpurun_out/attn_test.log; [ $rc -eq 0 ] || exit $rc
for w in 4 8; do
  echo "== CORA_ATTN_WAVES=$w"
  export CORA_ATTN_WAVES=$w
  for cfg in imagenet224 video16; do
  timeout -k 10 300 python bench.py --config $cfg --steps 3 --warmup 1 --no-cpu-baseline --no-extra > gpurun_out/aw.json 2>/dev/null || exit 1
  python3 -c "
import json; r=json.load(open('gpurun_out/aw.json')); print('$cfg', round(r['value']), r['ms_per_step']); print(r['kernel_ms_per_inference_step'])"
  done
done
