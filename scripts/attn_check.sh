# general-path parity (ragged N, cfg4 / cfg5 shapes, per-op 12-block) + imagenet224 bench fp32/bf16 + video16
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize_oracle.py -x -q --timeout 300 --timeout-method thread -p no:cacheprovider -k "n36 or cfg4 or cfg5 or d96 or cfg3 or tiny or 0block" > gpurun_out/attn_test.log 2>&1
rc=$?; tail -3 gpurun_out/attn_test.log; [ $rc -eq 0 ] || exit $rc
CORA_NO_FUSED=1 timeout -k 10 300 python -u -m pytest tests/test_gpu_parity.py -x -q --timeout 300 --timeout-method thread -p no:cacheprovider -k "cfg2_shape or d64 or cfg3_shape" > gpurun_out/attn_test2.log 2>&1
rc=$?; tail -2 gpurun_out/attn_test2.log; [ $rc -eq 0 ] || exit $rc
bash scripts/i224.sh
timeout -k 10 300 python bench.py --config video16 --steps 3 --warmup 1 --no-cpu-baseline --no-extra > gpurun_out/v16.json 2>/dev/null || exit 1
python3 -c "
import json; r=json.load(open('gpurun_out/v16.json')); print('video16', r['value'], r['ms_per_step']); print(r['kernel_ms_per_inference_step'])"
