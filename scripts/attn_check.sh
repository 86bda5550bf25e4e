# general-path parity (ragged N, cfg4 / cfg5 shapes) + imagenet224 bench fp32/bf16
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py -x -q --timeout 300 --timeout-method thread -p no:cacheprovider -k "n36 or cfg4 or cfg5 or d96 or cfg3_shape or tiny" > gpurun_out/attn_test.log 2>&1
rc=$?; tail -3 gpurun_out/attn_test.log; [ $rc -eq 0 ] || exit $rc
bash scripts/i224.sh
