# bf16 mode: parity test (bar 1e-2) and the imagenet224 shape at B=64, fp32 against bf16
mkdir -p gpurun_out
timeout -k 10 400 python -u -m pytest tests/test_gpu_parity.py -k bf16 -x -q -s --timeout 300 --timeout-method thread -p no:cacheprovider > gpurun_out/bf16_test.log 2>&1
rc=$?; grep -E "bf16 max rel err|passed|failed" gpurun_out/bf16_test.log | tail -12; [ $rc -eq 0 ] || exit $rc
bash scripts/i224.sh
