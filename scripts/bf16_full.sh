mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests/test_gpu_fullsize_oracle.py -k bf16 -x -q -s --timeout 300 --timeout-method thread -p no:cacheprovider > gpurun_out/bf16_full.log 2>&1
rc=$?; grep -E "^bf16 |passed|failed|Error|error" gpurun_out/bf16_full.log | cut -c1-600 | tail -12; exit $rc
