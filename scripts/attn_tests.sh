# attention-related GPU tests after a kernel change
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest tests/test_gpu_parity.py -x -q --timeout 300 --timeout-method thread -p no:cacheprovider -k "bf16 or attention or n144 or cfg4 or cfg5 or n36" > gpurun_out/attn_tests.log 2>&1
rc=$?; tail -5 gpurun_out/attn_tests.log; exit $rc
