# weight-gradient parity on the per-op learning phase (all shapes incl. vode LayerNorm) + 12-block / cfg 4 benches
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize_oracle.py tests/test_gpu_fullsize.py -x -q --timeout 300 --timeout-method thread -p no:cacheprovider > gpurun_out/learn_test.log 2>&1
rc=$?; tail -3 gpurun_out/learn_test.log; [ $rc -eq 0 ] || exit $rc
bash scripts/wide_bench.sh && bash scripts/i224.sh
