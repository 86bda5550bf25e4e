timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py -x -q --timeout 300 --timeout-method thread -p no:cacheprovider -k "cfg3_shape or d96 or cfg2_shape_B4" > gpurun_out/w1.log 2>&1
rc=$?; tail -5 gpurun_out/w1.log; [ $rc -eq 0 ] || exit $rc
bash scripts/wide_bench.sh || exit 1
