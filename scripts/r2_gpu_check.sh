# Round-2 GPU check: the whole GPU suite, the allocation-free entry point's test, a short bench.
# Each GPU step under its own time limit, chained with && (nothing runs after a failure).
mkdir -p gpurun_out
timeout -k 10 1000 python -u -m pytest tests -m gpu -x -q --timeout 400 --timeout-method thread -p no:cacheprovider > gpurun_out/r2_pytest.log 2>&1
rc=$?; tail -5 gpurun_out/r2_pytest.log; [ $rc -eq 0 ] || exit $rc
if [ -f tests/wip_no_alloc.py ]; then
  timeout -k 10 300 python -u -m pytest tests/wip_no_alloc.py -x -q --timeout 120 --timeout-method thread -p no:cacheprovider > gpurun_out/r2_wip.log 2>&1
  rc=$?; tail -5 gpurun_out/r2_wip.log; [ $rc -eq 0 ] || exit $rc
fi
timeout -k 10 400 python bench.py --steps 20 --warmup 3 > gpurun_out/r2_bench.json 2> gpurun_out/r2_bench.err
rc=$?; tail -c 600 gpurun_out/r2_bench.err; cut -c1-700 gpurun_out/r2_bench.json; exit $rc
