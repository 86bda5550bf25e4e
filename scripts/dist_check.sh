# data-parallel path on one GPU: world-1 RCCL tests, the bench forced through cora_step_dist (graphs, then eager collectives)
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests/test_gpu_dp_exact.py tests/test_gpu_parity.py -x -q --timeout 300 --timeout-method thread -p no:cacheprovider -k "rccl or two_batches or bf16" > gpurun_out/dist_test.log 2>&1
rc=$?; tail -3 gpurun_out/dist_test.log; [ $rc -eq 0 ] || exit $rc
CORA_FORCE_DP=1 timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra > gpurun_out/dist_bench_graph.json 2> gpurun_out/dist_bench_graph.err || exit 1
CORA_FORCE_DP=1 CORA_DIST_NO_GRAPH=1 timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra > gpurun_out/dist_bench_eager.json 2> gpurun_out/dist_bench_eager.err || exit 1
python3 -c "
import json
for t in ('graph','eager'):
    txt=open('gpurun_out/dist_bench_%s.json'%t).read().strip().splitlines()
    r=json.loads(txt[-1]); print(t, len(txt), 'stdout line(s)', r['value'], r['ms_per_step'], r['config']['parallelism'])"
