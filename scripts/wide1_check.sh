# one-workgroup wide path (D=96, D=128 with heads of 64 / 128): parity incl. INIT / recon chain, full-size fused-vs-per-op, grid bench
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_fullsize_oracle.py -x -q --timeout 300 --timeout-method thread -p no:cacheprovider -k "d96 or D96 or D128 or cfg3 or init_forward" > gpurun_out/w1c.log 2>&1
rc=$?; tail -3 gpurun_out/w1c.log; [ $rc -eq 0 ] || exit $rc
for cfg in "96 4" "96 1" "128 2"; do set -- $cfg
  timeout -k 10 300 python bench.py --config 12block --hidden-size $1 --num-heads $2 --steps 6 --warmup 2 --no-cpu-baseline --no-extra > gpurun_out/g.json 2>/dev/null || exit 1
  python3 -c "
import json; r=json.load(open('gpurun_out/g.json')); print('D=$1 H=$2', round(r['value']), round(r['ms_per_step'],2))"
done
