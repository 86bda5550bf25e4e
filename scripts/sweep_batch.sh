for b in "$@"; do
  timeout -k 10 120 python bench.py --batch $b --steps 4 --warmup 2 --no-cpu-baseline 2>/dev/null > gpurun_out/sb_$b.json
  python3 - $b <<'PY'
import sys, json
b = sys.argv[1]
d = json.loads(open(f"gpurun_out/sb_{b}.json").read().strip().splitlines()[-1])
print(b, {k: round(v, 4) for k, v in d["kernel_ms_per_inference_step"].items()}, round(d["value"]))
PY
done
