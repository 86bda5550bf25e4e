# fp32 vs opt-in bf16 on the per-op path configs (12block: D=128 H=4 L=12; imagenet224: D=384 H=8 N=196)
for cfg in 12block imagenet224; do for dt in fp32 bf16; do
  extra=""; [ $cfg = imagenet224 ] && extra="--batch 64"
  timeout -k 10 300 python bench.py --config $cfg --dtype $dt --steps 3 --warmup 1 --no-cpu-baseline $extra > gpurun_out/bm_${cfg}_${dt}.json 2> gpurun_out/bm_${cfg}_${dt}.err || { tail -5 gpurun_out/bm_${cfg}_${dt}.err; exit 1; }
  python3 -c "
import json,sys
d=json.load(open('gpurun_out/bm_${cfg}_${dt}.json')); print('$cfg','$dt',d['value'],d['ms_per_step'],{k:round(v,4) for k,v in sorted(d.get('kernel_ms_per_inference_step',{}).items(), key=lambda kv:-kv[1])[:6]})"
done; done
