run() { echo "== $ARGS $*"; env "$@" CORA_TRACE=1 timeout -k 10 200 python bench.py --gpus 1 --no-cpu-baseline $ARGS 2>&1 | grep -E "^trace" | python3 -c "
import sys,ast
for l in sys.stdin:
    t=ast.literal_eval(l.split(':',1)[1].strip().replace('inf','1e999').replace('nan','1e999'))
    print(['%.5g'%a[0] for a in t])"; }
ARGS="--steps 5 --warmup 2"; run A=1; run CORA_NO_GRAPH=1
ARGS="--steps 6 --warmup 1"; run A=1
ARGS="--steps 3 --warmup 4"; run A=1
ARGS="--steps 9 --warmup 4"; run A=1; run CORA_NO_GRAPH=1
