# SQ counters of the 12block per-op path, per (kernel, grid)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmc1
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d gpurun_out/pmc1 -- python3 bench.py --config 12block --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/pmc1.log 2>&1
echo exit=$?
f=$(find gpurun_out/pmc1 -name "*counter_collection.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
rows = list(csv.DictReader(open(sys.argv[1])))
print(list(rows[0].keys()))
for r in rows:
    k = (r["Kernel_Name"][:44], r.get("Grid_Size", ""))
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
for k, d in agg.items():
    if "gemm" in k[0] or "attn" in k[0]:
        print(k, {c.replace("SQ_", ""): round(v / cnt[(k, c)]) for c, v in d.items()})
PY
