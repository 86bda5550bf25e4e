# SQ counters of the block kernels and GEMMs (one pass, 8 SQ counters).  usage: pmc_sq.sh [tag] [bench args...]
tag=${1:-cfg2}; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmc_sq_$tag
timeout -s KILL 180 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d gpurun_out/pmc_sq_$tag -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extra "$@" > gpurun_out/pmc_sq_$tag.log 2>&1
rc=$?; echo exit=$rc; [ $rc -eq 0 ] || exit $rc
f=$(find gpurun_out/pmc_sq_$tag -name "*counter_collection.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].replace("(anonymous namespace)::", "")[:48]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
for k, d in agg.items():
    if "pc_block" in k or "gemm" in k or "rows_" in k or "attn" in k:
        print(k, {c: round(v / cnt[(k, c)]) for c, v in d.items()})
PY
