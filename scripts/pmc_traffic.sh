# HBM traffic of every kernel of the bench: FETCH_SIZE and WRITE_SIZE in separate passes
# (MI355X_MICROARCH.md: TCC slots; FETCH_SIZE is in KB and reads 1/2 of wide coalesced reads on gfx950)
# usage: pmc_traffic.sh [tag] [bench args...]   ->  gpurun_out/pmc_traffic_<tag>.json
tag=${1:-cfg2}; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmc_${tag}_$c
  timeout -s KILL 180 rocprofv3 --pmc $c --output-format csv -d gpurun_out/pmc_${tag}_$c -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extra "$@" > gpurun_out/pmc_${tag}_$c.log 2>&1
  rc=$?; echo $c exit=$rc; [ $rc -eq 0 ] || exit $rc
done
python3 - $tag <<'PY'
import csv, glob, collections, json, sys
tag = sys.argv[1]
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"gpurun_out/pmc_{tag}_{c}/**/*counter_collection.csv", recursive=True)[0]
    agg = collections.defaultdict(float); cnt = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]
        agg[k] += float(r["Counter_Value"]); cnt[k] += 1
    for k in agg:
        out.setdefault(k, {})[c + "_per_launch_KB"] = agg[k] / cnt[k]
        out[k]["launches"] = cnt[k]
json.dump(out, open(f"gpurun_out/pmc_traffic_{tag}.json", "w"), indent=1)
for k, v in sorted(out.items(), key=lambda kv: -kv[1].get("FETCH_SIZE_per_launch_KB", 0))[:8]:
    print(k[:60], v)
PY
