cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -s KILL 120 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc_clk -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_clk.log 2>&1
echo exit=$?
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/pmc_clk/**/*counter_collection.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
print(rows[0].keys())
agg = collections.defaultdict(list)
for r in rows:
    k = r["Kernel_Name"].split("(")[0]
    dur = None
    if "End_Timestamp" in r and "Start_Timestamp" in r:
        dur = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    agg[k].append((float(r["Counter_Value"]), dur))
for k, v in agg.items():
    if "fused" in k or "update" in k:
        c = sum(a for a, _ in v) / len(v)
        d = [b for _, b in v if b]
        dd = sum(d) / len(d) if d else None
        print(k[:50], "GUI_ACTIVE/launch", c, "dur_ns", dd, "GHz(est)=", (c / 8 / dd) if dd else None)
PY
