# rocprofv3 kernel trace (every launch with its timestamps) of the default bench workload, reduced on the box to
# gpurun_out/trace_<tag>.txt: one line per launch of the block kernel (start offset us, duration us, name)
tag=${1:-run}; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_$tag
timeout -s KILL 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_$tag -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-extra "$@" > gpurun_out/prof_$tag.log 2>&1
echo exit=$?
f=$(find gpurun_out/prof_$tag -name "*kernel_trace.csv" | head -1)
python3 - "$f" gpurun_out/trace_$tag.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
with open(sys.argv[2], "w") as o:
    prev_end = t0
    for r in rows:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        o.write("%10.1f %8.1f gap %7.1f  %s\n" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, r["Kernel_Name"][:70]))
        prev_end = e
PY
rm -rf gpurun_out/prof_$tag
tail -1 gpurun_out/prof_$tag.log | cut -c1-300
