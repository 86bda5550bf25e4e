# rocprofv3 kernel-trace stats of the default bench workload; summary copied to gpurun_out/stats_<tag>.csv
tag=${1:-run}; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_$tag
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -- python3 bench.py --steps 30 --warmup 10 --no-cpu-baseline "$@" > gpurun_out/prof_$tag.log 2>&1
echo exit=$?
f=$(find gpurun_out/prof_$tag -name "*kernel_stats.csv" | head -1)
cp "$f" gpurun_out/stats_$tag.csv && cut -d, -f1-7 gpurun_out/stats_$tag.csv | head -30
tail -1 gpurun_out/prof_$tag.log | cut -c1-400
