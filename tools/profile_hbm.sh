# rocprofv3 kernel stats + HBM counter passes of the default bench workload (config 3) -> gpurun_out/prof_c<N>/
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
CFG=${1:-3}
OUT=$R/gpurun_out/prof_c$CFG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o run -- python3 $R/bench.py --config $CFG --no-cpu-baseline --no-legs --steps 10 --warmup 2 > $OUT/bench_under_rocprof.json 2> $OUT/stats.err || { tail -5 $OUT/stats.err; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o run -- python3 $R/bench.py --config $CFG --no-cpu-baseline --no-legs --steps 2 --warmup 1 --ring 3 > /dev/null 2> $OUT/fetch.err || { tail -5 $OUT/fetch.err; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -o run -- python3 $R/bench.py --config $CFG --no-cpu-baseline --no-legs --steps 2 --warmup 1 --ring 3 > /dev/null 2> $OUT/write.err || { tail -5 $OUT/write.err; exit 1; }
cd $R
python3 tools/pmc_traffic.py $OUT/fetch $OUT/write $OUT/traffic_kernels.json
find $OUT/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
head -12 $OUT/kernel_stats.csv
# keep only the small summaries (the traces are large)
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -size +5M -delete
du -sh $OUT
