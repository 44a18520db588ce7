# usage: kstat.sh <outdir> -- python3 /root/repo/tools/x.py args : rocprofv3 kernel stats, our kernels only
out=$1; shift 2
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/$out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$out -o run -- "$@" > /dev/null 2> $R/gpurun_out/$out/err.txt || { tail -3 $R/gpurun_out/$out/err.txt; exit 1; }
cd $R && python3 - $out <<'PY'
import csv, glob, sys
f = glob.glob(f"gpurun_out/{sys.argv[1]}/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    n = r["Name"]
    if "at::native" in n and "k_" not in n.split("(")[0]: n = "torch:" + n.split("<")[0][-40:]
    print(f"{n[:90]:90s} calls={r['Calls']:>4s} avg_us={float(r['AverageNs'])/1e3:9.1f} pct={r['Percentage']}")
PY
