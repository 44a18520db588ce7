# usage: pmc_kernel.sh <outdir> "<counters>" <kernel substring> -- python3 script args: average of the counters over that kernel's dispatches
out=$1; ctrs=$2; kern=$3; shift 4
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/$out
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc $ctrs --output-format csv -d $R/gpurun_out/$out -o run -- "$@" > /dev/null 2> $R/gpurun_out/$out/err.txt || { tail -3 $R/gpurun_out/$out/err.txt; exit 1; }
cd $R && python3 - $out "$kern" <<'PY'
import csv, glob, sys
from collections import defaultdict
acc = defaultdict(list)
for f in glob.glob(f"gpurun_out/{sys.argv[1]}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if sys.argv[2] in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    print(f"{k:32s} n={len(v)} mean={sum(v)/len(v):.4g}")
PY
