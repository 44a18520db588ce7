#!/bin/bash
# per-diagonal time of k_sweep<K>: rocprofv3 kernel stats of tools/sweep_probe.py.  usage: sweep_stats.sh <out dir name> [B]
cd /tmp && export TMPDIR=/tmp && timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/$1 -- python $GRAFT_REPO_ROOT/tools/sweep_probe.py $2 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT && python - "$1" <<'PY'
import csv, glob, sys
f = glob.glob(f'gpurun_out/{sys.argv[1]}/*/*kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    if 'k_sweep' in r['Name']:
        K = r['Name'].split('k_sweep<')[1].split('>')[0]
        U = {'1': 63, '2': 100, '4': 199, '8': 399}[K]
        avg = float(r['AverageNs'])
        print(f"K={K} avg {avg / 1000:.1f} us  per diagonal {avg / (500 + U):.0f} ns")
PY
