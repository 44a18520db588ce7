#!/bin/bash
# per-dispatch times of the sweep kernels (k_sweep<K>, k_sweep_wide, k_skew_records) of tools/sweep_probe.py from a
# rocprofv3 kernel trace, grouped by label length.  usage: sweep_trace.sh <out dir name> [B]   (RNNT_MI355X_LIB selects a variant)
cd /tmp && export TMPDIR=/tmp && timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/$1 -- python $GRAFT_REPO_ROOT/tools/sweep_probe.py $2 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT && python - "$1" <<'PY'
import csv, glob, sys
f = glob.glob(f'gpurun_out/{sys.argv[1]}/*/*kernel_trace.csv')[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
sweeps = [r for r in rows if 'k_sweep' in r['Kernel_Name']]
skews = [r for r in rows if 'k_skew' in r['Kernel_Name']]
us = lambda r: (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
Us = (63, 100, 199, 399)
per = len(sweeps) // len(Us)
out = []
for i, U in enumerate(Us):
    g = sweeps[i * per:(i + 1) * per][1:]
    name = g[0]['Kernel_Name'].split('(')[0].split('::')[-1]
    t = sum(map(us, g)) / len(g)
    out.append(f"U={U} {name} {t:.1f} us {t * 1e3 / (500 + U):.0f} ns/diag")
if skews:
    out.append("skew " + " ".join(f"{us(r):.1f}" for r in skews[1::5]))
print(sys.argv[1], "|", " | ".join(out))
PY
