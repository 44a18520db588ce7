#!/bin/bash
# first-pass time per case of tools/lse_v_probe.py from a rocprofv3 kernel trace.  usage: lse_v_trace.sh <out dir name>
# (RNNT_MI355X_LIB selects a variant)
cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/$1 -- python $GRAFT_REPO_ROOT/tools/lse_v_probe.py > $GRAFT_REPO_ROOT/gpurun_out/$1.cases 2>&1
cd $GRAFT_REPO_ROOT && python - "$1" <<'PY'
import csv, glob, sys
f = glob.glob(f'gpurun_out/{sys.argv[1]}/*/*kernel_trace.csv')[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
lse = [r for r in rows if 'k_lse' in r['Kernel_Name']]
cases = [l.split()[1:] for l in open(f'gpurun_out/{sys.argv[1]}.cases') if l.startswith('CASE')]
per = len(lse) // len(cases)
us = lambda r: (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
out = []
for i, (V, ragged, cells) in enumerate(cases):
    g = lse[i * per:(i + 1) * per][1:]
    t = sum(map(us, g)) / len(g)
    out.append(f"V={V}{'r' if ragged == '1' else ''} {t:.0f} us {int(cells) * int(V) * 4 / t / 1e6:.2f} TB/s")
print(sys.argv[1], "|", " | ".join(out))
PY
