#!/bin/bash
# bandwidth of the two streaming passes (k_lse*, k_grad*) against the row length: rocprofv3 kernel trace of tools/small_v_probe.py
# usage: small_v_stats.sh <out dir name> [V | B:T:U:V ...]
out=$1; shift
cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/$out -- python $GRAFT_REPO_ROOT/tools/small_v_probe.py "$@" > /dev/null 2>&1
cd $GRAFT_REPO_ROOT && python - "$out" "$@" <<'PY'
import csv, glob, os, sys
f = sorted(glob.glob(f'gpurun_out/{sys.argv[1]}/*/*kernel_trace.csv'), key=os.path.getmtime)[-1]
rows = [r for r in csv.DictReader(open(f)) if 'k_lse' in r['Kernel_Name'] or 'k_grad' in r['Kernel_Name']]
shapes = [tuple(int(v) for v in a.split(':')) if ':' in a else (64, 500, 100, int(a)) for a in (sys.argv[2:] or ['32', '40', '62', '64'])]
for gi, (B, T, U, V) in enumerate(shapes):
    for r in rows[gi * 10:(gi + 1) * 10][-2:]:   # five iterations per V: the last lse / grad pair
        dur = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1000
        nbytes = B * T * (U + 1) * V * 4
        lse = 'k_lse' in r['Kernel_Name']
        print(f"B={B} T={T} U={U} V={V:3d} {'lse ' if lse else 'grad'} {dur:7.1f} us  {(1 if lse else 2) * nbytes / dur / 1e6:5.2f} TB/s")
PY
