for lib in rnnt_amd/csrc/librnnt_mi355x.so tools/variants/fos.so tools/variants/old.so; do
RNNT_MI355X_LIB=$PWD/$lib bash tools/small_v_stats.sh r2v_$(basename $lib .so) 50 62 90 126 > /dev/null 2>&1
python - r2v_$(basename $lib .so) <<'PY'
import csv, glob, os, sys
f = sorted(glob.glob(f'gpurun_out/{sys.argv[1]}/*/*kernel_trace.csv'), key=os.path.getmtime)[-1]
rows = [r for r in csv.DictReader(open(f)) if 'k_lse' in r['Kernel_Name'] or 'k_grad' in r['Kernel_Name']]
out=[]
for gi, V in enumerate([50,62,90,126]):
    B,T,U=64,500,100
    for r in rows[gi*10:(gi+1)*10][-2:]:
        dur=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1000
        nbytes=B*T*(U+1)*V*4; lse='k_lse' in r['Kernel_Name']
        out.append(f"V={V} {r['Kernel_Name'].split('::')[-1][:14]} {dur:.0f}us {(1 if lse else 2)*nbytes/dur/1e6:.2f}")
print(sys.argv[1], ' | '.join(out))
PY
done
