D=gpurun_out/r2u; mkdir -p $D
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > $D/pytest.txt 2>&1; echo "pytest exit $?"; tail -3 $D/pytest.txt
RNNT_MI355X_LIB=$PWD/tools/variants/fos.so timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > $D/pytest_fos.txt 2>&1; echo "pytest(flat over small) exit $?"; tail -3 $D/pytest_fos.txt
for c in 2 1; do
for lib in rnnt_amd/csrc/librnnt_mi355x.so tools/variants/fos.so; do
RNNT_MI355X_LIB=$PWD/$lib timeout -k 10 300 python -u bench.py --config $c --no-cpu-baseline --no-legs --graph on --steps 40 > $D/tmp.json 2>$D/tmp.err || tail -3 $D/tmp.err
python - $lib $c <<'PY'
import json,sys
j=json.loads(open("gpurun_out/r2u/tmp.json").read().strip().splitlines()[-1])
print(sys.argv[1].split('/')[-1], "config", sys.argv[2], "graph step %.4f ms frac %.3f" % (j["ms_per_step"], j["roofline"]["frac"]), [(k["name"], round(k["ms"],4)) for k in j["roofline"]["kernels"]])
PY
done; done
RNNT_MI355X_LIB=$PWD/tools/variants/fos.so bash tools/small_v_stats.sh r2u_v 32 40 62 64 80 100 128 2>&1 | grep "TB/s" | awk '{printf "%s %s %s %s | ", $4, $5, $6, $8} NR%2==0 {print ""}' | tee $D/small_v_fos.txt
bash tools/small_v_stats.sh r2u_v2 32 40 62 64 80 100 128 2>&1 | grep "TB/s" | awk '{printf "%s %s %s %s | ", $4, $5, $6, $8} NR%2==0 {print ""}' | tee $D/small_v_default.txt
