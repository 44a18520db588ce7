D=gpurun_out/r2s; mkdir -p $D
RNNT_MI355X_LIB=$PWD/tools/variants/flat.so timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > $D/pytest_flat.txt 2>&1; echo "pytest(flat) exit $?"; tail -4 $D/pytest_flat.txt
timeout -k 10 600 python -u tools/variant_bench.py --kinds n1,peaked base flat base flat 2>&1 | tee $D/variants_flat.txt
timeout -k 10 300 python -u tools/variant_bench.py --config 5 --kinds n1 base flat 2>&1 | tee -a $D/variants_flat.txt
timeout -k 10 300 python -u tools/variant_bench.py --config 4 --kinds n1 base flat 2>&1 | tee -a $D/variants_flat.txt
RNNT_MI355X_LIB=$PWD/tools/variants/flat.so bash tools/small_v_stats.sh r2s_v 128 132 160 200 256 300 384 512 768 1024 2>&1 | grep "TB/s" | awk '{printf "%s %s %s %s | ", $4, $5, $6, $8} NR%2==0 {print ""}' | tee $D/small_v_flat.txt
