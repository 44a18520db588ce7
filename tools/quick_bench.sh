# usage: quick_bench.sh <outdir> <config> [config ...] : pytest -m gpu, then a short bench per config
D=gpurun_out/$1; shift
mkdir -p $D
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > $D/pytest.txt 2>&1; echo "pytest exit $?"; tail -3 $D/pytest.txt
for c in "$@"; do
  timeout -k 10 400 python -u bench.py --config $c --no-cpu-baseline --no-legs > $D/q_c$c.json 2> $D/q_c$c.err || { echo "bench $c failed"; tail -5 $D/q_c$c.err; exit 1; }
  python - <<PY
import json
j=json.loads(open("$D/q_c$c.json").read().strip().splitlines()[-1])
k={e["name"]:round(e["ms"],3) for e in j["roofline"]["kernels"]}
print("config $c: step %.3f ms frac %.3f pruned %.3f" % (j["ms_per_step"], j["roofline"]["frac"], j["pruned_row_frac"]), k)
PY
done
