# pytest -m gpu, then bench.py on every BASELINE configuration -> gpurun_out/$1/
D=gpurun_out/${1:-all}
mkdir -p $D
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > $D/pytest.txt 2>&1; echo "pytest exit $?"; tail -3 $D/pytest.txt
for c in 3 1 2 4 5; do
  timeout -k 10 400 python -u bench.py --config $c > $D/bench_c$c.json 2> $D/bench_c$c.err || { echo "bench $c failed"; tail -5 $D/bench_c$c.err; exit 1; }
  python - <<PY
import json
j=json.loads(open("$D/bench_c$c.json").read().strip().splitlines()[-1])
k={e["name"]:round(e["ms"],3) for e in j["roofline"]["kernels"]}
print("config $c: step %.3f ms frac %.3f pruned %.3f" % (j["ms_per_step"], j["roofline"]["frac"], j["pruned_row_frac"]), k, {n:(round(l["ms_per_step"],3), round(l["frac"],3)) for n,l in j.get("legs",{}).items()}, "cpu", round(j["cpu_baseline"]["value"]/1e6,2))
PY
done
for c in 3 2 1; do
timeout -k 10 300 python -u bench.py --config $c --graph --no-cpu-baseline --no-legs > $D/bench_c${c}_graph.json 2> $D/bench_c${c}_graph.err; echo "graph exit $?"
python -c "
import json
j=json.loads(open('$D/bench_c${c}_graph.json').read().strip().splitlines()[-1]); print('config $c graph: step %.4f ms frac %.3f' % (j['ms_per_step'], j['roofline']['frac']))"
done
