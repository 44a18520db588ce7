D=gpurun_out/r2m; mkdir -p $D
show() { python - "$1" <<'PY'
import json,sys
j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1].split('/')[-1], "n_gpus", j["n_gpus"], "scaling", j["scaling"], "graph", j["config"]["hip_graph"], "step %.4f ms" % j["ms_per_step"], "ranks", j["ranks_ms_per_step"], "value %.1f M" % (j["value"]/1e6), [(k["name"], round(k["ms"],3)) for k in j["roofline"]["kernels"]])
PY
}
BENCH_FORCE_PROCESS_GROUP=1 timeout -k 10 200 python -u bench.py --config 3 --batch 8 --no-cpu-baseline --no-legs > $D/pg1_graph.json 2> $D/pg1_graph.err; echo "exit $?"; grep -v amdgpu.ids $D/pg1_graph.err | tail -3; show $D/pg1_graph.json
BENCH_FORCE_PROCESS_GROUP=1 timeout -k 10 200 python -u bench.py --config 3 --batch 8 --no-cpu-baseline --no-legs --graph off > $D/pg1_eager.json 2> $D/pg1_eager.err; echo "exit $?"; show $D/pg1_eager.json
timeout -k 10 200 python -u bench.py --config 3 --batch 8 --no-cpu-baseline --no-legs > $D/b8_eager_nopg.json 2> $D/b8.err; echo "exit $?"; show $D/b8_eager_nopg.json
timeout -k 10 200 python -u bench.py --config 3 --batch 8 --no-cpu-baseline --no-legs --graph on > $D/b8_graph_nopg.json 2> $D/b8g.err; echo "exit $?"; show $D/b8_graph_nopg.json
timeout -k 10 300 python -u bench.py --gpus 2 --backend gloo --config 3 --batch 16 --steps 10 > $D/gloo2_c3.json 2> $D/gloo2_c3.err; echo "exit $?"; grep -v "amdgpu.ids\|hostname" $D/gloo2_c3.err | tail -3; show $D/gloo2_c3.json
timeout -k 10 300 python -u bench.py --gpus 2 --backend gloo --config 5 --steps 4 --warmup 1 --ring 2 > $D/gloo2_c5.json 2> $D/gloo2_c5.err; echo "exit $?"; grep -v "amdgpu.ids\|hostname" $D/gloo2_c5.err | tail -3; show $D/gloo2_c5.json
