set -e
for c in 1 2 3 4 5; do timeout -k 10 400 python bench.py --config $c 2>/dev/null | tail -1 > gpurun_out/bench_c$c.json; python -c "import json;d=json.load(open('gpurun_out/bench_c$c.json'));print($c, round(d['ms_per_step'],4), round(d['value']/1e6,1),'Mcells/s', round(d['roofline']['frac'],3), d.get('pruned_row_frac'))"; done
timeout -k 10 200 python tools/joint_bench.py --config 2 2>&1 | grep "config 2:"
timeout -k 10 200 python tools/joint_bench.py 2>&1 | grep "config 3:"
timeout -k 10 200 python tools/fc_joint_bench.py 2>&1 | grep -v amdgpu.ids | tail -4
timeout -k 10 200 python tools/crop_bench.py 2>&1 | grep "B="
