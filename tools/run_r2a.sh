set -x
mkdir -p gpurun_out/r2a
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/r2a/pytest.txt 2>&1 && \
timeout -k 10 120 python -u -m pytest tests/wip_no_alloc.py -x -v --timeout 120 --timeout-method thread > gpurun_out/r2a/noalloc.txt 2>&1 && \
timeout -k 10 300 python -u bench.py > gpurun_out/r2a/bench_c3.json 2> gpurun_out/r2a/bench_c3.err && \
timeout -k 10 300 python -u bench.py --gpus 2 --backend gloo --config 3 --batch 16 --steps 10 > gpurun_out/r2a/bench_2rank.json 2> gpurun_out/r2a/bench_2rank.err && \
timeout -k 10 300 python -u tools/prune_study.py > gpurun_out/r2a/prune_study.jsonl 2>&1
echo "exit $?"
tail -3 gpurun_out/r2a/pytest.txt; tail -5 gpurun_out/r2a/noalloc.txt; cat gpurun_out/r2a/bench_c3.err | tail -5; cat gpurun_out/r2a/bench_2rank.err | tail -5
