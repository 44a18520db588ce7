set -x
mkdir -p gpurun_out/r2b
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/r2b/pytest.txt 2>&1 && \
timeout -k 10 120 python -u -m pytest tests/test_zz_no_alloc.py -m gpu -x -v --timeout 120 --timeout-method thread > gpurun_out/r2b/noalloc_alone.txt 2>&1
echo "exit $?"
tail -3 gpurun_out/r2b/pytest.txt; tail -5 gpurun_out/r2b/noalloc_alone.txt
