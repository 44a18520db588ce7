mkdir -p gpurun_out/r2d
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/r2d/pytest.txt 2>&1; echo "pytest exit $?"; tail -5 gpurun_out/r2d/pytest.txt
timeout -k 10 900 python -u tools/variant_bench.py --kinds n1,p28,peaked nospan s8x8 s4x8 s8x4 s16x4 s6x8 s16x8 2>&1 | tee gpurun_out/r2d/variants2.txt
