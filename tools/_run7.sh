python -m pytest tests -x -q -m gpu > gpurun_out/r02n_tests.txt 2>&1; grep -n "passed\|failed" gpurun_out/r02n_tests.txt | tail -2
python tools/gpu_bench_textured.py 2>&1 | tail -2
