python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "config2 or config3" > gpurun_out/r02j_tests.txt 2>&1; tail -15 gpurun_out/r02j_tests.txt
