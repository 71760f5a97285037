python -m pytest tests -x -q -m gpu > gpurun_out/r02n_tests.txt 2>&1; tail -4 gpurun_out/r02n_tests.txt
