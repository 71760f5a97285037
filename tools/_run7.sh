python -m pytest tests -x -q -m gpu > gpurun_out/r02i_tests.txt 2>&1; tail -5 gpurun_out/r02i_tests.txt
