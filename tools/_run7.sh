python -m pytest tests/test_gpu_bench_rehearsal.py tests/test_gpu_comm.py -x -q -m gpu > gpurun_out/r02s.txt 2>&1; grep -n "passed\|failed\|Error" gpurun_out/r02s.txt | tail -4
