python -m pytest tests -x -q -m gpu > gpurun_out/r02c_gputests.txt 2>&1; tail -15 gpurun_out/r02c_gputests.txt
python tools/ab_kernel.py --rounds 3 build_ab/libtr_r01.so transmission_renderer_amd/libtr_shade.so > gpurun_out/r02c_ab.txt 2>&1; tail -5 gpurun_out/r02c_ab.txt
