python -m pytest tests/test_gpu_comm.py -x -q -m gpu > gpurun_out/r02d_comm.txt 2>&1; tail -15 gpurun_out/r02d_comm.txt
python bench.py --steps 100 --warmup 10 > gpurun_out/r02d_bench.json 2> gpurun_out/r02d_bench.err; cat gpurun_out/r02d_bench.json; tail -3 gpurun_out/r02d_bench.err
python tools/ab_kernel.py --rounds 3 build_ab/libtr_estslice.so transmission_renderer_amd/libtr_shade.so > gpurun_out/r02d_ab.txt 2>&1; tail -4 gpurun_out/r02d_ab.txt
