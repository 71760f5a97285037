python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py -x -q -m gpu > gpurun_out/r02e_tests.txt 2>&1; tail -3 gpurun_out/r02e_tests.txt
python tools/ab_kernel.py --rounds 3 build_ab/libtr_estslice.so build_ab/libtr_nosunfirst.so transmission_renderer_amd/libtr_shade.so > gpurun_out/r02e_ab.txt 2>&1; tail -4 gpurun_out/r02e_ab.txt
python tools/ab_kernel.py --rounds 1 build_ab/libtr_timing.so > gpurun_out/r02e_timing.txt 2>&1; tail -4 gpurun_out/r02e_timing.txt
for b in 1024 2048 4096; do echo "TR_BLOCKS_PER_XCD=$b"; TR_BLOCKS_PER_XCD=$b python tools/ab_kernel.py --rounds 2 transmission_renderer_amd/libtr_shade.so 2>&1 | tail -1; done
