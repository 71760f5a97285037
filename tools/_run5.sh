python -m pytest tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/r02g_tests.txt 2>&1; tail -3 gpurun_out/r02g_tests.txt
for b in 1 2 3 4 8 16; do echo "TR_XCD_CHUNKS=$b"; TR_XCD_CHUNKS=$b python tools/ab_kernel.py --rounds 2 build_ab/libtr_nosunfirst.so transmission_renderer_amd/libtr_shade.so 2>&1 | tail -2; done
