python -m pytest tests/test_gpu_textures.py tests/test_gpu_golden.py tests/test_gpu_raster.py tests/test_gpu_parity.py tests/test_cli.py tests/test_gpu_frame_zones.py -x -q -m gpu > gpurun_out/r02l_tests.txt 2>&1; tail -5 gpurun_out/r02l_tests.txt
python tools/make_demo_gltf.py gpurun_out/demo.glb > /dev/null
python tools/gpu_bench_frame.py gpurun_out/demo.glb 2>&1 | tail -1
python tools/gpu_bench_frame.py meshes 2>&1 | tail -1
python tools/gpu_bench_frame.py gpurun_out/demo.glb 1920 1080 2>&1 | tail -1
python tools/gpu_bench_textured.py 2>&1 | tail -2
python -m transmission_renderer_amd.cli gpurun_out/demo.glb --width 3840 --height 2160 --timings --out gpurun_out/demo4k.png 2>&1 | tail -11
