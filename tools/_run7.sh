python -m pytest tests/test_gpu_raster.py tests/test_gpu_frame_zones.py tests/test_cli.py tests/test_gpu_textures.py -x -q -m gpu > gpurun_out/r02t.txt 2>&1; grep -n "passed\|failed" gpurun_out/r02t.txt | tail -2
python tools/make_demo_gltf.py gpurun_out/demo.glb > /dev/null
python tools/gpu_bench_frame.py gpurun_out/demo.glb 2>&1 | tail -1
python tools/gpu_bench_frame.py meshes 2>&1 | tail -1
