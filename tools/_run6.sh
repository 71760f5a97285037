for lib in dual8 dual6; do
TR_TEST_LIB=build_ab/libtr_$lib.so python - <<'PY' > gpurun_out/r02h_check_$lib.txt 2>&1
import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from transmission_renderer_amd import _lib
_lib.LIB_PATH = os.path.abspath(os.environ["TR_TEST_LIB"])
import pytest
sys.exit(pytest.main(["-x", "-q", "-m", "gpu", "tests/test_gpu_parity.py", "-k", "transmissive_pass_parity or opaque_pass_parity or 4k_properties or assigned"]))
PY
tail -2 gpurun_out/r02h_check_$lib.txt
done
python tools/ab_kernel.py --rounds 3 build_ab/libtr_nodual.so build_ab/libtr_dual8.so build_ab/libtr_dual6.so > gpurun_out/r02h_ab.txt 2>&1; tail -4 gpurun_out/r02h_ab.txt
