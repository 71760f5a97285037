"""The multi-GPU surface of the C ABI on the one GPU the test box has (run with -m gpu): RCCL is loaded with dlopen,
a one-rank communicator is created from a unique id, and the in-place band all-gather runs through ncclAllGather.
N > 1 ranks cannot share a GPU under RCCL: the N-rank path is covered on CPU over gloo (tests/test_sharding_cpu.py,
tests/test_bench_launch.py) and on hardware by the driver's multi-GPU bench."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


def test_one_rank_communicator_and_inplace_allgather():
    from transmission_renderer_amd import wire
    from transmission_renderer_amd.renderer import TransmissionRenderer
    r = TransmissionRenderer(0)
    lib = r.lib
    ident = (C.c_uint8 * 128)()
    assert lib.tr_comm_unique_id(C.byref(ident)) == 0
    assert any(ident)                                                 # RCCL wrote an id
    comm = C.c_void_p()
    assert lib.tr_comm_create(r._ctx, C.byref(ident), 1, 0, C.byref(comm)) == 0 and comm.value
    assert lib.tr_comm_last_error(comm) == 0
    rng = np.random.default_rng(1)
    for dt, fmt in ((torch.float16, wire.FORMAT_RGBA16F), (torch.float32, wire.FORMAT_RGBA32F)):
        frame = torch.from_numpy(rng.random((24, 40, 4), dtype=np.float32)).to(r.device).to(dt)
        want = frame.clone()
        assert lib.tr_allgather_frame(r._ctx, comm, frame.data_ptr(), 40, 24, fmt, torch.cuda.current_stream().cuda_stream) == 0
        torch.cuda.synchronize()
        assert torch.equal(frame, want)                               # one rank: its band is the frame
    ldr = torch.from_numpy(rng.integers(0, 256, (24, 40, 4), dtype=np.uint8)).to(r.device)      # the tonemapped frame
    want = ldr.clone()
    assert lib.tr_allgather_frame(r._ctx, comm, ldr.data_ptr(), 40, 24, wire.FORMAT_RGBA8, torch.cuda.current_stream().cuda_stream) == 0
    torch.cuda.synchronize()
    assert torch.equal(ldr, want)
    # argument checks
    assert lib.tr_allgather_frame(r._ctx, comm, None, 40, 24, wire.FORMAT_RGBA16F, None) == 1
    assert lib.tr_allgather_frame(r._ctx, comm, frame.data_ptr(), 0, 24, wire.FORMAT_RGBA16F, None) == 1
    bad = C.c_void_p()
    assert lib.tr_comm_create(r._ctx, C.byref(ident), 2, 2, C.byref(bad)) == 1 and not bad.value
    assert lib.tr_comm_destroy(comm) == 0
    r.close()


def test_compositor_single_rank_is_a_no_op():
    from transmission_renderer_amd import sharded
    comp = sharded.Compositor(1, 0)
    frame = torch.ones((8, 4, 4), dtype=torch.float16, device="cuda")
    comp.allgather_rows(frame)
    assert comp.backend == "none" and bool((frame == 1).all())
