"""MI355X-native transmission / volume PBR shading path (hot path of expenses/transmission-renderer).

`wire`       ctypes twins of the reference's wire structs + the host helpers that feed the path
`synthetic`  TGB-v1 synthetic G-buffer scenes (benchmark / parity workload)
`renderer`   the frame recorder over libtr_shade.so (imports torch; needs a HIP device to run)
`sharded`    row-band sharding across the GPUs of a node (torch.distributed / RCCL)
"""
from . import wire, synthetic  # noqa: F401

__all__ = ["wire", "synthetic"]
