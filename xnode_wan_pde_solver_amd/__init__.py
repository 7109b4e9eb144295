"""MI355X-native XNODE-WAN hot path (gfx950 HIP kernels behind a C ABI, include/xnwan.h).

Layout:
    csrc/        hand-written HIP kernels + the extern "C" entry points  ->  libxnwan.so (built in-tree)
    _lib.py      ctypes binding of the C ABI (fails loudly when the library is missing)
    kernels.py   tensor-level wrappers: shape / dtype checks, stream plumbing
    nets.py      parameter blobs + nn.Module shells with the reference's state_dict keys
    sampling.py  Hypercube / sphere domains and the loader (host RNG parity with the reference)
    engine.py    the fused generator / discriminator step (graph-capturable)
    dist.py      path sharding + the RCCL exchange of the scalar partials and packed gradients
"""
__all__ = ['_lib', 'kernels']
