"""MI355X-native implementation of the ParSeNet hot path (kNN / edge-conv encoder,
mean-shift clustering, differentiable fitting, Chamfer) behind the reference's Python
call signatures.  Compute runs in hand-written HIP kernels for gfx950 reached through a
C ABI (include/parsenet_hip.h); there is no CPU fallback."""

__version__ = "0.1.0"
