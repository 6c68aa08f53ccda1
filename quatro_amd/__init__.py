"""quatro_amd — MI355X (gfx950) back end for the url-kaist/Quatro registration hot path.

Layout: csrc/ (HIP kernels + the C ABI of include/quatro_hip.h), lib.py (ctypes binding; raises if the
HIP library is missing — there is no CPU fallback), api.py (host-side mirror of the reference's
Quatro / FPFHManager / voxelize interface), synth.py (synthetic KITTI-64-shaped inputs), dist.py
(pair sharding + final gather), build.py (hipcc driver).
"""
__all__ = ["lib", "api", "synth", "dist", "build"]
