"""lzzx_nerf_amd -- MI355X (gfx950) implementation of the nerf_triplane volumetric-rendering hot path.

Operator API (drop-in for the reference's packages):
    lzzx_nerf_amd.encoding.get_encoder, .gridencoder.GridEncoder / grid_encode, .shencoder.SHEncoder,
    .freqencoder.FreqEncoder, .raymarching.* (17 functions)
Fast path (no reference counterpart): .head.FusedTriplaneHead, .renderer.TriplaneRenderer, .dist

Everything computes in liblzzx_nerf_hip.so (hand-written HIP, C ABI in include/lzzx_nerf_hip.h); importing a
module that needs it raises if the library is missing -- there is no CPU / eager fallback.
"""
__version__ = "0.1.0"
