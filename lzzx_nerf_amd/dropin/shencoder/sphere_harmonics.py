"""reference path shencoder/sphere_harmonics.py -> lzzx_nerf_amd.shencoder"""
from lzzx_nerf_amd.shencoder import SHEncoder, _sh_encoder, sh_encode  # noqa: F401
