"""`from encoding import get_encoder` (reference: /root/reference/encoding.py) -> lzzx_nerf_amd.encoding"""
from lzzx_nerf_amd.encoding import get_encoder  # noqa: F401
