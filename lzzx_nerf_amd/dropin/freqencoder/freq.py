"""reference path freqencoder/freq.py -> lzzx_nerf_amd.freqencoder"""
from lzzx_nerf_amd.freqencoder import FreqEncoder, _freq_encoder, freq_encode  # noqa: F401
