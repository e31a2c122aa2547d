"""reference path gridencoder/grid.py -> lzzx_nerf_amd.gridencoder"""
from lzzx_nerf_amd.gridencoder import GridEncoder, _grid_encode, grid_encode  # noqa: F401
