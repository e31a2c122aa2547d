from .grid import GridEncoder, grid_encode  # noqa: F401  (reference's __init__ is empty; `from gridencoder import GridEncoder` is what encoding.py does)
