"""`get_encoder` -- the plugin boundary through which the reference's networks obtain their encoders
(/root/reference/encoding.py:6-37).  Same call signature, defaults and `(encoder, output_dim)` result; built as a small registry."""
from .freqencoder import FreqEncoder
from .gridencoder import GridEncoder
from .shencoder import SHEncoder


def _identity(cfg):
    return (lambda x, **kwargs: x), cfg["input_dim"]


def _frequency(cfg):
    return FreqEncoder(input_dim=cfg["input_dim"], degree=cfg["multires"])


def _sh(cfg):
    return SHEncoder(input_dim=cfg["input_dim"], degree=cfg["degree"])


def _grid(kind):
    def make(cfg):
        keys = ("input_dim", "num_levels", "level_dim", "base_resolution", "log2_hashmap_size", "desired_resolution", "align_corners")
        return GridEncoder(gridtype=kind, **{k: cfg[k] for k in keys})
    return make


_REGISTRY = {"frequency": _frequency, "spherical_harmonics": _sh, "hashgrid": _grid("hash"), "tiledgrid": _grid("tiled")}


def get_encoder(encoding, input_dim=3, multires=6, degree=4, num_levels=16, level_dim=2, base_resolution=16,
                log2_hashmap_size=19, desired_resolution=2048, align_corners=False, **kwargs):
    cfg = dict(input_dim=input_dim, multires=multires, degree=degree, num_levels=num_levels, level_dim=level_dim,
               base_resolution=base_resolution, log2_hashmap_size=log2_hashmap_size, desired_resolution=desired_resolution,
               align_corners=align_corners)
    if encoding == "None":
        return _identity(cfg)
    make = _REGISTRY.get(encoding)
    if make is None:   # the reference also lists an `ash` encoder whose package is not in its tree (encoding.py:31-33)
        raise NotImplementedError("Unknown encoding mode, choose from [None, frequency, spherical_harmonics, hashgrid, tiledgrid]")
    encoder = make(cfg)
    return encoder, encoder.output_dim
