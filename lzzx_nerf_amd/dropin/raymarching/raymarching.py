"""reference path raymarching/raymarching.py -> lzzx_nerf_amd.raymarching (17 operators)"""
from lzzx_nerf_amd.raymarching import (composite_rays, composite_rays_ambient, composite_rays_ambient_sigma,  # noqa: F401
                                       composite_rays_train, composite_rays_train_sigma, composite_rays_train_triplane,
                                       composite_rays_train_uncertainty, composite_rays_triplane, composite_rays_uncertainty,
                                       march_rays, march_rays_train, morton3D, morton3D_dilation, morton3D_invert,
                                       near_far_from_aabb, packbits, sph_from_ray)
