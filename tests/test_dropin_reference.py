"""Drop-in check against the reference's own, unmodified caller code.  Runs only where /root/reference exists (the
build container; the GPU box has no reference tree, and these tests need no GPU): with lzzx_nerf_amd/dropin in front
of sys.path the reference's nerf_triplane.network.NeRFNetwork must construct on top of this repo's encoder modules
and expose the same state_dict contract."""
import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"

pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "nerf_triplane")), reason="reference tree not present")

SCRIPT = textwrap.dedent("""
    import sys, types
    sys.path.insert(0, %(dropin)r); sys.path.insert(1, %(root)r); sys.path.insert(2, %(ref)r)
    for name in ("lpips", "trimesh", "mcubes", "cv2", "tensorboardX", "torch_ema", "imageio", "pydub", "numba"):
        try: __import__(name)
        except Exception: sys.modules[name] = types.ModuleType(name)   # absent third-party packages off the hot path
    import encoding, raymarching, gridencoder, shencoder, freqencoder
    assert encoding.__file__.startswith(%(dropin)r) and raymarching.__file__.startswith(%(dropin)r)
    from nerf_triplane.network import NeRFNetwork            # reference source, unmodified
    import lzzx_nerf_amd.gridencoder as G, lzzx_nerf_amd.shencoder as S, lzzx_nerf_amd.freqencoder as F
    class Opt:
        bound=1; min_near=0.05; density_thresh=10; density_thresh_torso=0.01; exp_eye=True; test_train=False
        smooth_lips=False; torso=%(torso)s; cuda_ray=True; ind_num=10; ind_dim=4; ind_dim_torso=8; train_camera=False
        emb=False; asr_model='deepspeech'; att=2; unc_loss=1; torso_shrink=0.8
    net = NeRFNetwork(Opt())
    assert type(net.encoder_xy) is G.GridEncoder and type(net.encoder_dir) is S.SHEncoder
    assert net.in_dim == 36 and net.in_dim_dir == 16
    sd = net.state_dict()
    assert tuple(sd['encoder_xy.embeddings'].shape) == (163584, 1) and str(sd['encoder_yz.offsets'].dtype) == 'torch.int32'
    assert tuple(sd['sigma_net.net.0.weight'].shape) == (64, 69) and tuple(sd['density_bitfield'].shape) == (262144,)
    if %(torso)s:
        assert type(net.torso_deform_encoder) is F.FreqEncoder and net.torso_deform_in_dim == 34
        assert type(net.torso_encoder) is G.GridEncoder and net.torso_encoder.gridtype == 'tiled' and net.torso_in_dim == 32
    # the fused head consumes this state_dict as is (construction + packing need a GPU, so only the key contract here)
    from lzzx_nerf_amd import head
    assert all(k in sd for k in head._W_KEYS)
    print('DROPIN_OK', len(sd))
""")


@pytest.mark.parametrize("torso", [False, True])
def test_reference_network_constructs_on_dropin(torso):
    code = SCRIPT % dict(dropin=os.path.join(ROOT, "lzzx_nerf_amd", "dropin"), root=ROOT, ref=REF, torso=torso)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "DROPIN_OK" in r.stdout, r.stderr[-3000:]
