#!/usr/bin/env python3
"""Generate tests/golden/reference_autocast.npz: the REFERENCE's own NeRFNetwork.forward / .density (network.py:252-311) run under
torch autocast with dtype float16 -- the arithmetic the reference renders and trains in by default (`-O` => --fp16, train.py:143-145;
`torch.cuda.amp.autocast(enabled=self.fp16)`, TrainerUtil.py:455,535,649,858).  Build container only; only arrays travel.

There is no CUDA device here, so the autocast region is `torch.autocast("cpu", dtype=torch.float16)`: the SAME Python graph, the same
casting machinery, and the same policy for every op of this graph except three, which the fixture records so that the test can tell them
apart (tests/test_golden_autocast.py holds the list and how the checker resolves each from torch's CUDA autocast lists):

    op (network.py)                      CUDA autocast                    CPU autocast (this fixture)
    torch.exp(h[..., 0])        :302     fp32 list: half -> f32 -> f32    not listed: half in, half out
    aud_ch_att.norm(dim=-1)     :308     fp32 list: f32 out               not listed: half in, half out
    torch.log(1 + torch.exp(u)) :278     exp / log in the fp32 list       not listed: half throughout
    nn.Linear                   :87      half list (both)                 half list
    cat                         :267,296 promote to widest (both)         promote to widest
    relu / sigmoid / mul / sub           not listed: run in the input type, ordinary type promotion (both)

So the fixture stores, besides the five outputs, the half OUTPUT OF EVERY nn.Linear (forward hooks) -- the values both policies agree
on and from which exp / norm / softplus start -- and the aten-level op trace (op, input dtypes -> output dtype) seen under the
autocast layer, which is what "per-layer dtypes" means here.  Two arrangements of the conditioning input:
  * `h`: enc_a is HALF, as `encode_audio` returns it inside the reference's autocast region (renderer.py:241: Conv1d / Linear are on the
    half list) -- enc_a * att is then a half product;
  * `f`: enc_a is float32 (a caller that feeds a precomputed feature): enc_a * att is an f32 product by type promotion, rounded to half
    only where sigma_net's first Linear casts its input.
Test mode (net.testing = True: constant uncertainty) and training mode (unc_net) are both recorded.

Network, tables, inputs: exactly those of make_golden.py (seed 0 / table seed 1234 / input seed 2); asserted equal to the committed
reference_python.npz, so the tests take weights and inputs from that fixture and this one holds outputs only.

Run:  python tests/golden/make_golden_autocast.py
"""
import os
import sys

import numpy as np
import torch
from torch.utils._python_dispatch import TorchDispatchMode

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG  # noqa: E402  (puts the reference and the checker on sys.path, installs the back-end adapters)


LAYER_ROWS = 256     # rows of every Linear output that are stored (the five outputs are stored for all 777 samples)


class OpTrace(TorchDispatchMode):
    """aten ops below the autocast layer with the dtypes they ran in"""

    def __init__(self):
        super().__init__()
        self.rows = []

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))

        def dts(x):
            if isinstance(x, torch.Tensor):
                return [str(x.dtype).replace("torch.", "")]
            if isinstance(x, (list, tuple)):
                return [d for y in x for d in dts(y)]
            return []
        name = func.__name__ if hasattr(func, "__name__") else str(func)
        if name.split(".")[0] in ("detach", "clone", "t", "view", "lift_fresh", "empty"):     # the hooks' copies and pure metadata ops
            return out
        self.rows.append("%s(%s)->%s" % (name, ",".join(d for a in args for d in dts(a)), ",".join(dts(out))))
        return out


def build_net():
    """the network of make_golden.main(): same seeds, same table values"""
    from nerf_triplane.network import NeRFNetwork  # reference
    torch.manual_seed(0)
    net = NeRFNetwork(MG.Opt())
    net.eval()
    trng = np.random.default_rng(1234)
    for n in ("xy", "yz", "xz"):
        enc = getattr(net, f"encoder_{n}")
        enc.embeddings.data.copy_(torch.from_numpy(trng.uniform(-1, 1, tuple(enc.embeddings.shape)).astype(np.float32)))
    return net


def main():
    MG.install_backends()
    net = build_net()
    G = np.load(os.path.join(HERE, "reference_python.npz"))
    sd = net.state_dict()
    for k in G.files:
        if k.startswith("sd/"):
            assert np.array_equal(sd[k[3:]].numpy(), G[k]), k          # same network as the f32 fixture
    tx, td = torch.from_numpy(G["net_xyz"]), torch.from_numpy(G["net_dirs"])
    enc_a, ind, eye = torch.from_numpy(G["net_enc_a"]), torch.from_numpy(G["net_ind"]), torch.from_numpy(G["net_eye"])

    layer_out = {}
    hooks = []
    for name, mod in net.named_modules():
        if isinstance(mod, torch.nn.Linear) and name.split(".")[0] in ("sigma_net", "color_net", "unc_net", "aud_ch_att_net", "eye_att_net"):
            hooks.append(mod.register_forward_hook(lambda m, i, o, name=name: layer_out.__setitem__(name, (i[0].detach().clone(), o.detach().clone()))))

    out = {}
    for tag, a in (("h", enc_a.half()), ("f", enc_a)):
        for mode in ("test", "train"):
            net.testing = mode == "test"
            layer_out.clear()
            tr = OpTrace()
            with torch.no_grad(), torch.autocast("cpu", dtype=torch.float16), tr:
                sig, rgb, aa, ae, unc = net(tx, td, a, ind, eye)
            p = f"{tag}_{mode}_"
            for name, (i, o) in layer_out.items():
                # ReLU is in place (network.py:89): the hook's clone was taken before it ran, so this is the Linear's own output
                # stored once: arrangements that reproduce `h_test`'s bits for a layer (all of train mode but unc_net; everything upstream
                # of sigma_net in `f`) keep a marker instead of a copy
                first = out.get("h_test_lin/" + name)
                out[p + "lin/" + name] = np.array("=h_test") if first is not None and p != "h_test_" and np.array_equal(first, o.numpy()[:LAYER_ROWS]) else o.numpy()[:LAYER_ROWS]
                out[p + "lin_in_dtype/" + name] = np.array(str(i.dtype).replace("torch.", ""))
            for n, t in (("sigma", sig), ("rgb", rgb), ("amb_aud", aa), ("amb_eye", ae), ("unc", unc)):
                out[p + n] = t.numpy() if n != "unc" or mode == "train" else t.numpy().reshape(t.shape[0], -1)[:, :1]
                out[p + n + "_dtype"] = np.array(str(t.dtype).replace("torch.", ""))
            out[p + "op_trace"] = np.array(tr.rows)
    net.testing = False
    # the density() entry point alone (renderer.py:744: the occupancy-grid update calls it under the same autocast region)
    with torch.no_grad(), torch.autocast("cpu", dtype=torch.float16):
        d = net.density(tx, enc_a.half(), eye)
    out["h_density_sigma"] = d["sigma"].numpy()
    out["h_density_geo"] = d["geo_feat"].numpy()[:LAYER_ROWS]
    for h in hooks:
        h.remove()
    path = os.path.join(HERE, "reference_autocast.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KB")
    for k in sorted(out):
        if k.endswith("_dtype") or "lin_in_dtype" in k:
            print(k, out[k])
    print("\n".join(out["h_train_op_trace"]))


if __name__ == "__main__":
    main()
