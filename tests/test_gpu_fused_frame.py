"""The whole frame as one persistent kernel (csrc/lz_frame.hip, TriplaneRenderer(mode="fused")) against the multi-launch loop and the
CPU checker.
  cap = "reference" (the default): the fused frame reproduces the reference loop's frame-wide cap C_eff (renderer.py:503-548) and is
    bit-identical -- pixels, depth, sums, per-ray marched counts -- to the loop / checker under the REFERENCE schedule (budget_factor,
    n_step_cap) = (1, 8), whatever steps_per_pass is and also as tiles of one frame (test_fused_reference_cap_*).
  cap = "per_ray": the loop under the schedule n_step = S; against the loop / checker run with (S, S) everything is bit-identical."""
import numpy as np
import pytest
import torch

from conftest import ellipsoid_bitfield, synthetic_camera
from oracle.head import TriplaneSpec
from oracle.render import render_inference

pytestmark = pytest.mark.gpu
KEYS = ("image", "image_raw", "weights_sum", "depth", "amb_aud_sum", "amb_eye_sum", "uncertainty_sum", "nears", "fars")


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def setup(params, golden, H, W, scene, precision="f32"):
    from lzzx_nerf_amd.head import FusedTriplaneHead
    from lzzx_nerf_amd.utils import frame_rays
    head = FusedTriplaneHead({k: torch.from_numpy(v) for k, v in params.items()}, bound=1.0, precision=precision)
    bits = np.full(128 ** 3 // 8, 255, np.uint8) if scene == "ones" else ellipsoid_bitfield()[0]
    pose, intr = synthetic_camera(H, W)
    ro, rd = frame_rays(dev(pose), intr, H, W)
    cond = (dev(golden["net_enc_a"]), dev(golden["net_ind"]), dev(golden["net_eye"]))
    return head, bits, ro, rd, cond


def both(head, bits, ro, rd, cond, loop_schedule=(1, 1), steps_per_pass=1, cap="per_ray", **kw):
    from lzzx_nerf_amd.renderer import TriplaneRenderer
    fr = TriplaneRenderer(head, dev(bits), bound=1.0, mode="fused", cap=cap)
    fr.steps_per_pass = steps_per_pass
    fused = fr.render(ro, rd, *cond, count_samples=True, **kw)
    fused = {k: v.clone() for k, v in fused.items()}
    loop = TriplaneRenderer(head, dev(bits), bound=1.0, budget_factor=loop_schedule[0], n_step_cap=loop_schedule[1]).render(
        ro, rd, *cond, count_samples=True, **kw)
    return fused, loop


@pytest.mark.parametrize("scene,H,W,kw", [
    ("ellipsoid", 64, 64, dict(max_steps=64)),
    ("ones", 48, 40, dict(max_steps=96)),
    ("ones", 40, 40, dict(max_steps=16, dt_gamma=0.0)),                 # max_steps binds: every ray is cut at 16 samples
    ("ellipsoid", 64, 64, dict(max_steps=64, T_thresh=0.6)),            # T_thresh terminates rays early
])
def test_fused_frame_bit_exact_vs_loop_and_checker(params, golden, scene, H, W, kw):
    head, bits, ro, rd, cond = setup(params, golden, H, W, scene)
    fused, loop = both(head, bits, ro, rd, cond, **kw)
    for k in KEYS:
        assert torch.equal(fused[k], loop[k]), k
    assert torch.equal(fused["ray_counts"], loop["ray_counts"])
    assert int(fused["state"][5]) == int(loop["state"][5]) == int(fused["ray_counts"].sum())
    assert int(fused["state"][3]) == 1 and int(fused["state"][72]) >= int(fused["state"][5])
    st = {}
    ref = render_inference(TriplaneSpec(1.0), params, ro.cpu().numpy(), rd.cpu().numpy(), bits, golden["net_enc_a"], golden["net_ind"],
                           golden["net_eye"], stats=st, budget_factor=1, n_step_cap=1, **kw)
    assert np.array_equal(fused["image"].cpu().numpy(), ref["image"])
    assert np.array_equal(fused["depth"].cpu().numpy(), ref["depth"])
    assert np.array_equal(fused["ray_counts"].cpu().numpy().astype(np.int64), st["samples_per_ray"])
    if kw.get("T_thresh", 0) > 0.1:
        assert int((fused["weights_sum"] > 0.4).sum()) > 0      # the threshold did cut rays
    if kw.get("max_steps") == 16:
        assert int(fused["ray_counts"].max()) == 16


def test_fused_frame_equals_other_schedules_on_the_bench_frame(params, golden):
    """512 x 512, max_steps 192, all-ones occupancy: pixels, sums and (no ray is cut by T_thresh here) sample counts equal the
    headline schedule (4, 4) of the multi-launch loop"""
    head, bits, ro, rd, cond = setup(params, golden, 512, 512, "ones")
    fused, loop = both(head, bits, ro, rd, cond, loop_schedule=(4, 4), max_steps=192)
    for k in KEYS:
        assert torch.equal(fused[k], loop[k]), k
    assert torch.equal(fused["ray_counts"], loop["ray_counts"])
    assert int(fused["state"][5]) == 23928228
    rows = int(fused["state"][72])
    assert rows < 1.08 * 23928228          # refill keeps the slices full: < 8 % of the rows the head evaluates are empty slots


def test_fused_frame_f16_equals_loop_f16(params, golden):
    head, bits, ro, rd, cond = setup(params, golden, 96, 96, "ellipsoid", precision="f16")
    fused, loop = both(head, bits, ro, rd, cond, max_steps=64)
    for k in KEYS:
        assert torch.equal(fused[k], loop[k]), k
    assert torch.equal(fused["ray_counts"], loop["ray_counts"])


def test_fused_frame_edge_cases(params, golden):
    from lzzx_nerf_amd.renderer import TriplaneRenderer
    head, bits, ro, rd, cond = setup(params, golden, 32, 32, "ellipsoid")
    r = TriplaneRenderer(head, dev(bits), bound=1.0, mode="fused")
    # every ray misses the box: background everywhere, nothing queued
    up = torch.zeros_like(rd)
    up[:, 1] = 1.0
    up[:, 0] = 1e-3
    up[:, 2] = 1e-3
    out = r.render(ro + torch.tensor([0.0, 5.0, 0.0], device="cuda"), up, *cond, max_steps=32, count_samples=True)
    assert float(out["image"].min()) == 1.0 and int(out["state"][5]) == 0 and int(out["state"][1]) == 0
    assert int(out["ray_counts"].sum()) == 0 and float(out["weights_sum"].abs().max()) == 0.0
    # 5 rays, a per-ray background, RGB24 hand-off
    bg = torch.rand(5, 3, device="cuda")
    sel = torch.tensor([0, 500, 528, 700, 1023], device="cuda")
    o5 = r.render(ro[sel], rd[sel], *cond, max_steps=32, bg_color=bg, rgb24=True)
    loop = TriplaneRenderer(head, dev(bits), bound=1.0).render(ro[sel], rd[sel], *cond, max_steps=32, bg_color=bg, rgb24=True)
    assert torch.equal(o5["image"], loop["image"]) and torch.equal(o5["image_rgb24"], loop["image_rgb24"])
    # determinism across runs (the queue order inside a key bin is free; results are not)
    a = r.render(ro, rd, *cond, max_steps=32)["image"].clone()
    b = r.render(ro, rd, *cond, max_steps=32)["image"].clone()
    assert torch.equal(a, b)


@pytest.mark.parametrize("S", [2, 4, 8, 16])
@pytest.mark.parametrize("scene,kw", [("ellipsoid", dict(max_steps=64)), ("ones", dict(max_steps=50, T_thresh=0.5)), ("ones", dict(max_steps=19, dt_gamma=0.0))])
def test_fused_frame_multi_step_equals_schedule_s(params, golden, S, scene, kw):
    """steps_per_pass = S (what the host picks for small ray counts) is the loop under n_step = S: bit-identical to the multi-launch
    loop and to the checker run with (budget_factor, n_step_cap) = (S, S), counts of T_thresh-cut rays and the max_steps rounding included"""
    head, bits, ro, rd, cond = setup(params, golden, 40, 48, scene)
    fused, loop = both(head, bits, ro, rd, cond, loop_schedule=(S, S), steps_per_pass=S, **kw)
    for k in KEYS:
        assert torch.equal(fused[k], loop[k]), k
    assert torch.equal(fused["ray_counts"], loop["ray_counts"])
    assert int(fused["state"][5]) == int(fused["ray_counts"].sum()) == int(loop["state"][5])
    st = {}
    ref = render_inference(TriplaneSpec(1.0), params, ro.cpu().numpy(), rd.cpu().numpy(), bits, golden["net_enc_a"], golden["net_ind"],
                           golden["net_eye"], stats=st, budget_factor=S, n_step_cap=S, **kw)
    assert np.array_equal(fused["image"].cpu().numpy(), ref["image"])
    assert np.array_equal(fused["ray_counts"].cpu().numpy().astype(np.int64), st["samples_per_ray"])


def test_fused_frame_auto_steps_per_pass_keeps_pixels(params, golden):
    """a 1/8 tile of the bench frame (32 768 rays: the host picks S = 4) renders the same pixels as the frame rendered whole (S = 1)"""
    from lzzx_nerf_amd import dist as D
    from lzzx_nerf_amd.renderer import TriplaneRenderer
    head, bits, ro, rd, cond = setup(params, golden, 512, 512, "ones")
    r = TriplaneRenderer(head, dev(bits), bound=1.0, mode="fused", cap="per_ray")
    full = r.render(ro, rd, *cond, max_steps=192)["image"].clone()
    px = D.tile_pixels(512, 512, 3, 8, "interleaved", "cuda")
    tile = r.render(ro[px].contiguous(), rd[px].contiguous(), *cond, max_steps=192)
    assert torch.equal(tile["image"], full[px])
    assert int(tile["state"][72]) < 1.08 * int(tile["state"][5])      # rows wasted behind the last sample of a ray: < 8 %


def test_cfg5_1024_ellipsoid_f16_fused_equals_loop(params, golden):
    """BASELINE cfg5 at full size: 1024 x 1024 rays, ellipsoid occupancy (2.9 % of the cells: empty-space skipping + compaction),
    f16 MLP on the matrix cores -- the fused frame equals the multi-launch loop under the reference's schedule, pixel for pixel and
    count for count (no ray of this scene is cut by T_thresh), and most rays never enter the queue"""
    head, bits, ro, rd, cond = setup(params, golden, 1024, 1024, "ellipsoid", precision="f16")
    fused, loop = both(head, bits, ro, rd, cond, loop_schedule=(1, 8), max_steps=192)
    for k in KEYS:
        assert torch.equal(fused[k], loop[k]), k
    assert torch.equal(fused["ray_counts"], loop["ray_counts"])
    n_hit = int(fused["state"][1])
    assert 0.05 * 1024 * 1024 < n_hit < 0.5 * 1024 * 1024           # only the rays that meet the ellipsoid are queued
    assert int((fused["ray_counts"] > 0).sum()) == n_hit


@pytest.mark.parametrize("S", [1, 4])
def test_fused_frame_two_cascades(golden, S):
    """bound 2 (cascade 2: mip level from position / step size, dt_max = 2 sqrt(3) 2 / 128, 128^3 cells per cascade): the fused frame,
    the multi-launch loop and the checker agree bit for bit on a scene with cells in both cascades"""
    from conftest import make_params
    from lzzx_nerf_amd.head import FusedTriplaneHead
    from lzzx_nerf_amd.utils import frame_rays
    from oracle import oracle as O
    bound = 2.0
    spec = TriplaneSpec(bound)
    P = make_params(golden)
    rng = np.random.default_rng(21)
    for n in ("xy", "yz", "xz"):
        P[f"encoder_{n}.embeddings"] = rng.uniform(-1, 1, (spec.n_params, 1)).astype(np.float32)
        P[f"encoder_{n}.offsets"] = spec.offsets.astype(np.int32)
    # occupancy: a ball of radius 0.6 (cascade 0 cells) and a shell 1.2 < r < 1.6 (cascade 1 cells), Morton-ordered per cascade
    c = np.arange(128, dtype=np.int32)
    X, Y, Z = np.meshgrid(c, c, c, indexing="ij")
    coords = np.stack([X.ravel(), Y.ravel(), Z.ravel()], 1)
    idx = O.morton3D(coords)
    grid = np.zeros((2, 128 ** 3), np.float32)
    for cas, half in ((0, 1.0), (1, 2.0)):
        xyz = (coords.astype(np.float32) + 0.5) / 128 * 2 * half - half
        r = np.sqrt((xyz ** 2).sum(1))
        grid[cas, idx] = ((r < 0.6) if cas == 0 else ((r > 1.2) & (r < 1.6))).astype(np.float32)
    bits = O.packbits(grid, 0.5)
    H = W = 56
    pose, intr = synthetic_camera(H, W)
    pose = pose.copy()
    pose[2, 3] = -5.0
    intr = [intr[0] * 0.5, intr[1] * 0.5, intr[2], intr[3]]             # wider field of view: rays cross both cascades
    head = FusedTriplaneHead({k: torch.from_numpy(v) for k, v in P.items()}, bound=bound)
    ro, rd = frame_rays(dev(pose), intr, H, W)
    cond = (dev(golden["net_enc_a"]), dev(golden["net_ind"]), dev(golden["net_eye"]))
    from lzzx_nerf_amd.renderer import TriplaneRenderer
    fr = TriplaneRenderer(head, dev(bits), bound=bound, mode="fused", cap="per_ray")
    fr.steps_per_pass = S
    fused = {k: v.clone() for k, v in fr.render(ro, rd, *cond, max_steps=96, count_samples=True).items()}
    loop = TriplaneRenderer(head, dev(bits), bound=bound, budget_factor=S, n_step_cap=S).render(ro, rd, *cond, max_steps=96, count_samples=True)
    for k in KEYS:
        assert torch.equal(fused[k], loop[k]), k
    assert torch.equal(fused["ray_counts"], loop["ray_counts"])
    st = {}
    ref = render_inference(spec, P, ro.cpu().numpy(), rd.cpu().numpy(), bits, golden["net_enc_a"], golden["net_ind"], golden["net_eye"],
                           cascade=2, max_steps=96, stats=st, budget_factor=S, n_step_cap=S)
    assert np.array_equal(fused["image"].cpu().numpy(), ref["image"])
    assert np.array_equal(fused["ray_counts"].cpu().numpy().astype(np.int64), st["samples_per_ray"])
    cnt = st["samples_per_ray"]
    assert cnt.max() > 20 and (cnt == 0).any()        # rays through the shell and the ball, and rays that miss everything


@pytest.mark.parametrize("scene,H,W,kw", [("ellipsoid", 64, 64, dict(max_steps=64)), ("ones", 48, 40, dict(max_steps=96, T_thresh=0.3))])
def test_fold_geo_frame_and_head(params, golden, scene, H, W, kw):
    """fold_geo (precision 2): geo = Wg s2 folded into color_net.0 at pack time.  sigma -- and with it every ray's sample count, the weights,
    depth and the ambient sums -- is the exact path's bit for bit; rgb and the image move by the reassociation only (north_star's bar
    is 1e-4 abs).  Fused frame and loop mode, and the bare head on random points."""
    from lzzx_nerf_amd.head import FusedTriplaneHead
    from lzzx_nerf_amd.renderer import TriplaneRenderer
    head, bits, ro, rd, cond = setup(params, golden, H, W, scene)
    folded = FusedTriplaneHead({k: torch.from_numpy(v) for k, v in params.items()}, bound=1.0, fold_geo=True)
    for mode in ("fused", "loop"):
        a = TriplaneRenderer(head, dev(bits), bound=1.0, mode=mode).render(ro, rd, *cond, count_samples=True, **kw)
        a = {k: v.clone() for k, v in a.items()}
        b = TriplaneRenderer(folded, dev(bits), bound=1.0, mode=mode).render(ro, rd, *cond, count_samples=True, **kw)
        for k in ("weights_sum", "depth", "amb_aud_sum", "amb_eye_sum", "uncertainty_sum", "nears", "fars", "ray_counts"):
            assert torch.equal(a[k], b[k]), (mode, k)
        assert float((a["image"] - b["image"]).abs().max()) < 2e-6, mode
        assert not torch.equal(a["image_raw"], b["image_raw"]) or scene == "never"   # it IS a different association
    g = torch.Generator(device="cuda").manual_seed(5)
    xyz = torch.rand(5000, 3, device="cuda", generator=g) * 2 - 1
    d = torch.nn.functional.normalize(torch.randn(5000, 3, device="cuda", generator=g), dim=-1)
    o0, o1 = head.forward(xyz, d, *cond), folded.forward(xyz, d, *cond)
    assert torch.equal(o0[0], o1[0]) and torch.equal(o0[2], o1[2]) and torch.equal(o0[3], o1[3])
    assert float((o0[1] - o1[1]).abs().max()) < 2e-6
    with pytest.raises(RuntimeError):
        folded.forward(xyz, d, *cond, testing=False)


@pytest.mark.parametrize("scene,S,kw", [("ellipsoid", 1, dict(max_steps=64)), ("ones", 1, dict(max_steps=48)), ("ellipsoid", 4, dict(max_steps=64))])
def test_perturb_on_the_fast_paths_equals_the_reference_loop_with_the_same_noise(params, golden, scene, S, kw):
    """perturb (renderer.py:344,521: march_rays gets it on the FIRST iteration only; raymarching.cu:873: every ray starts at
    near + clamp(near dt_gamma, dt_min, dt_max) * noise): loop mode and the fused frame, with recorded draws, against the checker's loop
    handed the same draws -- pixels, depth, sums and per-ray sample counts bit for bit; and it does move the samples."""
    from lzzx_nerf_amd.renderer import TriplaneRenderer
    head, bits, ro, rd, cond = setup(params, golden, 56, 48, scene)
    N = ro.shape[0]
    noise = np.random.default_rng(77).uniform(0, 1, N).astype(np.float32)
    st = {}
    ref = render_inference(TriplaneSpec(1.0), params, ro.cpu().numpy(), rd.cpu().numpy(), bits, golden["net_enc_a"], golden["net_ind"],
                           golden["net_eye"], stats=st, budget_factor=S, n_step_cap=S, noises=noise, **kw)
    plain = render_inference(TriplaneSpec(1.0), params, ro.cpu().numpy(), rd.cpu().numpy(), bits, golden["net_enc_a"], golden["net_ind"],
                             golden["net_eye"], budget_factor=S, n_step_cap=S, **kw)
    assert not np.array_equal(ref["image"], plain["image"])
    fr = TriplaneRenderer(head, dev(bits), bound=1.0, mode="fused", cap="per_ray")
    fr.steps_per_pass = S
    fused = {k: v.clone() for k, v in fr.render(ro, rd, *cond, count_samples=True, noises=dev(noise), **kw).items()}
    loop = TriplaneRenderer(head, dev(bits), bound=1.0, budget_factor=S, n_step_cap=S).render(ro, rd, *cond, count_samples=True, noises=dev(noise), **kw)
    for name, out in (("fused", fused), ("loop", loop)):
        assert np.array_equal(out["image"].cpu().numpy(), ref["image"]), name
        assert np.array_equal(out["depth"].cpu().numpy(), ref["depth"]), name
        assert np.array_equal(out["weights_sum"].cpu().numpy(), ref["weights_sum"]), name
        assert np.array_equal(out["ray_counts"].cpu().numpy().astype(np.int64), st["samples_per_ray"]), name
    # the reference's schedule (1, 8): perturbed starts shift where a ray ends, so only rays that end before the cap keep their pixels --
    # all of them here (max_steps exceeds the longest chord)
    ref18 = TriplaneRenderer(head, dev(bits), bound=1.0).render(ro, rd, *cond, noises=dev(noise), **kw)
    assert torch.equal(ref18["image"], fused["image"])
    # perturb=True draws its own noise (torch.rand like raymarching.py:298): another image, same background for the rays that miss
    own = fr.render(ro, rd, *cond, perturb=True, **kw)["image"]
    assert not torch.equal(own, fused["image"])
    with pytest.raises(RuntimeError, match="one value per ray"):
        fr.render(ro, rd, *cond, noises=dev(noise[:5]), **kw)


def _bitfield_from_cells(cells, grid_size=128, cascade=1):
    """cells: list of (level, x, y, z) -> bitfield through the checker's morton3D (bit index = level * H^3 + morton, raymarching.cu:267-300)"""
    from oracle import oracle as O
    bits = np.zeros(cascade * grid_size ** 3 // 8, np.uint8)
    c = np.asarray(cells, np.int64).reshape(-1, 4)
    idx = c[:, 0] * grid_size ** 3 + O.morton3D(c[:, 1:].astype(np.int32)).astype(np.int64)
    np.bitwise_or.at(bits, idx // 8, (1 << (idx % 8)).astype(np.uint8))
    return bits


def _blob(level, centre, radius, grid_size=128):
    g = np.arange(grid_size)
    x, y, z = np.meshgrid(g, g, g, indexing="ij")
    m = (x - centre[0]) ** 2 + (y - centre[1]) ** 2 + (z - centre[2]) ** 2 <= radius ** 2
    return [(level, int(a), int(b), int(c)) for a, b, c in zip(x[m], y[m], z[m])]


@pytest.mark.parametrize("case", ["ellipsoid", "two_cascades", "rim", "empty", "ones"])
def test_occupied_bounds_match_numpy(case):
    """lz_occupied_bounds: the world-space box of the set bits of a density bitfield, per level ((n .. n + 1) / H * 2 - 1) * min(2^level,
    bound) (raymarching.cu:409-417), dilated by max(margin cells of the level, 4 dt_max); open towards the rim of the outermost level"""
    from lzzx_nerf_amd._util import call, ptr, stream
    H, margin = 128, 8
    if case == "ellipsoid":
        C, bound, bits = 1, 1.0, ellipsoid_bitfield()[0]
    elif case == "two_cascades":
        C, bound = 2, 2.0
        cells = _blob(0, (40, 64, 90), 5) + _blob(1, (70, 60, 64), 3)
        bits = _bitfield_from_cells(cells, H, C)
    elif case == "rim":
        C, bound = 1, 1.0
        cells = _blob(0, (3, 64, 120), 3)
        bits = _bitfield_from_cells(cells, H, C)
    elif case == "empty":
        C, bound, bits = 1, 1.0, np.zeros(H ** 3 // 8, np.uint8)
    else:
        C, bound, bits = 1, 1.0, np.full(H ** 3 // 8, 255, np.uint8)
    box, ws = torch.empty(6, device="cuda"), torch.empty(48, dtype=torch.int32, device="cuda")
    call("lz_occupied_bounds", ptr(dev(bits)), C, H, bound, margin, ptr(ws), ptr(box), stream())
    got = box.cpu().numpy()
    from oracle import oracle as O
    FMAX = np.finfo(np.float32).max
    idx = np.nonzero(np.unpackbits(bits, bitorder="little"))[0]
    if len(idx) == 0:
        assert np.all(got == FMAX)
        return
    lo, hi = np.full(3, np.inf), np.full(3, -np.inf)
    dt_max = 2 * np.sqrt(3) * 2 ** (C - 1) / H
    for l in range(C):
        sel = idx[(idx // H ** 3) == l] % H ** 3
        if len(sel) == 0:
            continue
        xyz = O.morton3D_invert(sel.astype(np.int32)).astype(np.int64)
        mb = min(2.0 ** l, bound)
        pad = max(margin * 2 * mb / H, 4 * dt_max)
        w0 = (xyz.min(0) / H * 2 - 1) * mb - pad
        w1 = ((xyz.max(0) + 1) / H * 2 - 1) * mb + pad
        if l == C - 1:
            w0 = np.where(w0 <= -mb, -np.inf, w0)
            w1 = np.where(w1 >= mb, np.inf, w1)
        lo, hi = np.minimum(lo, w0), np.maximum(hi, w1)
    want = np.concatenate([lo, hi])
    for g, w in zip(got, want):
        if np.isinf(w):
            assert abs(g) == FMAX and np.sign(g) == np.sign(w)
        else:
            assert abs(g - w) < 1e-5, (got, want)
    if case == "rim":
        assert got[0] == -FMAX and got[5] == FMAX and abs(got[1]) < 1
    if case == "ones":
        assert np.all(np.abs(got) == FMAX)


@pytest.mark.parametrize("scene,precision,kw", [
    ("ellipsoid", "f32", dict(max_steps=192)),
    ("ellipsoid", "f16", dict(max_steps=192)),
    ("blobs", "f32", dict(max_steps=128)),
    ("blobs", "f32", dict(max_steps=1024, perturb=True)),     # the reference's inference cap: dt_min = 0.22 cells, ~600 steps across the box
    ("rim", "f32", dict(max_steps=96)),
    ("empty", "f32", dict(max_steps=64)),
])
def test_march_confined_to_the_occupied_bounds_changes_no_sample(params, golden, scene, precision, kw):
    """TriplaneRenderer(mode="fused").clip_to_occupancy (lz_frame_fused.occupied_aabb): the stretch of a ray in front of the occupied
    cells is walked with the march's own step and no cell test, the ray ends where it leaves their bounds.  Every output, the per-ray
    sample counts and the sample total equal the unclipped kernel's and the multi-launch loop's (whose march tests every cell), bit for bit."""
    from lzzx_nerf_amd.renderer import TriplaneRenderer
    H = W = 160
    head, bits, ro, rd, cond = setup(params, golden, H, W, "ellipsoid", precision=precision)
    if scene == "blobs":        # two separate objects, one off-centre near a face of the box: empty space in front, between and behind
        bits = _bitfield_from_cells(_blob(0, (40, 52, 30), 9) + _blob(0, (84, 70, 100), 13) + _blob(0, (64, 64, 64), 2))
    elif scene == "rim":        # occupied cells on the rim of the grid: the box is open on those sides
        bits = _bitfield_from_cells(_blob(0, (60, 64, 2), 6) + _blob(0, (70, 60, 126), 5))
    elif scene == "empty":
        bits = np.zeros(128 ** 3 // 8, np.uint8)
    noises = None
    if kw.pop("perturb", False):
        noises = torch.rand(ro.shape[0], device="cuda", generator=torch.Generator(device="cuda").manual_seed(5))
    outs = []
    for clip in (True, False):
        r = TriplaneRenderer(head, dev(bits), bound=1.0, mode="fused", cap="per_ray")
        r.clip_to_occupancy = clip
        r.steps_per_pass = 1
        o = r.render(ro, rd, *cond, count_samples=True, noises=noises, **kw)
        outs.append({k: v.clone() for k, v in o.items()})
    loop = TriplaneRenderer(head, dev(bits), bound=1.0, budget_factor=1, n_step_cap=1).render(ro, rd, *cond, count_samples=True, noises=noises, **kw)
    for other in (outs[1], loop):
        for k in KEYS + ("ray_counts",):
            assert torch.equal(outs[0][k], other[k]), k
        assert int(outs[0]["state"][5]) == int(other["state"][5])
    if scene != "empty":
        assert int(outs[0]["state"][5]) > 1000
    else:
        assert int(outs[0]["state"][5]) == 0 and float(outs[0]["image"].min()) == 1.0


@pytest.mark.parametrize("precision", ["f32", "f16"])
def test_fused_frame_two_cascades_with_occupancy_bounds(params, golden, precision):
    """bound = 2 (two cascade levels, renderer.py:93; tables sized for desired_resolution 512 * bound, network.py:131): the fused frame with
    the march confined to the occupied bounds of BOTH levels (a blob in the inner level, a larger one in the outer level only) against
    the unclipped kernel and the multi-launch loop -- mip_from_pos / mip_from_dt, per-level dilation and the union of the level boxes"""
    from lzzx_nerf_amd.gridencoder import GridEncoder
    from lzzx_nerf_amd.head import FusedTriplaneHead
    from lzzx_nerf_amd.renderer import TriplaneRenderer
    from lzzx_nerf_amd.utils import frame_rays
    bound = 2.0
    enc = GridEncoder(input_dim=2, num_levels=12, level_dim=1, base_resolution=64, log2_hashmap_size=14, desired_resolution=512 * bound)
    g = torch.Generator().manual_seed(21)
    sd = {k: torch.from_numpy(v) for k, v in params.items()}
    for n in ("xy", "yz", "xz"):
        sd[f"encoder_{n}.embeddings"] = torch.rand(enc.embeddings.shape, generator=g) * 2 - 1
        sd[f"encoder_{n}.offsets"] = enc.offsets.clone()
    head = FusedTriplaneHead(sd, bound=bound, precision=precision)
    H = W = 96
    pose, intr = synthetic_camera(H, W)
    ro, rd = frame_rays(dev(pose), intr, H, W)
    cond = (dev(golden["net_enc_a"]), dev(golden["net_ind"]), dev(golden["net_eye"]))
    # level 0 spans [-1, 1]^3, level 1 [-2, 2]^3: a blob around the origin in level 0 and a shell part only level 1 sees
    cells = _blob(0, (64, 64, 64), 14) + _blob(1, (64, 64, 64), 9) + _blob(1, (80, 64, 100), 6)
    bits = _bitfield_from_cells(cells, 128, 2)
    aabb = torch.tensor([-bound, -bound / 2, -bound, bound, bound / 2, bound], device="cuda")
    outs = []
    for clip in (True, False):
        r = TriplaneRenderer(head, dev(bits), bound=bound, cascade=2, aabb=aabb, mode="fused", cap="per_ray")
        r.clip_to_occupancy = clip
        r.steps_per_pass = 1
        outs.append({k: v.clone() for k, v in r.render(ro, rd, *cond, count_samples=True, max_steps=256).items()})
    loop = TriplaneRenderer(head, dev(bits), bound=bound, cascade=2, aabb=aabb, budget_factor=1, n_step_cap=1).render(ro, rd, *cond, count_samples=True, max_steps=256)
    for other in (outs[1], loop):
        for k in KEYS + ("ray_counts",):
            assert torch.equal(outs[0][k], other[k]), k
        assert int(outs[0]["state"][5]) == int(other["state"][5])
    assert int(outs[0]["state"][5]) > 5000


def test_occupied_bounds_follow_an_in_place_update_of_the_bitfield(params, golden):
    """The renderer caches the bounds of the occupied cells per (bitfield tensor, version).  raymarching.packbits into a supplied bitfield and
    occupancy.update_density_grid write through raw pointers and bump the tensor's version, so the next frame rescans: rendering after
    such an update equals a fresh renderer on the new bitfield (a stale box would cut the object that moved out of it)."""
    from lzzx_nerf_amd import raymarching as R
    from lzzx_nerf_amd.renderer import TriplaneRenderer
    from oracle import oracle as O
    head, bits0, ro, rd, cond = setup(params, golden, 96, 96, "ellipsoid")
    bitfield = dev(bits0).clone()
    r = TriplaneRenderer(head, bitfield, bound=1.0, mode="fused")
    a = {k: v.clone() for k, v in r.render(ro, rd, *cond, max_steps=128, count_samples=True).items()}
    box0 = r.occupied_bounds().clone()
    # a different scene written into the SAME tensor: a blob well outside the first box
    cells = np.asarray(_blob(0, (30, 50, 20), 8), np.int64)
    grid = np.zeros((1, 128 ** 3), np.float32)
    grid[0, O.morton3D(cells[:, 1:].astype(np.int32))] = 1.0
    R.packbits(dev(grid), 0.5, bitfield)
    b = {k: v.clone() for k, v in r.render(ro, rd, *cond, max_steps=128, count_samples=True).items()}
    assert not torch.equal(box0, r.occupied_bounds())
    fresh = TriplaneRenderer(head, bitfield.clone(), bound=1.0, mode="fused").render(ro, rd, *cond, max_steps=128, count_samples=True)
    for k in KEYS + ("ray_counts",):
        assert torch.equal(b[k], fresh[k]), k
    assert int(b["state"][5]) > 500 and not torch.equal(a["image"], b["image"])


# ---- cap = "reference": the fused frame under the reference's own schedule, also where max_steps binds ---------------------------------
def _scene_bits(scene):
    if scene == "ones":
        return np.full(128 ** 3 // 8, 255, np.uint8)
    if scene == "blobs":
        return _bitfield_from_cells(_blob(0, (40, 52, 30), 9) + _blob(0, (84, 70, 100), 13) + _blob(0, (64, 64, 64), 2))
    return ellipsoid_bitfield()[0]


@pytest.mark.parametrize("S", [1, 0, 4])
@pytest.mark.parametrize("scene,H,W,kw", [
    ("ellipsoid", 64, 64, dict(max_steps=16)),                      # the reference's deployed cap (HubertInferenceMQ.py:69, train.py:35)
    ("ellipsoid", 64, 64, dict(max_steps=32)),
    ("blobs", 72, 56, dict(max_steps=16)),
    ("blobs", 72, 56, dict(max_steps=32, T_thresh=0.5)),            # T_thresh cuts rays inside multi-step chunks: marched counts
    ("ones", 40, 40, dict(max_steps=16, dt_gamma=0.0)),             # every ray that enters the box reaches the cap
    ("ellipsoid", 64, 64, dict(max_steps=64, T_thresh=0.6)),        # the cap does not bind
])
def test_fused_reference_cap_equals_the_reference_schedule(params, golden, scene, H, W, kw, S):
    """mode="fused", cap="reference" against the multi-launch loop under the reference's schedule (1, 8) and the checker's
    render_inference(budget_factor=1, n_step_cap=8) = run_cuda_for_inference (renderer.py:503-548): image, depth, every sum and the
    per-ray marched counts bit for bit, for steps_per_pass 1, auto and 4; C_eff and the iteration count the device replayed equal the
    checker's schedule."""
    from lzzx_nerf_amd.renderer import TriplaneRenderer
    head, _, ro, rd, cond = setup(params, golden, H, W, "ones")
    bits = _scene_bits(scene)
    fr = TriplaneRenderer(head, dev(bits), bound=1.0, mode="fused", cap="reference")
    fr.steps_per_pass = S
    fused = {k: v.clone() for k, v in fr.render(ro, rd, *cond, count_samples=True, **kw).items()}
    loop = TriplaneRenderer(head, dev(bits), bound=1.0, budget_factor=1, n_step_cap=8).render(ro, rd, *cond, count_samples=True, **kw)
    for k in KEYS:
        assert torch.equal(fused[k], loop[k]), k
    assert torch.equal(fused["ray_counts"], loop["ray_counts"])
    assert int(fused["state"][5]) == int(loop["state"][5]) == int(fused["ray_counts"].sum())
    st = {}
    ref = render_inference(TriplaneSpec(1.0), params, ro.cpu().numpy(), rd.cpu().numpy(), bits, golden["net_enc_a"], golden["net_ind"],
                           golden["net_eye"], stats=st, budget_factor=1, n_step_cap=8, **kw)
    for k in ("image", "depth", "weights_sum", "amb_aud_sum", "amb_eye_sum", "uncertainty_sum"):
        assert np.array_equal(fused[k].cpu().numpy(), ref[k]), k
    assert np.array_equal(fused["ray_counts"].cpu().numpy().astype(np.int64), st["samples_per_ray"])
    sched = st["schedule"]
    c_eff = sum(n for _, n in sched)
    assert int(fused["state"][11]) == len(sched) and int(fused["state"][10]) == c_eff
    if kw["max_steps"] == 16 and scene != "blobs":
        assert int(fused["ray_counts"].max()) == c_eff                        # rays stand at the cap and receive C_eff samples
    if kw["max_steps"] == 16 and scene == "ellipsoid":
        assert c_eff > 16                                                     # ... past max_steps
        per_ray = TriplaneRenderer(head, dev(bits), bound=1.0, mode="fused", cap="per_ray")
        per_ray.steps_per_pass = 1
        other = per_ray.render(ro, rd, *cond, **kw)["image"]
        assert int(((other - fused["image"]).abs().max(1).values > 1e-4).sum()) > 50   # what the per-ray cap gets wrong


def test_fused_reference_cap_f16_equals_loop_f16(params, golden):
    from lzzx_nerf_amd.renderer import TriplaneRenderer
    head, _, ro, rd, cond = setup(params, golden, 96, 96, "ones", precision="f16")
    bits = _scene_bits("ellipsoid")
    for ms in (16, 24):
        fused = {k: v.clone() for k, v in TriplaneRenderer(head, dev(bits), bound=1.0, mode="fused").render(ro, rd, *cond, count_samples=True, max_steps=ms).items()}
        loop = TriplaneRenderer(head, dev(bits), bound=1.0).render(ro, rd, *cond, count_samples=True, max_steps=ms)
        for k in KEYS + ("ray_counts",):
            assert torch.equal(fused[k], loop[k]), (ms, k)
        assert int(fused["ray_counts"].max()) > ms


@pytest.mark.parametrize("tiles", ["interleaved", "contiguous"])
@pytest.mark.parametrize("max_steps", [16, 24])
def test_fused_reference_cap_tiles_equal_the_unsharded_reference_frame(params, golden, tiles, max_steps):
    """4 tiles of one frame, each rendered by its own fused renderer with its own steps_per_pass (auto: tiles pick a larger S than the
    frame would), the histograms of the four first phases summed as ShardedFrame's all-reduce does: the assembled frame equals the
    unsharded loop under the reference schedule -- pixels and counts -- although n_alive / N of renderer.py:513 are frame-wide."""
    from lzzx_nerf_amd import dist as D
    from lzzx_nerf_amd.renderer import TriplaneRenderer
    H = W = 64
    head, _, ro, rd, cond = setup(params, golden, H, W, "ones")
    bits = dev(_scene_bits("ellipsoid"))
    ref = TriplaneRenderer(head, bits, bound=1.0).render(ro, rd, *cond, count_samples=True, max_steps=max_steps)
    ref = {k: v.clone() for k, v in ref.items()}
    world = 4
    rs, ctxs, pxs = [], [], []
    for g in range(world):
        sf = D.ShardedFrame(H, W, g, world, tiles, device="cuda")
        sf.gatherer = None
        r = sf.configure(TriplaneRenderer(head, bits, bound=1.0, mode="fused"))
        assert r.cap == "reference" and r.frame_rays_total == H * W
        px = sf.pixels
        ctxs.append(r.fused_begin(ro[px].contiguous(), rd[px].contiguous(), *cond, max_steps=max_steps, count_samples=True))
        rs.append(r)
        pxs.append(px)
    total = torch.stack([c["hist"] for c in ctxs]).sum(0)
    assert int(total.sum()) == H * W
    imgs, cnts, deps = [], [], []
    for r, c in zip(rs, ctxs):
        c["hist"].copy_(total)                       # what dist.all_reduce leaves on every rank
        o = r.fused_finish(c)
        imgs.append(o["image"].clone()); cnts.append(o["ray_counts"].clone()); deps.append(o["depth"].clone())
    assert torch.equal(D.assemble_frame(torch.cat(imgs), H, W, world, tiles), ref["image"])
    assert torch.equal(D.assemble_frame(torch.cat(deps)[:, None], H, W, world, tiles)[:, 0], ref["depth"])
    assert torch.equal(D.assemble_frame(torch.cat(cnts)[:, None], H, W, world, tiles)[:, 0], ref["ray_counts"])
    assert int(ref["ray_counts"].max()) > max_steps


@pytest.mark.parametrize("tag", ["ms16", "ms32", "ms16_T", "ms32_perturb", "ms24_dg0"])
def test_hip_renderers_against_the_reference_loop_fixture(params, tag):
    """tests/golden/reference_loops.npz = the reference's OWN run_cuda_for_inference (renderer.py:406-570) run unmodified on the checker's
    kernels (make_golden_loops.py).  Loop mode under the reference schedule and the fused frame with the reference cap against it:
    iteration count, C_eff and per-ray marched counts exactly, image / depth / sums within north_star's 1e-4 (the fixture's MLP is
    torch's CPU GEMM, the kernels' the order-pinned MFMA chain)."""
    import os
    from lzzx_nerf_amd.head import FusedTriplaneHead
    from lzzx_nerf_amd.renderer import TriplaneRenderer
    loops = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_loops.npz"), allow_pickle=False)
    ms, dg, T, pert = loops[f"{tag}/kw"]
    kw = dict(max_steps=int(ms), dt_gamma=float(dg), T_thresh=float(T))
    pre = f"{tag}/inference/"
    head = FusedTriplaneHead({k: torch.from_numpy(v) for k, v in params.items()}, bound=1.0)
    bits = dev(ellipsoid_bitfield()[0])
    ro, rd = dev(loops["rays_o"]), dev(loops["rays_d"])
    cond = (dev(loops["enc_a"]), dev(loops["ind_code"]), dev(loops["eye"]))
    noises = dev(loops[pre + "noises"]) if pert else None
    sched = loops[pre + "schedule"]
    for mode in ("loop", "fused"):
        r = TriplaneRenderer(head, bits, bound=1.0, mode=mode)
        o = r.render(ro, rd, *cond, count_samples=True, noises=noises, **kw)
        assert np.array_equal(o["ray_counts"].cpu().numpy().astype(np.int64), loops[pre + "counts"]), mode
        if mode == "loop":
            assert int(o["state"][6]) == len(sched)
        else:
            assert int(o["state"][11]) == len(sched) and int(o["state"][10]) == int(sched[:, 1].sum())
        for k in ("image", "image_raw", "weights_sum", "depth", "amb_aud_sum", "amb_eye_sum", "uncertainty_sum"):
            assert float((o[k].cpu() - torch.from_numpy(loops[pre + k])).abs().max()) <= 1e-4, (mode, k)


def test_occupied_bounds_cache_is_keyed_on_the_tensor_object(params, golden):
    """(a) a NEW bitfield tensor assigned to the renderer -- which the caching allocator may well place at the freed address of
    the old one, with the same version counter -- must be rescanned (a stale box silently drops geometry); (b) a bitfield created under
    torch.inference_mode() tracks no version: it renders (rescanned per frame) instead of raising."""
    from lzzx_nerf_amd.renderer import TriplaneRenderer
    head, bits0, ro, rd, cond = setup(params, golden, 64, 64, "ellipsoid")
    other = _bitfield_from_cells(_blob(0, (30, 50, 20), 8))
    r = TriplaneRenderer(head, dev(bits0).clone(), bound=1.0, mode="fused")
    r.render(ro, rd, *cond, max_steps=64)
    addr = r.bitfield.data_ptr()
    r.bitfield = None
    r._occ = (None,) + tuple(r._occ[1:])     # drop the cache's reference too, as a long-lived server eventually would
    torch.cuda.synchronize()
    nb = dev(other).clone()                   # same size: the allocator's favourite block is the one just freed
    r.bitfield = nb
    b = {k: v.clone() for k, v in r.render(ro, rd, *cond, max_steps=64, count_samples=True).items()}
    fresh = TriplaneRenderer(head, dev(other), bound=1.0, mode="fused").render(ro, rd, *cond, max_steps=64, count_samples=True)
    for k in KEYS + ("ray_counts",):
        assert torch.equal(b[k], fresh[k]), (k, nb.data_ptr() == addr)
    assert int(b["state"][5]) > 100
    with torch.inference_mode():
        inf_bits = dev(other).clone()
    assert inf_bits.is_inference()
    ri = TriplaneRenderer(head, inf_bits, bound=1.0, mode="fused")
    c = ri.render(ro, rd, *cond, max_steps=64, count_samples=True)
    for k in KEYS + ("ray_counts",):
        assert torch.equal(c[k], fresh[k]), k


def test_cap_machinery_is_skipped_only_when_no_ray_can_reach_max_steps(params, golden):
    """_cap_can_bind: a ray holds at most diag(aabb) / dt_min + 1 samples.  The reference's aabb (y halved, renderer.py:110: diagonal 3 < 2 sqrt 3)
    never reaches a 192-step cap; the full cube can; the deployed max_steps = 16 (dt_min = dt_max) always does.  Where the renderer skips the
    schedule replay the frame equals the one rendered with it forced (count_samples asks for the marched counts, which need the schedule)."""
    from lzzx_nerf_amd.renderer import TriplaneRenderer
    head, bits, ro, rd, cond = setup(params, golden, 64, 64, "ones")
    r = TriplaneRenderer(head, dev(bits), bound=1.0, mode="fused")
    assert r._cap_can_bind(1 / 256, 192) is False and r._cap_can_bind(1 / 256, 16) is True
    assert r._cap_can_bind(1 / 256, 100) is True and r._cap_can_bind(1 / 256, 150) is False     # 3 / dt_max + 2 = 112.9 samples at most
    cube = TriplaneRenderer(head, dev(bits), bound=1.0, mode="fused", aabb=torch.tensor([-1.0, -1, -1, 1, 1, 1]))
    assert cube._cap_can_bind(1 / 256, 192) is True
    a = {k: v.clone() for k, v in r.render(ro, rd, *cond, max_steps=192).items()}                       # skipped (state word 10, C_eff, stays 0)
    b = r.render(ro, rd, *cond, max_steps=192, count_samples=True)                                       # forced
    assert int(a["state"][10]) == 0 and 0 < int(b["state"][10]) < 192     # forced: the replayed loop ends when nobody is alive (no ray holds 192 samples)
    for k in KEYS:
        assert torch.equal(a[k], b[k]), k
    r.aabb = torch.tensor([-1.0, -1, -1, 1, 1, 1], device="cuda")                                        # a new aabb tensor is looked at again
    assert r._cap_can_bind(1 / 256, 192) is True
