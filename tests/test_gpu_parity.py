"""GPU parity: the HIP path (through the C ABI) against the CPU oracle on identical seeded inputs.

Bar (BASELINE.json): ray and hitpoint counts identical; per-pixel accumulator equal to the
oracle's fp64 sum scaled by 1/spp and rounded to fp32.  The kernels use the reference's operation order in
fp64 without FMA contraction, so the comparison is made at 1e-6 absolute (fp32 rounding of values <= ~2), far
inside the 1e-4 tolerance the north star states; the exact-equality fraction is reported too."""
import os

import numpy as np
import pytest

import scenes
from backends import BackendScene, to_acc32

pytestmark = pytest.mark.gpu

CASES = [
    # name, scene factory, camera, W, H, spp, depth
    ("c1_spheres_depth1", scenes.scene_c1, scenes.cam_pinhole, 256, 256, 1, 1),
    ("c2_glass_pinhole", scenes.scene_c2, scenes.cam_pinhole, 192, 108, 1, 5),
    ("c2_glass_dof", scenes.scene_c2, scenes.cam_dof, 160, 90, 8, 5),
    ("c2_depth3", scenes.scene_c2, scenes.cam_dof, 96, 54, 4, 3),
    ("pyramid_diffuse", lambda: scenes.scene_pyramid(False), scenes.cam_pinhole, 96, 96, 1, 5),
    ("pyramid_glass", lambda: scenes.scene_pyramid(True), scenes.cam_dof, 96, 96, 4, 5),
    ("c3_bunny_glass_chess", lambda: scenes.scene_c3(True), scenes.cam_dof, 128, 128, 4, 5),
    ("dragon_diffuse", scenes.scene_dragon, scenes.cam_pinhole, 128, 128, 1, 5),
    ("bump_floor_glass_sphere",
     lambda: scenes.planes(scenes.stone_small_texture(True)) + [scenes.Sphere((5, -12, 30), 5, (1, 1, 1), 0.8, 0.5)],
     scenes.cam_dof, 96, 72, 2, 5),
    ("ragged_size", scenes.scene_c2, scenes.cam_dof, 67, 45, 3, 5),
    # Texture::color in all three orientations (texture.h:39-72) reached by primary, mirrored and refracted rays
    ("textured_walls", scenes.scene_textured_walls, scenes.cam_dof, 160, 120, 4, 5),
    # a refracting bump floor: displacement mesh through the tree path (counter observable), not the height field
    ("glass_bump_floor", scenes.scene_glass_bump_floor, scenes.cam_dof, 96, 72, 2, 5),
    # the reference's real floor texture at full size (texture/stone.jpg: 146 744 bump triangles)
    ("stone_full_bump_floor",
     lambda: scenes.planes(scenes.stone_texture(True)) + [scenes.Sphere((5, -12, 30), 5, (1, 1, 1), 0.8, 0.5)],
     scenes.cam_dof, 128, 96, 2, 5),
]


@pytest.mark.parametrize("name,mk,cam,W,H,spp,depth", CASES, ids=[c[0] for c in CASES])
def test_trace_grid_matches_oracle(gpu_ready, orc, name, mk, cam, W, H, spp, depth):
    import cgraytracing_amd as cg

    objs = mk()
    camera = cam()
    o = BackendScene(orc, objs)
    want = o.trace_grid(camera, W, H, spp, depth, seed=12345)
    o.close()
    sc = cg.Scene(objs)
    got = sc.trace_grid_host(W, H, spp, camera, depth, 12345)
    sc.close()
    assert got["nrays"] == want["nrays"], "ray count"
    assert got["nhp"] == int(want["nhit"].sum()), "hitpoint count"
    assert np.array_equal(got["nhit"], want["nhit"]), "per-pixel hitpoint counts"
    ref32 = to_acc32(want["acc_sum"], spp)
    diff = np.abs(got["rgb"].astype(np.float64) - ref32.astype(np.float64))
    exact = float((got["rgb"] == ref32).mean())
    print("%s: Linf=%.3e exact=%.6f rays=%d" % (name, diff.max(), exact, got["nrays"]))
    assert diff.max() <= 1e-6, "per-pixel RGB L-inf %g" % diff.max()
    assert exact > 0.999


GOLD = __import__("os").path.join(__import__("os").path.dirname(__import__("os").path.abspath(__file__)), "golden")


def bezier_report(name, got_rgb, ref32, nrays_got, nrays_want):
    """Bezier parity is statistical (device libm is not glibc, Newton from random starts is chaotic: SURVEY H3).  This prints --
    and stores under gpurun_out/bezier_parity/ -- WHICH pixels miss the north star's 1e-4 and by how much, so that a
    regression shows as names, not as a fraction.  Returns (fraction of pixels within 1e-4, L-inf, relative ray-count gap)."""
    import json
    err = np.abs(got_rgb.astype(np.float64) - ref32.astype(np.float64)).max(axis=-1)
    bad = np.argwhere(err >= 1e-4)
    frac = 1.0 - len(bad) / err.size
    gap = abs(nrays_got - nrays_want) / max(1, nrays_want)
    rec = {"name": name, "pixels": int(err.size), "within_1e-4": frac, "n_missing": int(len(bad)), "linf": float(err.max()),
           "exact_fraction": float((got_rgb == ref32).all(axis=-1).mean()), "rays_gpu": int(nrays_got), "rays_oracle": int(nrays_want),
           "missing_pixels_row_col_err": [[int(r), int(c), float(err[r, c])] for r, c in bad[:200]]}
    print("%s: %d of %d pixels miss 1e-4 (within: %.6f), Linf=%.3e, bit-equal %.6f, rays %d vs %d; missing: %s" %
          (name, len(bad), err.size, frac, err.max(), rec["exact_fraction"], nrays_got, nrays_want, rec["missing_pixels_row_col_err"][:12]))
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "bezier_parity")
    try:
        os.makedirs(out, exist_ok=True)
        json.dump(rec, open(os.path.join(out, name + ".json"), "w"), indent=1)
    except OSError:
        pass
    return frac, float(err.max()), gap


def test_function_level_intersect_vs_reference_golden(gpu_ready):
    """objs[i]->intersect() on the device against the compiled reference's answers (tests/golden/
    function_level.npz): sphere, triangle mesh (bunny) and plane are bit-exact in hit flag, distance and normal."""
    import cgraytracing_amd as cg
    g = np.load(__import__("os").path.join(GOLD, "function_level.npz"))
    s3 = cg.Scene(scenes.scene_c3(True))
    h, l, n = s3.intersect_rays(5, g["mesh_org"], g["mesh_dir"])
    m = h == 1
    assert np.array_equal(h, g["mesh_hit"]) and m.sum() > 100
    assert np.array_equal(l[m], g["mesh_len"][m]) and np.array_equal(n[m], g["mesh_n"][m])
    h, l, n = s3.intersect_rays(0, g["mesh_org"], g["mesh_dir"])
    m = h == 1
    assert np.array_equal(h, g["floor_hit"]) and np.array_equal(l[m], g["floor_len"][m])
    s3.close()
    s4 = cg.Scene(scenes.scene_c2())
    for k, ob in (("wall", 0), ("mirror", 6), ("glass", 7)):
        h, l, n = s4.intersect_rays(ob, g["sph_org"], g["sph_dir"])
        m = h == 1
        assert np.array_equal(h, g[k + "_hit"]), k
        assert np.array_equal(l[m], g[k + "_len"][m]) and np.array_equal(n[m], g[k + "_n"][m]), k
    s4.close()


def test_texture_color_three_orientations_vs_reference_golden(gpu_ready):
    """Texture::color on the DEVICE (cgrt_surface_colors -> texture_color) in all three orientations against the compiled
    reference's own getSurfaceColor answers (function_level.npz: back_col = |d.z| branch with the vertical flip,
    side_col = |d.x| branch, floor_col = |d.y| branch), points inside and outside the texture rectangle: bit for bit."""
    import cgraytracing_amd as cg
    from cgraytracing_amd.scene import Plane, Texture
    g = np.load(__import__("os").path.join(GOLD, "function_level.npz"))
    chess = scenes.load_asset("chessboard_rgb.npz")["rgb"]
    back = Plane((0, 0, 40), (0, 0, -1), (0.15, 0.15, 0.15), 0, 0, Texture(chess, (0, 0, -1), (-10, -10, 40), 20, 10))
    side = Plane((20, 0, 0), (-1, 0, 0), (0.15, 0.5, 0.15), 0, 0, Texture(chess, (-1, 0, 0), (20, -10, 10), 20, 25))
    with cg.Scene([back, side]) as sc:
        cb = sc.surface_colors(0, g["back_pts"])
        cs = sc.surface_colors(1, g["side_pts"])
    assert np.array_equal(cb, g["back_col"]) and np.array_equal(cs, g["side_col"])
    for c, flat in ((g["back_col"], (0.15, 0.15, 0.15)), (g["side_col"], (0.15, 0.5, 0.15))):
        inside = ~np.all(c == np.asarray(flat), axis=1)
        assert 0.1 < inside.mean() < 0.9  # both the textured and the flat-colour outcome are exercised
    with cg.Scene(scenes.scene_c3(True)) as s3:
        P = g["mesh_org"] + g["mesh_dir"] * g["floor_len"][:, None]
        assert np.array_equal(s3.surface_colors(0, P), g["floor_col"])


@pytest.mark.parametrize("transp", [0.0, 0.5], ids=["opaque_wide_walk", "glass_leaf_walk"])
def test_single_precision_box_tests_are_conservative(gpu_ready, orc, transp):
    """the walks test boxes in fp32 against per-ray grown boxes, which must accept every box the ray
    really touches whatever the magnitudes involved.  Meshes far from the origin (coordinates up to 3e4, where an fp32 ulp is
    2e-3 -- larger than many triangles' boxes are thick), rays from far away (1e6), rays with one or two direction components
    exactly zero, rays that start inside the mesh or on a box face, grazing rays: hit flag, distance and normal must be the
    oracle's, bit for bit, for an opaque owner (4-wide pruned walk) and a transparent one (leaf scan with per-triangle boxes)."""
    import cgraytracing_amd as cg
    rng = np.random.default_rng(77)
    for offset, scale in [((0.0, -5.0, 30.0), 1.0), ((1.0e4, -2.0e4, 3.0e4), 1.0), ((-3.0e4, 7.0, 11.0), 0.05), ((5.0, 5.0, 5.0), 300.0)]:
        off = np.asarray(offset)
        tris = (scenes.procedural_mesh(24, 12, (0.0, 0.0, 0.0), 8.0, seed=9) * scale + np.tile(off, 3)).reshape(-1, 9)
        objs = [scenes.TriangleMesh.from_triangles(tris, (0.6, 0.7, 0.9), 0.8 if transp else 0.0, transp)]
        P = tris.reshape(-1, 3, 3)
        lo, hi = P.min((0, 1)), P.max((0, 1))
        n = 4000
        tgt = P[rng.integers(0, len(P), n)].mean(1) + rng.normal(size=(n, 3)) * 0.3 * scale   # near the surface
        far = rng.choice([12.0 * scale, 300.0 * scale, 1.0e6], n)[:, None]
        dirs = rng.normal(size=(n, 3))
        dirs /= np.linalg.norm(dirs, axis=1)[:, None]
        org = tgt - dirs * far
        # axis-parallel rays (one or two components exactly zero), aimed at vertices' coordinates
        k = n // 4
        ax = rng.integers(0, 3, k)
        dirs[:k] = 0.0
        dirs[np.arange(k), ax] = rng.choice([-1.0, 1.0], k)
        mixed = rng.random(k) < 0.5
        dirs[:k][mixed, (ax[mixed] + 1) % 3] = rng.choice([-1.0, 1.0], mixed.sum()) * rng.uniform(0.1, 1, mixed.sum())
        dirs[:k] /= np.linalg.norm(dirs[:k], axis=1)[:, None]
        org[:k] = tgt[:k] - dirs[:k] * far[:k]
        # rays that start inside the mesh's box, some exactly on a face of it
        inside = rng.uniform(lo, hi, (k, 3))
        face = rng.random(k) < 0.3
        inside[face, 0] = lo[0]
        org[k:2 * k] = inside
        hw, lw, nw = BackendScene(orc, objs).intersect_batch(0, org, dirs)
        with cg.Scene(objs) as sc:
            hg, lg, ng = sc.intersect_rays(0, org, dirs)
        assert hw.sum() > n // 10
        assert np.array_equal(hg, hw), (offset, int((hg != hw).sum()))
        m = hw == 1
        assert np.array_equal(lg[m], lw[m]), offset
        if transp:
            assert np.array_equal(ng[m], nw[m]), offset
        else:  # an opaque owner's normal is defined up to the sign trace() gives it (main.cpp:73-76; PRUNE in cgrt_traverse.hpp)
            flip = lambda nv, dv: np.where(((nv * dv).sum(1) > 0)[:, None], -nv, nv)
            assert np.array_equal(flip(ng[m], dirs[m]), flip(nw[m], dirs[m])), offset


def test_tangent_rays_on_opaque_mesh(gpu_ready, orc):
    """The documented exception of the pruned traversal (cgrt_traverse.hpp, PRUNE): for an OPAQUE mesh the device
    drops the improvement counter, which only sets the SIGN of the returned normal; trace() re-orients the normal
    whenever n.d != 0 (main.cpp:73-76).  Rays lying exactly in a triangle's plane have det1 == 0 and miss it in the
    reference (inf/NaN comparisons, objects.h:101-104) and here; rays a few ulps off the plane hit with |n.d| ~ 1e-16
    and must give the same hit, distance and -- after trace()'s re-orientation -- the same Hitpoint normals."""
    import cgraytracing_amd as cg
    tris = scenes.pyramid_tris(1.0, (0.0, -5.0, 30.0))
    mesh = scenes.TriangleMesh.from_triangles(tris, (0.6, 0.7, 0.9), 0.0, 0.0)
    objs = [mesh]
    rng = np.random.default_rng(5)
    org, dirs = [], []
    for t in tris.reshape(-1, 3, 3):
        a, b, c = t
        nrm = np.cross(a - b, a - c)
        nrm /= np.linalg.norm(nrm)
        for _ in range(400):
            w = rng.dirichlet((1, 1, 1))
            target = w[0] * a + w[1] * b + w[2] * c              # a point of the triangle
            u = rng.normal(size=3)
            u -= nrm * (u @ nrm)                                   # direction inside the triangle's plane ...
            u /= np.linalg.norm(u)
            tilt = rng.choice([0.0, 0.0, 1e-17, -1e-17, 1e-15, -1e-15, 1e-12, -1e-12, 1e-9, -1e-9])
            d = u + nrm * tilt                                     # ... or tilted out of it by a few ulps
            d /= np.linalg.norm(d)
            org.append(target - d * rng.uniform(3, 30))
            dirs.append(d)
    org, dirs = np.asarray(org), np.asarray(dirs)
    hw, lw, nw = BackendScene(orc, objs).intersect_batch(0, org, dirs)
    with cg.Scene(objs) as sc:
        hg, lg, ng = sc.intersect_rays(0, org, dirs)
    assert np.array_equal(hg, hw) and 0.2 < hw.mean() < 0.99
    m = hw != 0
    assert np.array_equal(lg[m], lw[m])
    same = np.all(ng[m] == nw[m], axis=1) | np.all(ng[m] == -nw[m], axis=1)
    assert same.all()
    nd = np.einsum("ij,ij->i", nw[m], dirs[m])
    # what trace() does next (main.cpp:73-76): flip when n.d > 0.  Equal after the flip unless n.d == 0 exactly.
    fo = np.where((nd > 0)[:, None], -nw[m], nw[m])
    ndg = np.einsum("ij,ij->i", ng[m], dirs[m])
    fg = np.where((ndg > 0)[:, None], -ng[m], ng[m])
    exact_tangent = nd == 0
    assert np.array_equal(fo[~exact_tangent], fg[~exact_tangent])
    print("near-tangent hits: %d, min |n.d| = %.3e, exactly tangent hits: %d" % (m.sum(), np.abs(nd[~exact_tangent]).min(),
                                                                               int(exact_tangent.sum())))
    assert np.abs(nd).min() < 1e-12  # the fan really reaches the grazing regime


def test_bezier_intersect_vs_reference_golden(gpu_ready):
    """Bezier::intersect on the device vs the compiled reference, same keyed draws.  Device sin/cos/atan and
    the integer powers are not glibc's bit for bit, and Newton from random starts is chaotic, so the bar is:
    hit flags agree on >= 99.5 % of rays, and where both hit, distance and normal agree to 1e-6 on >= 99 %."""
    import cgraytracing_amd as cg
    g = np.load(__import__("os").path.join(GOLD, "function_level.npz"))
    s = cg.Scene([scenes.vase_bezier()])
    h, l, n = s.intersect_rays(0, g["bez_org"], g["bez_dir"], g["bez_keys"])
    s.close()
    agree = (h == g["bez_hit"]).mean()
    both = (h == 1) & (g["bez_hit"] == 1)
    close = np.abs(l[both] - g["bez_len"][both]) < 1e-6
    nclose = np.abs(n[both] - g["bez_n"][both]).max(axis=1) < 1e-6
    print("bezier: hit agreement %.4f, len within 1e-6: %.4f, normal: %.4f, max |dlen| on close: %.3e"
          % (agree, close.mean(), nclose.mean(), np.abs(l[both] - g["bez_len"][both])[close].max()))
    # measured in round 3: flags 1.0000, len 1.0000, normal 1.0000, |dlen| <= 2.9e-14 on the 1 024 golden rays
    assert agree == 1.0 and close.mean() >= 0.999 and nclose.mean() >= 0.999


# Bars for scenes with a Bezier object = what round 3 MEASURED on MI355X minus a margin (VERDICT r2 "weak" 1).  Measured
# (gpurun_out/bezier_parity/*.json, copied to profiles/r03_bezier_parity.json): bezier_scene 0 of 9 216 pixels miss 1e-4 and all
# are bit-equal, rays 18 699 = 18 699; everything_at_once 0 of 12 288, rays 32 331 = 32 331; c5_shape 0 of 16 384.  The margin
# allows for a handful of Newton outcomes flipping with another libm build (device pow/sin/cos are not glibc's): at most
# 5 pixels in 10 000 and 5 rays in 10 000 -- one tenth of what round 2 accepted.
BEZ_SCENE_BAR = (0.9995, 0.0005)  # (fraction of pixels within 1e-4, relative ray-count gap)


def test_bezier_scene_vs_oracle(gpu_ready, orc):
    """C5-shaped scene (planes + stone bump floor + Bezier vase, mirror-like refl 0.5): the oracle uses the
    same path-keyed draws, so the images agree except where Newton's outcome flips."""
    import cgraytracing_amd as cg
    objs = scenes.scene_c5(scenes.stone_small_texture(True))
    W, H, spp = 96, 96, 2
    o = BackendScene(orc, objs)
    want = o.trace_grid(scenes.cam_dof(), W, H, spp, 5, seed=7)
    o.close()
    sc = cg.Scene(objs)
    got = sc.trace_grid_host(W, H, spp, scenes.cam_dof(), 5, 7)
    sc.close()
    ref32 = to_acc32(want["acc_sum"], spp)
    frac, linf, gap = bezier_report("bezier_scene_96x96_spp2", got["rgb"], ref32, got["nrays"], want["nrays"])
    assert frac >= BEZ_SCENE_BAR[0] and gap <= BEZ_SCENE_BAR[1]


def test_stripes_and_sample_ranges_compose(gpu_ready):
    """Row stripes (the multi-GPU sharding) and split sample ranges reproduce the single-launch frame."""
    import cgraytracing_amd as cg
    from cgraytracing_amd.dist import assemble, local_rows
    import torch
    sc = cg.Scene(scenes.scene_c2())
    cam = scenes.cam_dof()
    W, H, spp = 100, 83, 4
    full = sc.trace_grid_host(W, H, spp, cam, 5, 99)
    for N, S in [(2, 8), (3, 16)]:
        parts, rays = [], 0
        for r in range(N):
            rows = local_rows(H, S, r, N)
            p = sc.trace_grid_host(W, H, spp, cam, 5, 99, rows=rows, stripe=(S, r, N))
            parts.append(torch.from_numpy(p["rgb"]))
            rays += p["nrays"]
        frame = assemble(torch.stack(parts), H, S, N).numpy()
        assert np.array_equal(frame, full["rgb"]), (N, S)
        assert rays == full["nrays"]
    a = sc.trace_grid_host(W, H, 1, cam, 5, 99, sample_offset=0, spp_total=4)
    b = sc.trace_grid_host(W, H, 3, cam, 5, 99, sample_offset=1, spp_total=4)
    assert np.abs((a["rgb"] + b["rgb"]) - full["rgb"]).max() < 1e-6
    assert a["nrays"] + b["nrays"] == full["nrays"]
    top = sc.trace_grid_host(W, H, spp, cam, 5, 99, rows=40, row_offset=43)
    assert np.array_equal(top["rgb"], full["rgb"][43:])
    sc.close()


def test_cpp_host_program_dropin(gpu_ready, orc, tmp_path):
    """examples/main_dropin.cpp (reference-style main() over include/cgrt_host.hpp, plain g++) end to end:
    C2 with the thin lens against the oracle, and a file-loaded glass mesh against the REFERENCE's golden."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "cgraytracing_amd", "cgrt_main")
    assert os.path.exists(exe), "build it with make -C cgraytracing_amd/csrc all"
    raw = str(tmp_path / "c2.f32")
    out = subprocess.run([exe, "--scene", "c2", "--width", "96", "--height", "54", "--spp", "4", "--dof", "--raw", raw],
                         capture_output=True, text=True, check=True).stdout
    o = BackendScene(orc, scenes.scene_c2())
    want = o.trace_grid(scenes.cam_dof(), 96, 54, 4, 5, seed=12345)
    got = np.fromfile(raw, np.float32).reshape(54, 96, 3)
    assert np.array_equal(got, to_acc32(want["acc_sum"], 4))
    assert "rays: %d " % want["nrays"] in out
    raw2 = str(tmp_path / "mesh.f32")
    subprocess.run([exe, "--scene", "planes", "--mesh", os.path.join(GOLD, "assets", "mesh_t0.txt"), "0", "--width", "48",
                    "--height", "48", "--raw", raw2], capture_output=True, text=True, check=True)
    g = np.load(os.path.join(GOLD, "loader_t0.npz"))
    got2 = np.fromfile(raw2, np.float32).reshape(48, 48, 3)
    assert np.array_equal(got2, to_acc32(g["acc_sum"], 1))
    # a textured, bump-mapped floor built from vector<vector<Vec3>> texels (byte/256), plus the Bezier vase
    raw3 = str(tmp_path / "chess.f32")
    subprocess.run([exe, "--scene", "chess", "--width", "80", "--height", "60", "--spp", "2", "--dof", "--raw", raw3],
                   capture_output=True, text=True, check=True)
    v = np.where((((np.arange(64)[:, None] // 8) + (np.arange(64)[None, :] // 8)) & 1) == 1, 230, 25).astype(np.uint8)
    rgb = np.stack([v, (v // 2 + 60).astype(np.uint8), (255 - v).astype(np.uint8)], -1)
    tex = scenes.Texture(np.ascontiguousarray(rgb), (0, 1, 0), (-21, 0, 0), 42, 40, True)
    o3 = BackendScene(orc, scenes.planes(tex))
    want3 = o3.trace_grid(scenes.cam_dof(), 80, 60, 2, 5, seed=12345)
    got3 = np.fromfile(raw3, np.float32).reshape(60, 80, 3)
    assert np.array_equal(got3, to_acc32(want3["acc_sum"], 2))


def test_cpp_sharded_host_program(gpu_ready, orc, tmp_path):
    """examples/render_sharded.cpp over include/cgrt_host_sharded.hpp (one host thread per GPU, block-cyclic stripes,
    ncclSend/ncclRecv gather): on this one-GPU box it runs with N = 1 -- scene replicated once, whole frame, no communicator --
    and must give the oracle's frame; the N > 1 path is the same code plus the grouped gather and is unverified here."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "cgraytracing_amd", "cgrt_sharded")
    assert os.path.exists(exe), "build it with make -C cgraytracing_amd/csrc all"
    raw = str(tmp_path / "c2.f32")
    out = subprocess.run([exe, "--gpus", "1", "--width", "160", "--height", "88", "--spp", "4", "--dof", "--raw", raw],
                         capture_output=True, text=True, check=True).stdout
    want = BackendScene(orc, scenes.scene_c2()).trace_grid(scenes.cam_dof(), 160, 88, 4, 5, seed=12345)
    got = np.fromfile(raw, np.float32).reshape(88, 160, 3)
    assert np.array_equal(got, to_acc32(want["acc_sum"], 4))
    assert "gpus: 1 rays: %d " % want["nrays"] in out


def _canon(hp, pix, smp):
    """Order-independent form of a hitpoint stream: sort by (pixel, sample, f, pos, normal)."""
    keys = [hp[:, k] for k in range(8, -1, -1)] + [smp, pix]
    idx = np.lexsort(keys)
    return hp[idx], pix[idx], smp[idx]


def test_hitpoint_stream_matches_reference_golden(gpu_ready):
    """Every Hitpoint record {f, pos, normal} the compiled REFERENCE stored (golden trace fixtures, emission order)
    against the GPU's hitpoint stream for the same grid: identical multisets, bit for bit (SURVEY.md 8d asks for
    pos/normal per hitpoint because flat-coloured walls make the accumulator insensitive to t and n errors)."""
    import os
    import sys
    import cgraytracing_amd as cg
    sys.path.insert(0, GOLD)
    import make_golden
    total = 0
    for name, mk, cam, W, H, spp, depth in make_golden.trace_cases():
        g = np.load(os.path.join(GOLD, "trace_%s.npz" % name))
        sc = cg.Scene(mk())
        r = sc.trace_grid_hitpoints(W, H, spp, cam(), depth, 12345)
        sc.close()
        assert r["count"] == len(g["hp"]), name
        a = _canon(r["hp"], r["pix"], r["smp"])
        b = _canon(g["hp"], g["hp_pix"].astype(np.int64), g["hp_smp"].astype(np.int64))
        assert np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2]), name
        assert np.array_equal(a[0], b[0]), "%s: hitpoint records differ" % name
        total += r["count"]
    assert total > 60000


def test_progressive_accumulation(gpu_ready, orc):
    """CGRT_GRID_ACCUMULATE: four passes of 2 samples added into one fp32 frame equal the single 8-sample launch
    to fp32 rounding, and both match the oracle (row f4 of SURVEY.md section 8: replaces average.cpp)."""
    import cgraytracing_amd as cg
    import torch
    sc = cg.Scene(scenes.scene_c3(True))
    cam = scenes.cam_dof()
    W, H, spp = 96, 96, 8
    one, _, _ = sc.trace_grid(W, H, spp, cam, 5, 21)
    acc = torch.zeros_like(one)
    for p in range(4):
        sc.trace_grid(W, H, 2, cam, 5, 21, sample_offset=2 * p, spp_total=spp, out=acc, accumulate=True)
    torch.cuda.synchronize()
    assert float((acc - one).abs().max()) < 2e-6
    o = BackendScene(orc, scenes.scene_c3(True))
    want = to_acc32(o.trace_grid(cam, W, H, spp, 5, 21)["acc_sum"], spp)
    assert np.array_equal(one.cpu().numpy(), want)
    assert np.abs(acc.cpu().numpy() - want).max() < 2e-6
    # passes of 4 samples are cost-scheduled (heavy tiles through the ordered sum, light tiles on the second stream): the
    # accumulated frame must equal, bit for bit, the same passes rendered in image order
    a, b = torch.zeros_like(one), torch.zeros_like(one)
    for p in range(2):
        sc.trace_grid(W, H, 4, cam, 5, 21, sample_offset=4 * p, spp_total=spp, out=a, accumulate=True)
        sc.trace_grid(W, H, 4, cam, 5, 21, sample_offset=4 * p, spp_total=spp, out=b, accumulate=True, reorder=False)
    torch.cuda.synchronize()
    assert torch.equal(a, b) and float((a - one).abs().max()) < 2e-6
    sc.close()


def test_everything_at_once(gpu_ready, orc):
    """One scene with every object kind and material: glass + mirror + diffuse spheres, textured bump floor, a glass
    mesh (LDS-cached tree), an opaque mesh (pruned traversal), the Bezier vase; thin lens, depth 5.  Exercises the
    most general kernel variant (TREES, BEZ, DOF, GLASS) and its largest LDS carve-up."""
    import cgraytracing_amd as cg
    objs = [scenes.Sphere((-12.0, -14.0, 32), 5, (1.0, 1.0, 1.0), 0.8, 0.5),
            scenes.Sphere((-2.0, -16.0, 24), 3.5, (0.9, 0.9, 1.0), 0.8, 0.0),
            scenes.Sphere((6.0, -17.0, 22), 2.5, (0.8, 0.3, 0.3), 0.0, 0.0)]
    objs += scenes.planes(scenes.stone_small_texture(True))
    objs.append(scenes.TriangleMesh.from_triangles(scenes.bunny_tris() * 0.6 + np.tile([4.0, -2.0, 10.0], 3),
                                                   (1.0, 1.0, 1.0), 0.8, 0.5))
    objs.append(scenes.TriangleMesh.from_triangles(scenes.pyramid_tris(0.6, (-6.0, -13.0, 38.0)), (0.6, 0.7, 0.9), 0.0, 0.0))
    objs.append(scenes.vase_bezier())
    W, H, spp = 128, 96, 2
    o = BackendScene(orc, objs)
    want = o.trace_grid(scenes.cam_dof(), W, H, spp, 5, seed=3)
    o.close()
    sc = cg.Scene(objs)
    got = sc.trace_grid_host(W, H, spp, scenes.cam_dof(), 5, 3)
    hp = sc.trace_grid_hitpoints(W, H, spp, scenes.cam_dof(), 5, 3)
    sc.close()
    ref32 = to_acc32(want["acc_sum"], spp)
    frac, linf, gap = bezier_report("everything_at_once_128x96_spp2", got["rgb"], ref32, got["nrays"], want["nrays"])
    assert frac >= BEZ_SCENE_BAR[0] and gap <= BEZ_SCENE_BAR[1]
    assert hp["count"] == got["nhp"]


def _bump_floor_rays(tex, n, seed):
    """Rays aimed at a bump floor from everywhere: from above at all angles, grazing, from inside the layer of
    heights, straight down, axis-aligned, and -- to force exact ties -- rays lying IN the vertical planes x = x_j and
    z = z_i of the quad grid (vertex coordinates as objects.h:485-488 computes them), which cross shared edges."""
    rng = np.random.default_rng(seed)
    R, C = tex.data.shape[:2]
    px, pz, lx, lz = tex.position[0], tex.position[2], tex.lenx, tex.leny
    o = np.stack([rng.uniform(px - 3, px + lx + 3, n), rng.uniform(-20.2, 5.0, n), rng.uniform(pz - 3, pz + lz + 3, n)], 1)
    tgt = np.stack([rng.uniform(px - 1, px + lx + 1, n), rng.uniform(-20.1, -19.4, n), rng.uniform(pz - 1, pz + lz + 1, n)], 1)
    d = tgt - o
    k = n // 8
    o[:k, 1] = rng.uniform(-20.0, -19.5, k)                     # start inside the layer
    d[k:2 * k] = [0.0, -1.0, 0.0]                                # straight down
    d[2 * k:3 * k, 2] = 0.0                                      # in a plane z = const
    d[3 * k:4 * k, 0] = 0.0                                      # in a plane x = const
    jj = rng.integers(0, C // 3, k)
    o[4 * k:5 * k, 0] = px + lx * jj * 3 / C                     # ... exactly on a grid line x = x_j
    d[4 * k:5 * k, 0] = 0.0
    ii = rng.integers(0, R // 3, k)
    o[5 * k:6 * k, 2] = pz + lz * ii * 3 / R                     # ... exactly on a grid line z = z_i
    d[5 * k:6 * k, 2] = 0.0
    d[6 * k:7 * k, 1] *= 0.02                                    # grazing
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    return o, d


@pytest.mark.parametrize("which", ["stone_small", "chessboard", "procedural_stone", "stone"])
def test_bump_floor_grid_walk_vs_oracle(gpu_ready, orc, which):
    """Plane::intersect with a bump map (objects.h:505-524) for an opaque floor runs the height-field walk on the
    device while the oracle runs the reference's tree: hit flag and distance must be identical
    bit for bit, the normal identical up to its sign (the sign comes from the improvement counter, which trace()
    overrides, main.cpp:73-76), including rays through shared edges and vertices."""
    import cgraytracing_amd as cg
    tex = {"stone_small": lambda: scenes.stone_small_texture(True), "chessboard": lambda: scenes.chessboard_texture(True),
           "procedural_stone": lambda: scenes.Texture(scenes.procedural_stone(), (0, 1, 0), (-21, 0, 0), 42, 40, True),
           "stone": lambda: scenes.stone_texture(True)}[which]()
    objs = scenes.planes(tex)
    o, d = _bump_floor_rays(tex, 40000 if which not in ("procedural_stone", "stone") else 16000, 3)
    hw, lw, nw = BackendScene(orc, objs).intersect_batch(0, o, d)
    sc = cg.Scene(objs)
    hg, lg, ng = sc.intersect_rays(0, o, d)
    sc.close()
    assert np.array_equal(hg, hw)
    m = hw != 0
    assert np.array_equal(lg[m], lw[m])
    same = np.all(ng[m] == nw[m], axis=1) | np.all(ng[m] == -nw[m], axis=1)
    assert same.all()
    bumped = m & (np.abs(nw[:, 1]) < 1.0)
    print("%s: %d rays, %d hit the floor, %d of them on a tilted facet" % (which, len(o), m.sum(), bumped.sum()))
    assert bumped.sum() > 1000


@pytest.mark.parametrize("mode", ["sah", "ref"])
def test_both_hierarchies_are_exact(gpu_ready, orc, mode, monkeypatch):
    """The device may walk either hierarchy over the reference's leaves -- its own SAH trees (per direction octant;
    triangle-level for the opaque dragon-like mesh, leaf-level for the glass bunny) or, with CGRT_TREE=ref, the reference's
    inner nodes; both must give the oracle's accumulator, hit counts and ray counts bit for bit."""
    import cgraytracing_amd as cg
    if mode == "ref":
        monkeypatch.setenv("CGRT_TREE", "ref")
    else:
        monkeypatch.delenv("CGRT_TREE", raising=False)
    opaque = scenes.TriangleMesh.from_triangles(scenes.procedural_mesh(48, 36, (6.0, -12.0, 28.0), 6.0), (0.25, 0.25, 0.5), 0.0, 0.0, 1)
    mirror = scenes.TriangleMesh.from_triangles(scenes.procedural_mesh(24, 18, (-9.0, -13.0, 33.0), 4.0), (0.9, 0.9, 0.9), 0.8, 0.0, 1)
    objs = scenes.scene_c3(True) + [opaque, mirror]
    W, H, spp = 96, 80, 4
    want = BackendScene(orc, objs).trace_grid(scenes.cam_dof(), W, H, spp, 5, seed=4242)
    sc = cg.Scene(objs)
    got = sc.trace_grid_host(W, H, spp, scenes.cam_dof(), 5, 4242)
    sc.close()
    assert got["nrays"] == want["nrays"]
    assert np.array_equal(got["nhit"], want["nhit"])
    assert np.array_equal(got["rgb"], to_acc32(want["acc_sum"], spp))


def _random_mesh(rng, ntri, center, size, dup_frac):
    """Random triangle soup around `center`: some triangles share vertices / edges, a fraction are exact duplicates of
    others (equal distances from different leaves: the tie rules), a few are degenerate (zero area)."""
    verts = center + (rng.random((max(4, ntri), 3)) - 0.5) * size
    tris = []
    for _ in range(ntri):
        if tris and rng.random() < dup_frac:
            tris.append(tris[rng.integers(len(tris))].copy())
        else:
            idx = rng.integers(0, len(verts), 3)
            if rng.random() < 0.03:
                idx[2] = idx[1]  # degenerate
            tris.append(verts[idx].reshape(9))
    return np.asarray(tris, np.float64)


@pytest.mark.parametrize("seed", range(int(os.environ.get("CGRT_FUZZ_SEEDS", "12"))))
def test_random_scenes_match_oracle_exactly(gpu_ready, orc, seed):
    """Fuzz: random triangle soups (duplicates, shared edges, degenerate triangles; 1 to ~400 triangles, so trees from a
    single leaf to several levels) as opaque, mirror and glass objects, plus random spheres, inside the reference's box of
    planes; pinhole or thin lens.  Accumulator, per-pixel hit counts and ray count must equal the oracle's exactly."""
    import cgraytracing_amd as cg
    rng = np.random.default_rng(1000 + seed)
    objs = [scenes.Sphere(tuple(rng.uniform((-12, -15, 20), (12, 5, 36))), float(rng.uniform(1.5, 4.0)),
                          tuple(rng.uniform(0.2, 1.0, 3)), *[(0.0, 0.0), (0.8, 0.0), (0.8, 0.5)][rng.integers(3)])
            for _ in range(int(rng.integers(0, 3)))]
    objs += scenes.planes(scenes.chessboard_texture(bool(seed % 2)) if seed % 3 == 0 else None)
    for k in range(int(rng.integers(1, 4))):
        ntri = int([1, 9, 10, 11, 40, 150, 400][rng.integers(7)])
        refl, transp = [(0.0, 0.0), (0.8, 0.0), (0.8, 0.5)][(seed + k) % 3]
        tri = _random_mesh(rng, ntri, rng.uniform((-10, -16, 22), (10, 0, 34)), float(rng.uniform(3.0, 9.0)), 0.15)
        objs.append(scenes.TriangleMesh.from_triangles(tri, tuple(rng.uniform(0.2, 1.0, 3)), refl, transp, int(rng.integers(0, 3))))
    cam = scenes.cam_dof() if seed % 2 else scenes.cam_pinhole()
    # spp >= 4 goes through the cost scheduler (probe, heavy-tile unit queue, ordered sum) and, when the planes are plain,
    # the light-tile split on the second stream; spp 2 is the plain image-order launch
    W, H, spp = 64, 48, (2 if seed % 4 == 0 else 4 + seed % 3)
    want = BackendScene(orc, objs).trace_grid(cam, W, H, spp, 5, seed=77 + seed)
    sc = cg.Scene(objs)
    got = sc.trace_grid_host(W, H, spp, cam, 5, 77 + seed)
    sc.close()
    assert got["nrays"] == want["nrays"]
    assert np.array_equal(got["nhit"], want["nhit"])
    assert np.array_equal(got["rgb"], to_acc32(want["acc_sum"], spp))


@pytest.mark.parametrize("bump", [False, True])
def test_single_ray_scenes_match_oracle_exactly(gpu_ready, orc, bump):
    """Diffuse planes (the floor bump-mapped or plainly textured) and diffuse spheres: every ray tree is one ray, the launch stays
    in image order at any sample count (DeviceScene::single_ray) and the walls behind a bump floor form the plane group's run.
    Accumulator, hit counts and ray count equal the oracle's."""
    import cgraytracing_amd as cg
    objs = scenes.planes(scenes.chessboard_texture(bump)) + [scenes.Sphere((4.0, -14.0, 28.0), 5.0, (0.8, 0.6, 0.4), 0.0, 0.0),
                                                             scenes.Sphere((-9.0, -16.0, 33.0), 3.0, (0.3, 0.7, 0.9), 0.0, 0.0)]
    W, H, spp = 96, 64, 6
    for cam in (scenes.cam_pinhole(), scenes.cam_dof()):
        want = BackendScene(orc, objs).trace_grid(cam, W, H, spp, 5, seed=99)
        with cg.Scene(objs) as sc:
            assert "sched" not in sc.kernel_variant(W, H, spp, cam)
            got = sc.trace_grid_host(W, H, spp, cam, 5, 99)
        assert got["nrays"] == want["nrays"] == W * H * spp
        assert np.array_equal(got["nhit"], want["nhit"])
        assert np.array_equal(got["rgb"], to_acc32(want["acc_sum"], spp))


@pytest.mark.parametrize("variant", ["edges", "origin_in_a_plane", "thin_lens"])
def test_plane_group_test_edges_corners_and_degenerate_rays(gpu_ready, orc, variant):
    """The leading run of axis-aligned planes is tested as a group (cgrt_scene_walk.hpp plane_run): approximate distances pick
    the winner, one exact division yields its distance, and a lane whose two nearest planes are within 2^-14 of each other -- or
    that has a zero numerator or denominator -- takes the planes one by one.  A room built so that whole pixel columns and
    rows look exactly into its edges (and four pixels into its corners): with the pinhole camera at (0,0,-10) and half-width
    10 the rays of column w = 48 of 64 have x : z = 5 : 10 and meet the walls x = 12 and z = 14 on their common edge; rows
    likewise for floor and ceiling.  The centre column and row have d.x = 0 resp. d.y = 0 (a ray parallel to two walls).
    `origin_in_a_plane` adds a plane through the camera (numerator 0 for every primary ray); `thin_lens` moves the origins off
    the axis.  One wall is a mirror, so secondary rays start on a plane.  Everything must equal the oracle exactly."""
    import cgraytracing_amd as cg
    P = scenes.Plane
    objs = [P((0.0, -12.0, 0), (0, 1, 0), (0.9, 0.2, 0.2), 0.0, 0.0), P((12.0, 0.0, 0), (-1, 0, 0), (0.2, 0.9, 0.2), 0.0, 0.0),
            P((-12.0, 0.0, 0), (1, 0, 0), (0.2, 0.2, 0.9), 0.8, 0.0), P((0.0, 0.0, 14.0), (0, 0, -1), (0.9, 0.9, 0.2), 0.0, 0.0),
            P((0.0, 12.0, 0), (0, -1, 0), (0.2, 0.9, 0.9), 0.0, 0.0)]
    if variant == "origin_in_a_plane":
        objs.append(P((0.0, 0.0, -10.0), (0, 0, 1), (0.5, 0.5, 0.5), 0.0, 0.0))
    objs.append(scenes.Sphere((3.0, -4.0, 6.0), 2.0, (0.7, 0.7, 0.7), 0.0, 0.0))
    cam = scenes.cam_dof() if variant == "thin_lens" else scenes.cam_pinhole()
    W = H = 64
    for spp in (1, 5):  # image order, and the scheduled path with its light-tile launch
        want = BackendScene(orc, objs).trace_grid(cam, W, H, spp, 5, seed=4242)
        with cg.Scene(objs) as sc:
            got = sc.trace_grid_host(W, H, spp, cam, 5, 4242)
        assert got["nrays"] == want["nrays"]
        assert np.array_equal(got["nhit"], want["nhit"])
        assert np.array_equal(got["rgb"], to_acc32(want["acc_sum"], spp))


@pytest.mark.parametrize("which", ["sphere", "plane", "mesh", "vase"])
def test_cpp_object_intersect_matches_oracle(gpu_ready, orc, which):
    """Object::intersect / intersect_batch of include/cgrt_host.hpp -- the reference's virtual (objects.h:20) -- called from
    a plain C++ program (tests/native/host_intersect.cpp) on the same objects the oracle builds: hit flags, distances
    and normals identical bit for bit (Bezier: the statistical bar of DESIGN.md section 2)."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "cgraytracing_amd", "cgrt_host_intersect")
    assert os.path.exists(exe), "build it with make -C cgraytracing_amd/csrc all"
    mesh_file = os.path.join(GOLD, "assets", "mesh_t0.txt")
    obj = {"sphere": lambda: scenes.Sphere((-8.0, -13.0, 25), 7, (1.0, 1.0, 1.0), 0.8, 0.5),
           "plane": lambda: scenes.Plane((0.0, -20, 0), (0, 1, 0), (0.15, 0.15, 0.15), 0.0, 0.0),
           "mesh": lambda: scenes.TriangleMesh(mesh_file, 3.0, (1.0, -4.0, 30.0), (0.6, 0.7, 0.9), 0.8, 0.5, 0),
           "vase": scenes.vase_bezier}[which]()
    rng = np.random.default_rng(17)
    n = 600
    o = np.tile(np.array([0.0, 0.0, -10.0]), (n, 1)) + rng.normal(0, 0.5, (n, 3))
    target = {"sphere": (-8.0, -13.0, 25.0), "plane": (0.0, -20.0, 30.0), "mesh": (1.0, -4.0, 30.0), "vase": (15.0, -10.0, 35.0)}[which]
    d = np.asarray(target) + rng.normal(0, 4.0, (n, 3)) - o
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    text = "\n".join("%.17g %.17g %.17g %.17g %.17g %.17g" % tuple(np.concatenate([o[i], d[i]])) for i in range(n))
    args = [exe, which] + ([mesh_file] if which == "mesh" else [])
    out = subprocess.run(args, input=text, capture_output=True, text=True, check=True).stdout
    got = np.array([[float(x) for x in line.split()] for line in out.strip().splitlines()])
    hw, lw, nw = BackendScene(orc, [obj]).intersect_batch(0, o, d, keys=np.zeros(n, np.uint64) if which == "vase" else None)
    assert got.shape == (n, 5) and hw.sum() > 50
    if which == "vase":
        agree = (got[:, 0] == hw)
        close = np.abs(got[:, 1] - lw)[agree & (hw != 0)] < 1e-6
        assert agree.mean() >= 0.995 and close.mean() >= 0.99
        return
    assert np.array_equal(got[:, 0], hw)
    m = hw != 0
    assert np.array_equal(got[m, 1], lw[m]) and np.array_equal(got[m, 2:], nw[m])


def test_split_samples_mode(gpu_ready, orc):
    """CGRT_GRID_SPLIT_SAMPLES: several workgroups share a tile's samples and the chunk sums are added in chunk order.
    Ray and hit counts are exactly the unsplit render's (the same rays are traced); the image is reproducible run to run
    and differs from the sample-by-sample fp64 sum only by the summation order: <= 1e-6 here, far inside the 1e-4 the
    north star allows (most pixels identical after the rounding to fp32).  Also with stripes, hit-count plane and
    progressive accumulation."""
    import cgraytracing_amd as cg
    import torch
    objs, cam = scenes.scene_c2(), scenes.cam_dof()
    W, H, spp = 160, 96, 96
    want = BackendScene(orc, objs).trace_grid(cam, W, H, spp, 5, seed=12345)
    sc = cg.Scene(objs)
    plain = sc.trace_grid_host(W, H, spp, cam, 5, 12345)
    a = sc.trace_grid_host(W, H, spp, cam, 5, 12345, split_samples=True)
    b = sc.trace_grid_host(W, H, spp, cam, 5, 12345, split_samples=True)
    assert np.array_equal(plain["rgb"], to_acc32(want["acc_sum"], spp))
    assert a["nrays"] == want["nrays"] and np.array_equal(a["nhit"], want["nhit"])
    assert np.array_equal(a["rgb"], b["rgb"])  # reproducible
    assert np.abs(a["rgb"] - plain["rgb"]).max() <= 1e-6
    print("split samples: pixels identical to the unsplit render: %.5f" % (a["rgb"] == plain["rgb"]).all(axis=2).mean())
    # stripes compose the same frame; two accumulated half passes equal one split pass up to fp32 addition
    n, S = 2, 8
    from cgraytracing_amd import dist as cdist
    rows_local = cdist.local_rows(H, S, 0, n)
    parts = [sc.trace_grid_host(W, H, spp, cam, 5, 12345, rows=rows_local, stripe=(S, r, n), split_samples=True)["rgb"] for r in range(n)]
    frame = cdist.assemble(torch.from_numpy(np.stack(parts)), H, S, n).numpy()
    assert np.array_equal(frame, a["rgb"])
    out = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
    sc.trace_grid(W, H, spp // 2, cam, 5, 12345, out=out, sample_offset=0, spp_total=spp, split_samples=True)
    sc.trace_grid(W, H, spp // 2, cam, 5, 12345, out=out, sample_offset=spp // 2, spp_total=spp, accumulate=True, split_samples=True)
    torch.cuda.synchronize()
    assert np.abs(out.cpu().numpy() - plain["rgb"]).max() <= 1e-6
    sc.close()


@pytest.mark.parametrize("name,mk,cam,W,H,spp", [
    ("c2", scenes.scene_c2, scenes.cam_dof, 333, 187, 8),
    ("bunny_glass", lambda: scenes.scene_c3(True), scenes.cam_dof, 256, 200, 6),
    ("dragon", scenes.scene_dragon, scenes.cam_pinhole, 200, 160, 4),
    ("vase", lambda: scenes.scene_c5(scenes.stone_small_texture(True)), scenes.cam_dof, 128, 96, 4),
])
def test_cost_ordered_schedule_changes_nothing(gpu_ready, name, mk, cam, W, H, spp):
    """Probe -> sort -> render (the default for >= 4 samples per pixel: wave tiles rendered heaviest first) against
    CGRT_GRID_NO_REORDER (image order): identical image bits, per-pixel hitpoint counts and counters -- the probe's rays
    are not counted twice --, also in stripes, with a sample offset, with accumulation and with split samples."""
    import cgraytracing_amd as cg
    with cg.Scene(mk()) as sc:
        a = sc.trace_grid_host(W, H, spp, cam(), 5, 77, force_reorder=True)  # sphere scenes are scheduled only on request
        b = sc.trace_grid_host(W, H, spp, cam(), 5, 77, reorder=False)
        assert np.array_equal(a["rgb"], b["rgb"]) and np.array_equal(a["nhit"], b["nhit"])
        assert np.array_equal(a["counters"][:2], b["counters"][:2]) and a["nrays"] >= W * H * spp
        # stripes of rank 1 of 3, second half of the samples
        kw = dict(rows=64, stripe=(8, 1, 3), sample_offset=spp // 2, spp_total=spp)
        c = sc.trace_grid_host(W, H, spp - spp // 2, cam(), 5, 77, force_reorder=True, **kw)
        d = sc.trace_grid_host(W, H, spp - spp // 2, cam(), 5, 77, reorder=False, **kw)
        assert np.array_equal(c["rgb"], d["rgb"]) and np.array_equal(c["counters"][:2], d["counters"][:2])
        if name != "vase":  # Bezier scenes always split samples
            e = sc.trace_grid_host(W, H, 64, cam(), 5, 77, split_samples=True, force_reorder=True)
            f = sc.trace_grid_host(W, H, 64, cam(), 5, 77, split_samples=True, reorder=False)
            assert np.array_equal(e["rgb"], f["rgb"]) and np.array_equal(e["counters"][:2], f["counters"][:2])


@pytest.mark.parametrize("seed", range(int(os.environ.get("CGRT_FUZZ_SEEDS_LIGHT", "6"))))
def test_light_tile_classification_is_conservative(gpu_ready, orc, seed):
    """The light-tile split: frames with plain diffuse planes and small "special" objects -- a mesh, a
    mirror and a glass sphere, a Bezier vase -- placed near the frame's edge, near the camera, behind other objects and at
    depths on both sides of the focus plane, thin lens and pinhole.  Most wave tiles are light, the objects' silhouettes
    (blurred by the lens) must all lie in full tiles: image, hit counts and ray count equal the oracle's exactly."""
    import cgraytracing_amd as cg
    rng = np.random.default_rng(500 + seed)
    objs = scenes.planes()
    for _ in range(3):
        c = rng.uniform((-17, -17, 2), (17, 15, 38))
        objs.append(scenes.Sphere(tuple(c), float(rng.uniform(0.4, 2.5)), (0.9, 0.9, 0.9), 0.8, float(rng.choice([0.0, 0.5]))))
    objs.append(scenes.Sphere(tuple(rng.uniform((-15, -15, 10), (15, 10, 35))), 2.0, (0.3, 0.5, 0.7), 0.0, 0.0))  # diffuse: not special
    tri = _random_mesh(rng, 60, rng.uniform((-14, -16, 4), (14, 10, 36)), float(rng.uniform(1.0, 5.0)), 0.1)
    objs.append(scenes.TriangleMesh.from_triangles(tri, (0.6, 0.7, 0.9), *[(0.0, 0.0), (0.8, 0.0), (0.8, 0.5)][seed % 3]))
    if seed % 2:
        objs.append(scenes.Bezier([(0, -3, 1.2), (0, 0.5, 1.2), (0, -0.5, 0), (0, 3, 0.6)], tuple(rng.uniform((-12, -10, 15), (12, 5, 30))),
                                  (1.0, 1.0, 1.0), 0.5, 0.0))
    cam = scenes.cam_dof() if seed % 3 else scenes.cam_pinhole()
    W, H, spp = 256, 144, 4
    want = BackendScene(orc, objs).trace_grid(cam, W, H, spp, 5, seed=9 + seed)
    with cg.Scene(objs) as sc:
        got = sc.trace_grid_host(W, H, spp, cam, 5, 9 + seed)
        nat = sc.trace_grid_host(W, H, spp, cam, 5, 9 + seed, reorder=False)
    assert np.array_equal(got["rgb"], nat["rgb"]) and got["nrays"] == nat["nrays"]
    if seed % 2 == 0:  # no Bezier object: bit-exact against the oracle as well
        assert got["nrays"] == want["nrays"] and np.array_equal(got["nhit"], want["nhit"])
        assert np.array_equal(got["rgb"], to_acc32(want["acc_sum"], spp))


@pytest.mark.parametrize("kind,lens", [("mirror", True), ("glass", True), ("diffuse", False), ("mirror", False)])
def test_light_tiles_around_a_long_thin_mesh(gpu_ready, orc, kind, lens):
    """Cover spheres: a ribbon of 2 400 triangles running diagonally through the frame and through the focus
    plane gets 32 spheres over median-split triangle groups instead of one sphere around everything, so the corners of that one
    sphere are light tiles now.  Every tile the ribbon (blurred by the lens) can reach must still be full: scheduled render ==
    image-order render == oracle, bit for bit, image, hit counts and ray count."""
    import cgraytracing_amd as cg
    rng = np.random.default_rng(77)
    n = 1200
    a, b = np.array([-16.0, -14.0, 8.0]), np.array([16.0, 10.0, 36.0])
    t = np.linspace(0.0, 1.0, n + 1)
    mid = a[None, :] + (b - a)[None, :] * t[:, None] + rng.normal(0, 0.05, (n + 1, 3))
    side = np.cross(b - a, [0.0, 0.0, 1.0])
    side = 0.35 * side / np.linalg.norm(side)
    tris = []
    for i in range(n):
        p0, p1 = mid[i] - side, mid[i] + side
        q0, q1 = mid[i + 1] - side, mid[i + 1] + side
        tris += [[p0, p1, q0], [p1, q1, q0]]
    tri = np.asarray(tris, dtype=np.float64)
    refl, transp = {"mirror": (0.8, 0.0), "glass": (0.8, 0.5), "diffuse": (0.0, 0.0)}[kind]
    objs = scenes.planes() + [scenes.TriangleMesh.from_triangles(tri, (0.7, 0.8, 0.9), refl, transp)]
    cam = scenes.cam_dof() if lens else scenes.cam_pinhole()
    # 1280 x 800: a wave tile's cone is ~0.7 units wide at the ribbon's depth, well below the cover spheres' radii (at 320 x 200
    # the cone's own margin would hide a wrong radius: checked by shrinking the radii to zero, which this size catches)
    W, H, spp = 1280, 800, 4
    with cg.Scene(objs) as sc:
        got = sc.trace_grid_host(W, H, spp, cam, 5, 31)
        nat = sc.trace_grid_host(W, H, spp, cam, 5, 31, reorder=False)
        part = sc.trace_grid_host(W, H, spp, cam, 5, 31, rows=272, stripe=(8, 2, 3))  # the stripes of rank 2 of 3
        part_nat = sc.trace_grid_host(W, H, spp, cam, 5, 31, rows=272, stripe=(8, 2, 3), reorder=False)
    assert np.array_equal(got["rgb"], nat["rgb"]) and np.array_equal(got["nhit"], nat["nhit"]) and got["nrays"] == nat["nrays"]
    assert np.array_equal(part["rgb"], part_nat["rgb"]) and part["nrays"] == part_nat["nrays"]
    if kind == "glass":  # one case against the oracle as well (the image-order launch is pinned to it by every other exact test)
        want = BackendScene(orc, objs).trace_grid(cam, W, H, spp, 5, seed=31)
        assert got["nrays"] == want["nrays"] and np.array_equal(got["nhit"], want["nhit"])
        assert np.array_equal(got["rgb"], to_acc32(want["acc_sum"], spp))


@pytest.mark.parametrize("refl,lens", [(0.8, False), (0.8, True)])
def test_wide_walk_with_a_deep_stack(gpu_ready, orc, refl, lens):
    """The 4-wide walk of opaque meshes keeps the first 16 stack entries in LDS and the rest in scratch.
    100 000 large thin triangles stacked along the view axis, each cut so that it just misses the axis: a central ray touches every
    box of the hierarchy and hits nothing, so at each of the ~8 levels of the descent three siblings stay pending (19-20 entries
    by the first leaf) and every one of them is taken up again; rays off the axis hit one of the first few triangles -- often not in the
    first leaf but in a sibling waiting at the deep end of the stack -- and are reflected, by these tilted mirrors, to whichever
    wall the triangle's normal points at.  Image, hit counts and ray count equal the oracle's, scheduled and in image order.
    (Checked by dropping the entries beyond the LDS part: 277 / 519 pixels of these two frames change; with 20 000 triangles
    the stack stops at 16 by the first leaf and nothing would.)"""
    import cgraytracing_amd as cg
    rng = np.random.default_rng(4242)
    n = 100000
    z = np.sort(rng.uniform(5.0, 38.0, n))
    flip = rng.integers(0, 4, n)
    a = np.stack([np.full(n, -2.0), np.full(n, -2.0), z], 1)
    b = np.stack([np.full(n, 2.0), np.full(n, -2.0), z + rng.uniform(-0.4, 0.4, n)], 1)  # tilted: as mirrors they send the ray
    c = np.stack([np.full(n, -2.0), 1.9 - rng.uniform(0, 0.05, n), z + rng.uniform(-0.4, 0.4, n)], 1)  # to differently coloured walls
    tri = np.stack([a, b, c], 1)
    for k in range(1, 4):  # the cut corner in any of the four quadrants
        m = flip == k
        if k & 1:
            tri[m, :, 0] *= -1
        if k & 2:
            tri[m, :, 1] *= -1
    objs = scenes.planes() + [scenes.TriangleMesh.from_triangles(tri, (0.7, 0.8, 0.9), refl, 0.0)]
    cam = scenes.cam_dof() if lens else scenes.cam_pinhole()
    W, H, spp = 256, 192, 4
    want = BackendScene(orc, objs).trace_grid(cam, W, H, spp, 5, seed=5)
    with cg.Scene(objs) as sc:
        assert sc.wide_dump(0)[2] > 16  # the host's bound on the stack depth: this tree can need more than the LDS part
        got = sc.trace_grid_host(W, H, spp, cam, 5, 5)
        nat = sc.trace_grid_host(W, H, spp, cam, 5, 5, reorder=False)
    assert np.array_equal(got["rgb"], nat["rgb"]) and np.array_equal(got["nhit"], nat["nhit"]) and got["nrays"] == nat["nrays"]
    assert got["nrays"] == want["nrays"] and np.array_equal(got["nhit"], want["nhit"])
    assert np.array_equal(got["rgb"], to_acc32(want["acc_sum"], spp))


def test_bezier_shell_cull_changes_nothing(gpu_ready, monkeypatch):
    """Rays that enter a Bezier object's box but cannot come within the acceptance radius of its surface (piecewise cylinder
    bounds, cgrt_bezier.hpp bez_shell_maybe) skip the ten Newton solves.  No such solve could have been accepted, so the frame
    must be the one rendered with every solve run (CGRT_NO_BEZIER_CULL=1 at commit): identical bits, hit counts and ray count --
    on the C5-shaped scene (vase beside a bump floor, thin lens, reflections off the vase and onto it) and on a squat Bezier
    object near the camera seen through a pinhole."""
    import cgraytracing_amd as cg
    cases = [(scenes.scene_c5(scenes.stone_small_texture(True)), scenes.cam_dof(), 640, 480, 4),
             (scenes.planes() + [scenes.Sphere((-6.0, -14.0, 22.0), 4.0, (0.9, 0.9, 0.9), 0.8, 0.0),
                                 scenes.Bezier([(0, -3, 1.2), (0, 0.5, 3.2), (0, -0.5, 0.2), (0, 3, 2.6)], (3.0, -12.0, 18.0), (1.0, 1.0, 1.0), 0.5, 0.0)],
              scenes.cam_pinhole(), 512, 384, 4)]
    for objs, cam, W, H, spp in cases:
        with cg.Scene(objs) as sc:
            fast = sc.trace_grid_host(W, H, spp, cam, 5, 11)
        monkeypatch.setenv("CGRT_NO_BEZIER_CULL", "1")
        with cg.Scene(objs) as sc:
            full = sc.trace_grid_host(W, H, spp, cam, 5, 11)
        monkeypatch.delenv("CGRT_NO_BEZIER_CULL")
        assert fast["nrays"] == full["nrays"] and np.array_equal(fast["nhit"], full["nhit"])
        assert np.array_equal(fast["rgb"], full["rgb"])
        assert fast["nrays"] > W * H * spp * 1.01  # the object is seen and reflects


@pytest.mark.parametrize("seed", range(int(os.environ.get("CGRT_FUZZ_SEEDS_BIG", "3"))))
def test_random_large_meshes_match_oracle_exactly(gpu_ready, orc, seed):
    """Fuzz with trees of real depth: triangle soups of 3 000 to 30 000 triangles of mixed sizes (shared vertices, duplicates,
    degenerate triangles: _random_mesh) as an opaque, a mirror or a glass object -- the 4-wide walk with its stack for the opaque
    ones, the leaf-level hierarchy with the per-triangle boxes for the glass ones -- beside a second, small mesh of the other
    kind, thin lens or pinhole, scheduled launch and image order.  Accumulator, hit counts and ray count equal the oracle's."""
    import cgraytracing_amd as cg
    rng = np.random.default_rng(9000 + seed)
    ntri = int([3000, 12000, 30000][seed % 3])
    refl, transp = [(0.0, 0.0), (0.8, 0.0), (0.8, 0.5)][(seed // 3 + seed) % 3]
    big = _random_mesh(rng, ntri, rng.uniform((-6, -12, 24), (6, -2, 32)), float(rng.uniform(6.0, 12.0)), 0.05)
    # mixed sizes: shrink most triangles towards their first vertex so that boxes nest and overlap at every scale
    t = big.reshape(-1, 3, 3)
    shrink = rng.choice([1.0, 0.3, 0.05], size=len(t), p=[0.1, 0.4, 0.5])[:, None, None]
    big = (t[:, :1, :] + (t - t[:, :1, :]) * shrink).reshape(-1, 9)
    small = _random_mesh(rng, 60, rng.uniform((-10, -16, 22), (10, 0, 34)), 4.0, 0.1)
    objs = scenes.planes() + [scenes.TriangleMesh.from_triangles(big, tuple(rng.uniform(0.2, 1.0, 3)), refl, transp),
                              scenes.TriangleMesh.from_triangles(small, (0.6, 0.7, 0.9), 0.8, 0.5 if transp == 0.0 else 0.0)]
    cam = scenes.cam_dof() if seed % 2 else scenes.cam_pinhole()
    W, H, spp = 160, 120, 4
    want = BackendScene(orc, objs).trace_grid(cam, W, H, spp, 5, seed=40 + seed)
    with cg.Scene(objs) as sc:
        got = sc.trace_grid_host(W, H, spp, cam, 5, 40 + seed)
        nat = sc.trace_grid_host(W, H, spp, cam, 5, 40 + seed, reorder=False)
    assert np.array_equal(got["rgb"], nat["rgb"]) and np.array_equal(got["nhit"], nat["nhit"]) and got["nrays"] == nat["nrays"]
    assert got["nrays"] == want["nrays"] and np.array_equal(got["nhit"], want["nhit"])
    assert np.array_equal(got["rgb"], to_acc32(want["acc_sum"], spp))

