"""Generates the golden vectors under tests/golden/ from the COMPILED, UNMODIFIED reference
(oracle/_ref/libcgrt_ref.so = ref_harness.cpp + #include "main.cpp").  Runs only in the build container,
where /root/reference exists; the .npz outputs are committed and are what pins the oracle (and through it the
HIP path) to the reference everywhere else.

    python tests/golden/make_golden.py            # everything
    python tests/golden/make_golden.py loader t1  # (internal) one reference file-load per process, quirk Q9

Contents of each trace fixture: acc_sum [H,W,3] f64 (sum of hp.f per pixel, emission order), nhit [H,W],
nrays, and the emission-ordered hitpoint stream hp [n,9] = f, pos, normal with hp_pix / hp_smp.
"""
import hashlib
import json
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
ASSETS = os.path.join(HERE, "assets")

import scenes  # noqa: E402
from backends import Backend, BackendScene, lens_samples  # noqa: E402
from cgraytracing_amd.scene import TriangleMesh  # noqa: E402


def trace_cases():
    S = scenes
    bump_scene = lambda: S.planes(S.stone_small_texture(True)) + [S.Sphere((5, -12, 30), 5, (1, 1, 1), 0.8, 0.5)]
    chess_bump = lambda: S.planes(S.chessboard_texture(True))
    return [
        # name, objs factory, camera factory, W, H, spp, depth
        ("c1_64", S.scene_c1, S.cam_pinhole, 64, 64, 1, 1),
        ("c2_96x54_spp1", S.scene_c2, S.cam_pinhole, 96, 54, 1, 5),
        ("c2_96x54_dof4", S.scene_c2, S.cam_dof, 96, 54, 4, 5),
        ("pyramid_64", lambda: S.scene_pyramid(False), S.cam_pinhole, 64, 64, 1, 5),
        ("pyramid_glass_64", lambda: S.scene_pyramid(True), S.cam_pinhole, 64, 64, 1, 5),
        ("bunny_glass_chess_64", lambda: S.scene_c3(True), S.cam_pinhole, 64, 64, 1, 5),
        ("bunny_glass_chess_dof_48", lambda: S.scene_c3(True), S.cam_dof, 48, 48, 4, 5),
        ("dragon_64", S.scene_dragon, S.cam_pinhole, 64, 64, 1, 5),
        ("stone_bump_64x48", bump_scene, S.cam_pinhole, 64, 48, 1, 5),
        ("chess_bump_48x36", chess_bump, S.cam_pinhole, 48, 36, 1, 5),
        # round 2: Texture::color in all three orientations under mirror / glass reflections; the reference's real floor
        # texture at full size (texture/stone.jpg, 146 744 bump triangles); a refracting bump floor (tree path)
        ("textured_walls_64x48", S.scene_textured_walls, S.cam_pinhole, 64, 48, 1, 5),
        ("textured_walls_dof_48x36", S.scene_textured_walls, S.cam_dof, 48, 36, 2, 5),
        ("stone_full_bump_64x48", lambda: S.planes(S.stone_texture(True)) + [S.Sphere((5, -12, 30), 5, (1, 1, 1), 0.8, 0.5)],
         S.cam_pinhole, 64, 48, 1, 5),
        ("glass_bump_floor_48x36", S.scene_glass_bump_floor, S.cam_pinhole, 48, 36, 1, 5),
    ]


def photon_cases():
    S = scenes
    bump = lambda: S.planes(S.stone_small_texture(True)) + [S.Sphere((5, -12, 30), 5, (1, 1, 1), 0.8, 0.5)]
    return [
        # name, objs, camera, W, H, spp, photons
        ("c2_48x36", S.scene_c2, S.cam_pinhole, 48, 36, 1, 20000),
        ("c2_dof_32x24", S.scene_c2, S.cam_dof, 32, 24, 2, 20000),
        ("bunny_40", lambda: S.scene_c3(True), S.cam_pinhole, 40, 40, 1, 15000),
        ("bump_40x30", bump, S.cam_pinhole, 40, 30, 1, 15000),
    ]


def photon_cases_bezier():
    """The Bezier vase (mirror-like, main.cpp:371-376) hidden from the camera behind a diffuse sphere: no eye ray's
    nearest hit is on the vase, so the eye pass does not depend on Newton draws, while photons from the ceiling
    light do reach it and take their Newton starts from the photon's own sequential stream, exactly like the
    reference's rand().  Pins the oracle's photon-pass Bezier bit for bit against the compiled reference."""
    S = scenes
    hidden = lambda: S.planes() + [S.Sphere((7.5, -5.0, 12.5), 7.5, (0.3, 0.3, 0.3), 0.0, 0.0), S.vase_bezier()]
    return [("vase_hidden_40x30", hidden, S.cam_pinhole, 40, 30, 1, 6000)]


def tonemap_input():
    """Pixel values for the tone-map fixture: linear and log-spaced radiances, exact 0, saturation, and a dense sweep
    across every 8-bit output step."""
    rng = np.random.default_rng(5)
    return np.concatenate([rng.random((20, 31, 3)) * 3, 10 ** rng.uniform(-6, 2, (20, 31, 3)), np.zeros((1, 31, 3)),
                           np.full((1, 31, 3), 50.0), np.linspace(0, 8, 3 * 31 * 60).reshape(60, 31, 3)])


def fingerprint(nodes, leaf):
    sizes = nodes[:, 2]
    leaves = sizes[sizes < 10]
    return dict(nnodes=int(len(nodes)), nleaves=int(len(leaves)), leaf_min=int(leaves.min()), leaf_max=int(leaves.max()),
                leaf_sha256=hashlib.sha256(np.ascontiguousarray(leaf, np.int32).tobytes()).hexdigest(),
                node_sha256=hashlib.sha256(np.ascontiguousarray(nodes, np.int32).tobytes()).hexdigest())


def write_test_meshes():
    """Small mesh files in the reference's three formats, authored here (octahedron / grid patch)."""
    v = np.array([[1, 0, 0], [-1, 0, 0], [0, 1.25, 0], [0, -1.5, 0], [0, 0, 1], [0, 0, -0.75]], float)
    f = [(0, 2, 4), (2, 1, 4), (1, 3, 4), (3, 0, 4), (2, 0, 5), (1, 2, 5), (3, 1, 5), (0, 3, 5)]
    # more faces so that the tree has inner nodes: a 4x4 grid patch
    gv, gf = [], []
    for i in range(5):
        for j in range(5):
            gv.append([i * 0.5 - 1, 0.1 * ((i * 7 + j * 3) % 5), j * 0.5 - 1])
    for i in range(4):
        for j in range(4):
            a, b, c, d = i * 5 + j, i * 5 + j + 1, (i + 1) * 5 + j, (i + 1) * 5 + j + 1
            gf += [(a, b, c), (d, b, c)]
    gv = np.array(gv, float)
    allv = np.vstack([v, gv + np.array([0, -2, 0])])
    allf = f + [(a + 6, b + 6, c + 6) for a, b, c in gf]
    with open(os.path.join(ASSETS, "mesh_t0.txt"), "w") as fh:
        for a, b, c in allf:
            fh.write("begin\n")
            for k in (a, b, c):
                fh.write("vertex %.6f %.6f %.6f\n" % tuple(allv[k]))
            fh.write("end\n\n")
    with open(os.path.join(ASSETS, "mesh_t1.txt"), "w") as fh:
        fh.write("%d\n" % len(allv))
        for p in allv:
            fh.write("v  %.4f %.4f %.4f\n" % tuple(p))
        fh.write("%d\n" % len(allf))
        for a, b, c in allf:
            fh.write("f %d %d %d \n" % (a + 1, b + 1, c + 1))
    with open(os.path.join(ASSETS, "mesh_t2.txt"), "w") as fh:
        fh.write("%d\n" % len(allv))
        for p in allv:
            fh.write("v %.6f %.6f %.6f\n" % tuple(p))
        fh.write("%d\n" % len(allf))
        for a, b, c in allf:
            fh.write("f %d/%d/%d %d/%d/%d %d/%d/%d\n" % (a + 1, a + 1, a + 1, b + 1, b + 1, b + 1, c + 1, c + 1, c + 1))


LOADER_CASES = {  # name: (file, a, b, typeofdata)
    "t0": ("mesh_t0.txt", 3.0, (1.0, -4.0, 30.0), 0),
    "t1": ("mesh_t1.txt", 2.5, (-3.0, -6.0, 28.0), 1),
    "t2": ("mesh_t2.txt", 4.0, (0.0, -8.0, 35.0), 2),
}


def loader_child(name):
    """One reference file-load per process (freopen(stdin), Q9); dumps triangles + tree + a small trace."""
    file, a, b, typ = LOADER_CASES[name]
    ref = Backend("ref")
    m = TriangleMesh(os.path.join(ASSETS, file), a, b, (0.6, 0.7, 0.9), 0.8, 0.5, typ)
    s = BackendScene(ref, scenes.planes() + [m])
    tris = s.mesh_tris(0)
    nodes, leaf, bbox = s.tree_dump(0, 0)
    r = s.trace_grid(scenes.cam_pinhole(), 48, 48, 1, 5, capture=True)
    np.savez_compressed(os.path.join(HERE, "loader_%s.npz" % name), tris=tris, nodes=nodes, leaf=leaf, bbox=bbox,
                        acc_sum=r["acc_sum"], nhit=r["nhit"], nrays=r["nrays"], hp=r["hp"], hp_pix=r["hp_pix"])


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "loader":
        return loader_child(sys.argv[2])
    ref = Backend("ref")
    meta = {}
    # `only NAME...`: (re)generate just these trace fixtures and merge their entries into trace_meta.json
    only = set(sys.argv[2:]) if len(sys.argv) > 2 and sys.argv[1] == "only" else None
    if only:
        meta = json.load(open(os.path.join(HERE, "trace_meta.json")))
    # ---- trace-level fixtures ----
    for name, mk, cam, W, H, spp, depth in trace_cases():
        if only and name not in only:
            continue
        s = BackendScene(ref, mk())
        r = s.trace_grid(cam(), W, H, spp, depth, seed=12345, capture=True)
        assert r["nhp"] == len(r["hp"])
        np.savez_compressed(os.path.join(HERE, "trace_%s.npz" % name), acc_sum=r["acc_sum"], nhit=r["nhit"],
                            nrays=np.int64(r["nrays"]), hp=r["hp"], hp_pix=r["hp_pix"].astype(np.int32),
                            hp_smp=r["hp_smp"].astype(np.int32))
        meta[name] = dict(W=W, H=H, spp=spp, depth=depth, nrays=r["nrays"], nhp=r["nhp"])
        if s.n_mesh:
            nodes, leaf, _ = s.tree_dump(0, 0)
            meta[name]["mesh_tree"] = fingerprint(nodes, leaf)
        for p in range(s.n_plane):
            if len(s.bump_tris(p)):
                nodes, leaf, _ = s.tree_dump(1, p)
                meta[name]["bump_tree"] = fingerprint(nodes, leaf)
                meta[name]["bump_tris_sha256"] = hashlib.sha256(s.bump_tris(p).tobytes()).hexdigest()
        print(name, meta[name])
    if only:
        json.dump(meta, open(os.path.join(HERE, "trace_meta.json"), "w"), indent=1, sort_keys=True)
        return
    # ---- lens sampler (the reference's uniform_sampling_circle on the keyed stream) ----
    rng = np.random.default_rng(5)
    pix = rng.integers(0, 1920 * 1080, 512)
    smp = rng.integers(0, 1024, 512).astype(np.int32)
    np.savez_compressed(os.path.join(HERE, "lens_samples.npz"), seed=np.uint64(12345), pix=pix, smp=smp,
                        out=lens_samples(ref, 12345, pix, smp, 1.5))
    # ---- function-level probes ----
    fl = {}
    cam = np.array([0, 0, -10.0])

    def rays(n, lo, hi, seed):
        r = np.random.default_rng(seed)
        org = np.tile(cam, (n, 1)) + r.normal(size=(n, 3)) * 0.7
        tgt = lo + r.random((n, 3)) * (np.asarray(hi, float) - np.asarray(lo, float))
        d = tgt - org
        d /= np.linalg.norm(d, axis=1)[:, None]
        return org, d

    # Bezier::intersect with keyed draws (bit-reproducible on the same libm)
    vase = scenes.vase_bezier()
    s = BackendScene(ref, [vase])
    org, d = rays(1024, np.array([9, -22, 29.0]), np.array([21, 2, 41.0]), 11)
    keys = np.random.default_rng(12).integers(0, 2 ** 63, 1024, dtype=np.uint64)
    h, l, n = s.intersect_batch(0, org, d, keys)
    fl.update(bez_org=org, bez_dir=d, bez_keys=keys, bez_hit=h, bez_len=l, bez_n=n)
    # near-singular Bezier: exercises the jitter branch and pins g++'s argument evaluation order (bezier.h:183)
    zs = 2e-5
    thin = scenes.Bezier([(0, -5, zs), (0, 0, 2 * zs), (0, 5, zs)], (5, -5, 25), (1, 1, 1), 0.5, 0)
    s2 = BackendScene(ref, [thin])
    r = np.random.default_rng(1)
    org2 = np.tile(cam, (2000, 1)) + r.normal(size=(2000, 3)) * 0.5
    tgt = np.array([5, -5, 25.0]) + np.stack([(r.random(2000) - 0.5) * 2 * zs, (r.random(2000) - 0.5) * 9,
                                              (r.random(2000) - 0.5) * 2 * zs], 1)
    d2 = tgt - org2
    d2 /= np.linalg.norm(d2, axis=1)[:, None]
    keys2 = r.integers(0, 2 ** 63, 2000, dtype=np.uint64)
    h2, l2, n2 = s2.intersect_batch(0, org2, d2, keys2)
    fl.update(thin_org=org2, thin_dir=d2, thin_keys=keys2, thin_hit=h2, thin_len=l2, thin_n=n2)
    # sphere / plane / mesh intersect() and Texture::color
    objs = scenes.scene_c3(True)
    s3 = BackendScene(ref, objs)
    org3, d3 = rays(2048, np.array([-12, -20, 30.0]), np.array([12, -2, 50.0]), 21)
    h3, l3, n3 = s3.intersect_batch(5, org3, d3)  # the bunny
    fl.update(mesh_org=org3, mesh_dir=d3, mesh_hit=h3, mesh_len=l3, mesh_n=n3)
    h4, l4, n4 = s3.intersect_batch(0, org3, d3)  # floor plane
    P = org3 + d3 * l4[:, None]
    fl.update(floor_hit=h4, floor_len=l4, floor_col=s3.surface_color_batch(0, P))
    s4 = BackendScene(ref, scenes.scene_c2())
    org5, d5 = rays(2048, np.array([-20, -20, 20.0]), np.array([20, 0, 40.0]), 31)
    for k, ob in (("wall", 0), ("mirror", 6), ("glass", 7)):
        hh, ll, nn = s4.intersect_batch(ob, org5, d5)
        fl.update({k + "_hit": hh, k + "_len": ll, k + "_n": nn})
    fl.update(sph_org=org5, sph_dir=d5)
    # textures in the other two orientations (texture.h:41-70)
    chess = scenes.load_asset("chessboard_rgb.npz")["rgb"]
    from cgraytracing_amd.scene import Plane, Texture
    back = Plane((0, 0, 40), (0, 0, -1), (0.15, 0.15, 0.15), 0, 0, Texture(chess, (0, 0, -1), (-10, -10, 40), 20, 10))
    side = Plane((20, 0, 0), (-1, 0, 0), (0.15, 0.5, 0.15), 0, 0, Texture(chess, (-1, 0, 0), (20, -10, 10), 20, 25))
    s5 = BackendScene(ref, [back, side])
    r = np.random.default_rng(41)
    pb = np.stack([r.random(1024) * 30 - 15, r.random(1024) * 20 - 12, np.full(1024, 40.0)], 1)
    ps = np.stack([np.full(1024, 20.0), r.random(1024) * 30 - 15, r.random(1024) * 40 + 5], 1)
    fl.update(back_pts=pb, back_col=s5.surface_color_batch(0, pb), side_pts=ps, side_col=s5.surface_color_batch(1, ps))
    np.savez_compressed(os.path.join(HERE, "function_level.npz"), **fl)
    # ---- Bezier trace-level (statistical: the reference's in-trace draws are not path-keyed) ----
    s6 = BackendScene(ref, scenes.scene_c5())
    r6 = s6.trace_grid(scenes.cam_pinhole(), 48, 48, 1, 5, seed=12345)
    np.savez_compressed(os.path.join(HERE, "trace_bezier_vase_48_statistical.npz"), acc_sum=r6["acc_sum"],
                        nhit=r6["nhit"], nrays=np.int64(r6["nrays"]))
    # ---- photon pass (row f1), serial keyed semantics: per-hitpoint (f,pos,normal,flux,r2,n) and the gathered image ----
    for name, mk, cam, W, H, spp, nph in photon_cases() + photon_cases_bezier():
        s7 = BackendScene(ref, mk())
        r7 = s7.ppm(cam(), W, H, spp, 5, nphotons=nph)
        np.savez_compressed(os.path.join(HERE, "ppm_%s.npz" % name), hp=r7["hp"], image=r7["image"])
        print("ppm", name, r7["n"], float(r7["hp"][:, 15].max()))
    # ---- tone map + flip (row f2): the reference's own gammaCorr and PNG pixel loop ----
    timg = tonemap_input()
    np.savez_compressed(os.path.join(HERE, "tonemap.npz"), image=timg, rgb8=ref.tonemap(timg))
    # ---- loaders: one process per file ----
    write_test_meshes()
    for name in LOADER_CASES:
        subprocess.check_call([sys.executable, os.path.abspath(__file__), "loader", name], stdout=subprocess.DEVNULL)
    json.dump(meta, open(os.path.join(HERE, "trace_meta.json"), "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
