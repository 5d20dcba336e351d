"""GPU: BASELINE.json's full-size configurations, checked through size-independent properties (the oracle
cannot render them in test time): sharding invariance, sample-range additivity, run-to-run determinism,
oracle agreement on crops, and counter identities.  Plus the edge cases: empty scene, 1x1 image, widest row,
the 96-object limit."""
import os

import numpy as np
import pytest
import torch

import scenes
from backends import BackendScene, to_acc32
from cgraytracing_amd.dist import assemble, local_rows

pytestmark = pytest.mark.gpu


def _frame(sc, W, H, spp, cam, depth=5, seed=12345, **kw):
    out, nhit, cnt = sc.trace_grid(W, H, spp, cam, depth, seed, **kw)
    torch.cuda.synchronize()
    return out, nhit, cnt


def test_c2_full_size_properties(gpu_ready, orc):
    """configs[1]: 1920x1080, spp 64, spheres + mirror + glass, depth 5, thin lens."""
    import cgraytracing_amd as cg
    W, H, spp = 1920, 1080, 64
    cam = scenes.cam_dof()
    sc = cg.Scene(scenes.scene_c2())
    full, nhit, cnt = _frame(sc, W, H, spp, cam)
    rays, hps = int(cnt[0]), int(cnt[1])
    # counters: every sample starts one ray; a ray tree has at most 31 rays / 16 hitpoints
    assert W * H * spp <= rays <= 31 * W * H * spp
    assert hps == int(nhit.view(torch.int32).to(torch.int64).sum())
    assert 1.5 < rays / (W * H * spp) < 1.6  # SURVEY 8d measured 1.552 rays per pixel-sample on this scene
    # determinism
    again, _, cnt2 = _frame(sc, W, H, spp, cam)
    assert torch.equal(full, again) and int(cnt2[0]) == rays
    # sharding invariance: 8 ranks x 8-row block-cyclic stripes re-assemble to the same bits
    N, S = 8, 8
    parts, rsum = [], 0
    for r in range(N):
        rows = local_rows(H, S, r, N)
        p, _, c = _frame(sc, W, H, spp, cam, rows=rows, stripe=(S, r, N))
        parts.append(p)
        rsum += int(c[0])
    assert torch.equal(assemble(torch.stack(parts), H, S, N), full)
    assert rsum == rays
    # sample-range additivity (two passes of 32 samples, normalised by 64)
    a, _, _ = _frame(sc, W, H, 32, cam, sample_offset=0, spp_total=64)
    b, _, _ = _frame(sc, W, H, 32, cam, sample_offset=32, spp_total=64)
    assert float((a + b - full).abs().max()) < 1e-6
    # value range: colours <= 1 and adj <= 1 => a sample contributes at most 16 hitpoints of weight <= 1
    assert float(full.min()) >= 0.0 and float(full.max()) <= 16.0
    # oracle agreement on three crops of the full-size frame (rows through the glass sphere, the mirror, the top)
    o = BackendScene(orc, scenes.scene_c2())
    for r0 in (120, 300, 1000):
        want = o.trace_grid(cam, W, H, spp, 5, 12345, row0=r0, nrows=4)
        got = full[r0:r0 + 4].cpu().numpy()
        assert np.array_equal(got, to_acc32(want["acc_sum"], spp)), r0
        assert np.array_equal(nhit[r0:r0 + 4].cpu().numpy().view(np.uint32), want["nhit"])
    sc.close()


def test_c3_full_size_properties(gpu_ready, orc):
    """configs[2]: 2048x2048 spp 64, glass bunny + ChessBoard floor."""
    import cgraytracing_amd as cg
    W, H, spp = 2048, 2048, 64
    cam = scenes.cam_dof()
    sc = cg.Scene(scenes.scene_c3(True))
    full, nhit, cnt = _frame(sc, W, H, spp, cam)
    assert int(cnt[0]) >= W * H * spp
    half = [_frame(sc, W, H, spp, cam, rows=H // 2, row_offset=k * (H // 2))[0] for k in range(2)]
    assert torch.equal(torch.cat(half), full)
    o = BackendScene(orc, scenes.scene_c3(True))
    for r0 in (420, 700):  # through the bunny / its refraction of the chessboard
        want = o.trace_grid(cam, W, H, spp, 5, 12345, row0=r0, nrows=2)
        assert np.array_equal(full[r0:r0 + 2].cpu().numpy(), to_acc32(want["acc_sum"], spp)), r0
    sc.close()


def test_c4_shape_dragon_stripes(gpu_ready, orc):
    """configs[3] shape (dragon mesh, row-tiled over 8 ranks) at 4096 x 4096 with spp 8 instead of 256: the
    per-sample work is identical and the properties are independent of spp."""
    import cgraytracing_amd as cg
    W, H, spp = 4096, 4096, 8
    cam = scenes.cam_dof()
    sc = cg.Scene(scenes.scene_dragon())
    full, _, cnt = _frame(sc, W, H, spp, cam)
    assert int(cnt[0]) == W * H * spp  # everything is diffuse: exactly one ray per sample
    N, S = 8, 16
    parts = [_frame(sc, W, H, spp, cam, rows=local_rows(H, S, r, N), stripe=(S, r, N))[0] for r in range(N)]
    assert torch.equal(assemble(torch.stack(parts), H, S, N), full)
    o = BackendScene(orc, scenes.scene_dragon())
    want = o.trace_grid(cam, W, H, spp, 5, 12345, row0=900, nrows=2)  # rows through the dragon
    assert np.array_equal(full[900:902].cpu().numpy(), to_acc32(want["acc_sum"], spp))
    sc.close()


def test_c4_full_configuration(gpu_ready, orc):
    """configs[3] exactly as BASELINE.json names it on one GPU -- 4096 x 4096, spp 256, dragon, thin lens (the frame
    `bench.py --config c4` times; cost-scheduled: probe, ~18 000 heavy wave tiles through the unit queue and the ordered
    sum, light tiles on the second stream): ray count = one per sample, and rows through the dragon's head, body and the
    floor in front of it equal the oracle's at the full 256 samples, bit for bit, as does one of the 8 ranks' stripes."""
    import cgraytracing_amd as cg
    W, H, spp = 4096, 4096, 256
    cam = scenes.cam_dof()
    with cg.Scene(scenes.scene_dragon()) as sc:
        full, nhit, cnt = _frame(sc, W, H, spp, cam)
        assert int(cnt[0]) == W * H * spp and int(cnt[1]) == int(nhit.view(torch.int32).to(torch.int64).sum())
        o = BackendScene(orc, scenes.scene_dragon())
        for r0 in (700, 1100, 1500):
            want = o.trace_grid(cam, W, H, spp, 5, 12345, row0=r0, nrows=1)
            assert np.array_equal(full[r0:r0 + 1].cpu().numpy(), to_acc32(want["acc_sum"], spp)), r0
        N, S, r = 8, 16, 3
        part, _, _ = _frame(sc, W, H, spp, cam, rows=local_rows(H, S, r, N), stripe=(S, r, N))
        rows = [((j // S) * N + r) * S + j % S for j in range(local_rows(H, S, r, N))]
        assert torch.equal(part, full[torch.as_tensor(rows, device=full.device)])


def test_c5_shape_bump_and_bezier(gpu_ready, orc):
    """configs[4] shape: 8192-wide rows, stone-sized bump floor (146 744 triangles) + Bezier vase, one GPU's
    share reduced to 8192 x 64 rows and spp 4."""
    import cgraytracing_amd as cg
    tex = scenes.stone_texture()
    objs = scenes.scene_c5(tex)
    W, H, spp = 8192, 8192, 4
    cam = scenes.cam_dof()
    sc = cg.Scene(objs)
    st = sc.stats()
    assert st["n_triangles"] == 146744 and st["n_nodes"] == 32767  # stone.jpg-sized bump mesh
    r0 = 1600
    got, _, cnt = _frame(sc, W, H, spp, cam, rows=64, row_offset=r0)
    again, _, _ = _frame(sc, W, H, spp, cam, rows=64, row_offset=r0)
    assert torch.equal(got, again)
    o = BackendScene(orc, objs)
    want = o.trace_grid(cam, W, H, spp, 5, 12345, row0=r0, nrows=2)
    from test_gpu_parity import BEZ_SCENE_BAR, bezier_report
    frac, linf, gap = bezier_report("c5_shape_8192x2_spp4", got[:2].cpu().numpy(), to_acc32(want["acc_sum"], spp), 0, 0)
    assert frac >= max(0.999, BEZ_SCENE_BAR[0]), frac
    sc.close()


def test_c5_full_sample_count_stripe(gpu_ready):
    """configs[4] at its FULL sample count on one 16-row stripe of the 8192 x 8192 frame through the Bezier vase: spp 1024,
    the reference's real stone.jpg bump floor (146 744 triangles), thin lens.  At this sample count a heavy tile is 65 536
    units in 256 items; the scheduled launch (vase tiles through the unit queue, values parked and summed in order) must
    equal the image-order launch -- one lane per pixel running its 1024 samples in sequence -- bit for bit, in image and ray
    count.  (The oracle runs one row per thread: a single 8192-pixel row at 1024 samples is ~3 minutes of CPU, so agreement
    with the oracle on this scene is checked at 4 samples in test_c5_shape_bump_and_bezier; Bezier parity is statistical.)"""
    import cgraytracing_amd as cg
    W, H, spp, r0 = 8192, 8192, 1024, 3968
    cam = scenes.cam_dof()
    objs = scenes.scene_c5(scenes.stone_texture())
    with cg.Scene(objs) as sc:
        got, _, cnt = _frame(sc, W, H, spp, cam, rows=16, row_offset=r0)
        nat = sc.trace_grid_host(W, H, spp, cam, 5, 12345, rows=16, row_offset=r0, reorder=False)
    assert np.array_equal(got.cpu().numpy(), nat["rgb"]) and int(cnt[0]) == nat["nrays"]
    assert int(cnt[0]) > W * 16 * spp  # the vase reflects: secondary rays


def test_c5_full_share_properties(gpu_ready):
    """configs[4] at its per-GPU size, once: share 0 of 8 (16-row block-cyclic stripes) of the 8192 x 8192 frame = 8192 x 1024
    rows at the FULL 1024 samples -- the frame `bench.py --config c5` times.  Size-independent properties: the same call
    twice gives the same bits; Sigma nhit = the hitpoint counter; at least one ray per pixel-sample; and on four sampled
    stripes (through the vase, beside it, floor only, near the ceiling) the scheduled launch -- vase tiles through the unit
    queue, their values parked and summed in order, light tiles on the second stream -- equals the image-order launch of the
    same stripe bit for bit, rays included."""
    import cgraytracing_amd as cg
    W, H, spp, S, N = 8192, 8192, 1024, 16, 8
    cam = scenes.cam_dof()
    rows = local_rows(H, S, 0, N)
    assert rows == 1024
    with cg.Scene(scenes.scene_c5(scenes.stone_texture())) as sc:
        a, nhit, cnt = _frame(sc, W, H, spp, cam, rows=rows, stripe=(S, 0, N))
        rays, hps = int(cnt[0]), int(cnt[1])
        assert hps == int(nhit.view(torch.int32).to(torch.int64).sum())
        assert rays >= W * rows * spp
        b, nhit2, cnt2 = _frame(sc, W, H, spp, cam, rows=rows, stripe=(S, 0, N))
        assert torch.equal(a, b) and torch.equal(nhit, nhit2) and int(cnt2[0]) == rays and int(cnt2[1]) == hps
        del b, nhit2
        for k in (31, 20, 5, 60):  # local stripe k of share 0 = global rows [k * N * S, k * N * S + S)
            r0 = k * N * S
            nat = sc.trace_grid_host(W, H, spp, cam, 5, 12345, rows=S, row_offset=r0, reorder=False)
            assert np.array_equal(a[k * S:(k + 1) * S].cpu().numpy(), nat["rgb"]), "stripe at global row %d" % r0
            assert np.array_equal(nhit[k * S:(k + 1) * S].cpu().numpy().view(np.uint32), nat["nhit"])
    print("c5 share: %d rays, %d hitpoints (%.4f rays per pixel-sample)" % (rays, hps, rays / (W * rows * spp)))


def test_edge_cases(gpu_ready, orc):
    import cgraytracing_amd as cg
    cam = scenes.cam_dof()
    # empty scene: every ray misses (main.cpp:64-66)
    sc = cg.Scene([])
    r = sc.trace_grid_host(33, 17, 3, cam, 5, 1)
    assert r["nrays"] == 33 * 17 * 3 and r["nhp"] == 0 and not r["rgb"].any()
    sc.close()
    # 1x1 image, one sample; and a single very wide row
    sc = cg.Scene(scenes.scene_c2())
    o = BackendScene(orc, scenes.scene_c2())
    for W, H, spp in [(1, 1, 1), (8191, 1, 2), (1, 257, 2)]:
        got = sc.trace_grid_host(W, H, spp, cam, 5, 3)
        want = o.trace_grid(cam, W, H, spp, 5, 3)
        assert got["nrays"] == want["nrays"]
        assert np.array_equal(got["rgb"], to_acc32(want["acc_sum"], spp)), (W, H)
    sc.close()
    # every depth budget
    sc = cg.Scene(scenes.scene_c2())
    for depth in (1, 2, 3, 4, 5):
        got = sc.trace_grid_host(64, 36, 2, cam, depth, 5)
        want = o.trace_grid(cam, 64, 36, 2, depth, 5)
        assert got["nrays"] == want["nrays"] and np.array_equal(got["rgb"], to_acc32(want["acc_sum"], 2)), depth
    sc.close()
    # 96 top-level objects (round 2's limit, the whole list in LDS), mixed materials, ties between coincident spheres
    rng = np.random.default_rng(0)
    objs = scenes.wall_spheres()
    while len(objs) < 96:
        c = (rng.uniform(-15, 15), rng.uniform(-18, 15), rng.uniform(15, 38))
        refl, transp = [(0, 0), (0.8, 0), (0.8, 0.5)][rng.integers(0, 3)]
        s = scenes.Sphere(c, rng.uniform(0.5, 3), tuple(rng.uniform(0.2, 1, 3)), refl, transp)
        objs.append(s)
        if len(objs) < 96 and rng.random() < 0.2:  # an exact duplicate: the earlier object must win (main.cpp:57)
            objs.append(scenes.Sphere(c, s.radius, (1.0, 0.0, 1.0), 0, 0))
    sc = cg.Scene(objs)
    got = sc.trace_grid_host(160, 120, 2, cam, 5, 9)
    want = BackendScene(orc, objs).trace_grid(cam, 160, 120, 2, 5, 9)
    assert got["nrays"] == want["nrays"] and np.array_equal(got["rgb"], to_acc32(want["acc_sum"], 2))
    sc.close()


def _many_spheres(n, seed):
    rng = np.random.default_rng(seed)
    objs = scenes.wall_spheres()
    while len(objs) < n:
        c = (rng.uniform(-17, 17), rng.uniform(-18, 17), rng.uniform(12, 38))
        refl, transp = [(0, 0), (0, 0), (0.8, 0), (0.8, 0.5)][rng.integers(0, 4)]
        s = scenes.Sphere(c, rng.uniform(0.3, 1.6), tuple(rng.uniform(0.2, 1, 3)), refl, transp)
        objs.append(s)
        if len(objs) < n and rng.random() < 0.05:  # an exact duplicate later in the list: the earlier object wins (main.cpp:57)
            objs.append(scenes.Sphere(c, s.radius, (1.0, 0.0, 1.0), 0, 0))
    return objs


def test_a_thousand_objects(gpu_ready, orc, monkeypatch):
    """`vector<Object*> objs` is unbounded (main.cpp:277).  1 000 spheres: 768 of them live in LDS (kLdsObjsMax), the other 232
    are read through the scalar cache; the oracle's frame, exactly -- mirror, glass and duplicated spheres included, with the
    duplicates on either side of the LDS boundary.  Then a scene of every object kind with only 7 of its 40 objects resident
    (CGRT_LDS_OBJS): the general loop's staging record, eye pass and photon pass."""
    import cgraytracing_amd as cg
    cam = scenes.cam_dof()
    objs = _many_spheres(1000, 11)
    with cg.Scene(objs) as sc:
        got = sc.trace_grid_host(160, 96, 2, cam, 5, 9)
    want = BackendScene(orc, objs).trace_grid(cam, 160, 96, 2, 5, 9)
    assert got["nrays"] == want["nrays"] and np.array_equal(got["nhit"], want["nhit"])
    assert np.array_equal(got["rgb"], to_acc32(want["acc_sum"], 2))
    # general loop: planes (one bump-mapped), spheres of every material, a glass mesh, an opaque mesh -- 40 objects, 7 resident
    objs = scenes.planes(scenes.stone_small_texture(True)) + _many_spheres(38, 3)[5:]
    objs.insert(9, scenes.TriangleMesh.from_triangles(scenes.pyramid_tris(0.6, (-6.0, -13.0, 36.0)), (0.6, 0.7, 0.9), 0.0, 0.0))
    objs.append(scenes.TriangleMesh.from_triangles(scenes.bunny_tris() * 0.5 + np.tile([6.0, -6.0, 12.0], 3), (1.0, 1.0, 1.0), 0.8, 0.5))
    assert len(objs) == 40
    want = BackendScene(orc, objs).trace_grid(cam, 128, 96, 2, 5, 4)
    wantp = BackendScene(orc, objs).ppm(scenes.cam_pinhole(), 48, 36, 1, 5, nphotons=3000)
    ph_images = []
    for resident in ("7", None):
        if resident:
            monkeypatch.setenv("CGRT_LDS_OBJS", resident)
        else:
            monkeypatch.delenv("CGRT_LDS_OBJS")
        with cg.Scene(objs) as sc:
            got = sc.trace_grid_host(128, 96, 2, cam, 5, 4)
            gotn = sc.trace_grid_host(128, 96, 2, cam, 5, 4, reorder=False)
            ph = sc.ppm_render(48, 36, 1, scenes.cam_pinhole(), 5, 12345, nphotons=3000)
        for g in (got, gotn):
            assert g["nrays"] == want["nrays"] and np.array_equal(g["nhit"], want["nhit"]), resident
            assert np.array_equal(g["rgb"], to_acc32(want["acc_sum"], 2)), resident
        nd = int((ph["image"] != wantp["image"]).any(axis=2).sum())
        print("resident=%s: photon image pixels differing from the oracle: %d (max %g), hitpoints %d vs %d" %
              (resident, nd, np.abs(ph["image"] - wantp["image"]).max(), ph["count"], wantp["n"]))
        ph_images.append(ph["image"])
    assert np.array_equal(ph_images[0], ph_images[1])
    assert nd == 0


def test_reference_main_configuration_properties(gpu_ready, orc):
    """Rows f1/f2 at the reference's committed size (main.cpp:28-29,292,320,348-353: 1024x768, spp 1, planes + stone-sized
    bump floor + diffuse dragon) with 2 M of its 20.48 M photons: size-independent properties -- one hitpoint per pixel
    (every primary ray ends on a diffuse surface of the closed box), five diffuse hits for all but a handful of photons, a row band rendered
    on its own equals the same rows of the full frame bit for bit, the tone-mapped bytes are the reference gammaCorr of
    the gathered image, and the first 20 000 photons alone give the oracle's serial image bit for bit."""
    import cgraytracing_amd as cg
    tex = scenes.stone_texture()
    objs = scenes.planes(tex) + [scenes.TriangleMesh.from_triangles(scenes.dragon_tris(), (0.25, 0.25, 0.5), 0.0, 0.0, 1)]
    W, H, nph = 1024, 768, 2000000
    cam = scenes.cam_pinhole()
    with cg.Scene(objs) as sc:
        full = sc.ppm_render(W, H, 1, cam, 5, 12345, nphotons=nph, want_rgb8=True)
        band = sc.ppm_render(W, H, 1, cam, 5, 12345, nphotons=nph, rows=64, row_offset=200)
        small = sc.ppm_render(W, H, 1, cam, 5, 12345, nphotons=20000)
    assert full["count"] == W * H
    assert 5 * nph - 1000 < full["n_events"] <= 5 * nph  # a handful of photons leave through a seam of the box
    assert np.array_equal(band["image"], full["image"][200:264])
    assert np.array_equal(full["rgb8"], orc.tonemap(full["image"]))
    assert 0.05 < full["image"].mean() < 1.0 and full["rgb8"][:40].max() == 255  # the ceiling light saturates
    want = BackendScene(orc, objs).ppm(cam, W, H, 1, 5, nphotons=20000)
    assert np.array_equal(small["image"], want["image"])
