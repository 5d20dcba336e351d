"""GPU: row f1 -- eye pass + photon pass + final gather (cgrt_ppm_render) against golden vectors from the compiled
reference run serially on the photons' keyed streams (tests/golden/ppm_*.npz), and against the oracle."""
import os
import sys

import numpy as np
import pytest

import scenes
from backends import BackendScene

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
sys.path.insert(0, GOLD)
import make_golden  # noqa: E402


def _canon(hp, spp):
    ps = hp[:, 0].astype(np.int64)
    pix, smp = ps // spp, ps % spp
    out = np.concatenate([pix[:, None].astype(np.float64), smp[:, None].astype(np.float64), hp[:, 2:]], axis=1)
    order = np.lexsort([out[:, 7], out[:, 6], out[:, 5], out[:, 1], out[:, 0]])
    return out[order]


@pytest.mark.parametrize("case", make_golden.photon_cases(), ids=[c[0] for c in make_golden.photon_cases()])
def test_photon_pass_matches_reference_golden(gpu_ready, case):
    import cgraytracing_amd as cg
    name, mk, cam, W, H, spp, nph = case
    g = np.load(os.path.join(GOLD, "ppm_%s.npz" % name))
    sc = cg.Scene(mk())
    r = sc.ppm_render(W, H, spp, cam(), 5, 12345, nphotons=nph, want_hitpoints=True)
    # a different batch size must not change anything (events are replayed in photon order)
    r2 = sc.ppm_render(W, H, spp, cam(), 5, 12345, nphotons=nph, batch=3000)
    sc.close()
    got = _canon(r["hp"], spp)
    assert got.shape == g["hp"].shape
    assert np.array_equal(got[:, :11], g["hp"][:, :11]), "hitpoint geometry"
    assert np.array_equal(got[:, 15], g["hp"][:, 15]), "photon counts n"
    assert np.array_equal(got[:, 14], g["hp"][:, 14]), "radii r2"
    assert np.array_equal(got[:, 11:14], g["hp"][:, 11:14]), "flux"
    assert np.array_equal(r["image"], g["image"]), "gathered image"
    assert np.array_equal(r2["image"], r["image"])


@pytest.mark.parametrize("case", make_golden.photon_cases()[:2], ids=[c[0] for c in make_golden.photon_cases()[:2]])
def test_pair_buffer_overflow_path_matches_reference_golden(gpu_ready, case):
    """the pair buffer's overflow path.  pair_cap (a cgrt_photons field, 0 = automatic) is forced far below
    the pairs one default batch produces, so the 64-bit pair count exceeds it, the batch is redone in halves -- several
    times -- and grows back as radii shrink; the result must still be the compiled reference's, bit for bit."""
    import cgraytracing_amd as cg
    name, mk, cam, W, H, spp, nph = case
    g = np.load(os.path.join(GOLD, "ppm_%s.npz" % name))
    with cg.Scene(mk()) as sc:
        base = sc.ppm_render(W, H, spp, cam(), 5, 12345, nphotons=nph)
        assert base["n_batch_halvings"] == 0 and base["n_pairs"] > 2000
        r = sc.ppm_render(W, H, spp, cam(), 5, 12345, nphotons=nph, want_hitpoints=True, pair_cap=max(128, base["n_pairs"] // 12))
    assert r["n_batch_halvings"] >= 3, r["n_batch_halvings"]
    got = _canon(r["hp"], spp)
    assert np.array_equal(got[:, :11], g["hp"][:, :11]) and np.array_equal(got[:, 11:16], g["hp"][:, 11:16])
    assert np.array_equal(r["image"], g["image"]) and np.array_equal(base["image"], g["image"])


def test_producer_stream_overlap_is_invisible(gpu_ready, monkeypatch):
    """With more photons than one batch the next batch is traced on a second stream while the current one is searched and
    replayed (cgrt_photon.hpp PhotonProducer), enqueued BEFORE the current batch's pair count is known.  Checked here: the
    overlapped run equals the single-stream run (CGRT_PHOTON_OVERLAP=0) and the reference golden bit for bit, also when
    the pair buffer overflows so that batches enqueued ahead no longer match the plan and are produced again."""
    import cgraytracing_amd as cg
    name, mk, cam, W, H, spp, nph = make_golden.photon_cases()[0]
    g = np.load(os.path.join(GOLD, "ppm_%s.npz" % name))
    with cg.Scene(mk()) as sc:
        base = sc.ppm_render(W, H, spp, cam(), 5, 12345, nphotons=nph)
        cap = max(128, base["n_pairs"] // 12)
        on = sc.ppm_render(W, H, spp, cam(), 5, 12345, nphotons=nph, batch=nph // 7 + 1, want_hitpoints=True)
        on_ovf = sc.ppm_render(W, H, spp, cam(), 5, 12345, nphotons=nph, batch=nph // 3 + 1, pair_cap=cap, want_hitpoints=True)
        monkeypatch.setenv("CGRT_PHOTON_OVERLAP", "0")
        off = sc.ppm_render(W, H, spp, cam(), 5, 12345, nphotons=nph, batch=nph // 7 + 1, want_hitpoints=True)
    assert on_ovf["n_batch_halvings"] >= 1, on_ovf["n_batch_halvings"]
    for r in (on, on_ovf, off):
        assert np.array_equal(r["image"], g["image"])
        assert np.array_equal(r["hp"], on["hp"])
    assert on["n_events"] == off["n_events"] == base["n_events"] and on["n_pairs"] == off["n_pairs"]


def test_initial_radius_field(gpu_ready, orc):
    """cgrt_photons.initial_radius: 0 is the reference's committed 200/768 (main.cpp:84,183); the same value passed
    explicitly gives the identical image; a host mirroring a reference compiled for another `height` gets another radius,
    hence another cell length (hash.h:25-26) and another image -- checked for sanity only (no reference build with another
    height exists to pin it: parity unpinned for values other than 200/768)."""
    import cgraytracing_amd as cg
    W, H, nph = 48, 36, 20000
    with cg.Scene(scenes.scene_c2()) as sc:
        a = sc.ppm_render(W, H, 1, scenes.cam_pinhole(), 5, 12345, nphotons=nph)
        b = sc.ppm_render(W, H, 1, scenes.cam_pinhole(), 5, 12345, nphotons=nph, initial_radius=200.0 / 768)
        c = sc.ppm_render(W, H, 1, scenes.cam_pinhole(), 5, 12345, nphotons=nph, initial_radius=200.0 / 1080, want_hitpoints=True)
    g = np.load(os.path.join(GOLD, "ppm_c2_48x36.npz"))
    assert np.array_equal(a["image"], g["image"]) and np.array_equal(b["image"], a["image"])
    assert not np.array_equal(c["image"], a["image"])
    assert c["hp"][:, 14].max() <= (200.0 / 1080) ** 2 and np.isfinite(c["image"]).all() and c["image"].max() > 0


def test_photon_pass_larger_vs_oracle(gpu_ready, orc):
    import cgraytracing_amd as cg
    objs = scenes.scene_c2()
    W, H, spp, nph = 96, 72, 1, 200000
    o = BackendScene(orc, objs)
    want = o.ppm(scenes.cam_pinhole(), W, H, spp, 5, nphotons=nph)
    sc = cg.Scene(objs)
    got = sc.ppm_render(W, H, spp, scenes.cam_pinhole(), 5, 12345, nphotons=nph)
    sc.close()
    assert np.array_equal(got["image"], want["image"])
    assert want["image"].max() > 1.0


@pytest.mark.parametrize("mk", [scenes.scene_c2, lambda: scenes.scene_c3(True),
                                lambda: scenes.planes(scenes.stone_small_texture(True)) + [scenes.Sphere((5, -12, 30), 5, (1, 1, 1), 0.8, 0.5)]],
                         ids=["spheres", "glass_bunny", "bump_floor"])
def test_photon_paths_match_oracle(gpu_ready, orc, mk):
    """Function-level: every diffuse photon hit (position, normal, carried flux) of 4096 photons, including the
    repeated self-hits of photons that start 1e-3 under the ceiling and the Russian roulette at glass."""
    import cgraytracing_amd as cg
    objs = mk()
    want = BackendScene(orc, objs).photon_events(1000, 4096)
    sc = cg.Scene(objs)
    got = sc.photon_events(1000, 4096)
    sc.close()
    assert got.shape == want.shape and len(want) > 8000
    assert np.array_equal(got, want)


def _by_photon(ev):
    out = {}
    for row in ev:
        out.setdefault(int(row[0]), []).append(row[1:])
    return out


def test_photon_paths_with_bezier_vs_oracle(gpu_ready, orc):
    """Photons that meet the Bezier vase take their Newton starts from the photon's own sequential stream
    (bezier.h:236,239 on rand(); pinned on the CPU side by tests/golden/ppm_vase_hidden_40x30.npz from the compiled
    reference).  Device pow/sin/cos are not glibc's bit for bit and Newton from random starts is chaotic, so the bar
    is statistical, as for the eye pass: Bezier hit distances agree to ~1e-14, not bit for bit, so a
    photon counts as agreeing when it has the same number of diffuse hits and every {P, n, flux} is within 1e-6;
    >= 99 % of photons must agree (the rest took a different Newton root or jitter branch)."""
    import cgraytracing_amd as cg
    objs = scenes.planes() + [scenes.Sphere((7.5, -5.0, 12.5), 7.5, (0.3, 0.3, 0.3), 0.0, 0.0), scenes.vase_bezier(0.0)]
    want = _by_photon(BackendScene(orc, objs).photon_events(0, 4096))
    sc = cg.Scene(objs)
    got = _by_photon(sc.photon_events(0, 4096))
    sc.close()
    on_vase = sum(1 for v in want.values() for e in v if np.hypot(e[0] - 15, e[2] - 35) < 4.2 and -19.9 < e[1] < 0)
    assert on_vase > 300  # the vase really is hit
    same = sum(1 for k, v in want.items()
               if k in got and len(got[k]) == len(v) and np.allclose(np.asarray(got[k]), np.asarray(v), rtol=0, atol=1e-6))
    exact = sum(1 for k, v in want.items()
                if k in got and len(got[k]) == len(v) and np.array_equal(np.asarray(got[k]), np.asarray(v)))
    frac = same / len(want)
    print("photons agreeing within 1e-6: %.4f, bit-identical: %.4f (%d on-vase events)" % (frac, exact / len(want), on_vase))
    assert frac >= 0.99


def test_photon_pass_with_bezier_vs_reference_golden(gpu_ready):
    import cgraytracing_amd as cg
    name, mk, cam, W, H, spp, nph = make_golden.photon_cases_bezier()[0]
    g = np.load(os.path.join(GOLD, "ppm_%s.npz" % name))
    sc = cg.Scene(mk())
    r = sc.ppm_render(W, H, spp, cam(), 5, 12345, nphotons=nph, want_hitpoints=True)
    sc.close()
    got = _canon(r["hp"], spp)
    assert got.shape == g["hp"].shape
    assert np.array_equal(got[:, :11], g["hp"][:, :11]), "hitpoint geometry (the vase is hidden from the camera)"
    same_n = (got[:, 15] == g["hp"][:, 15]).mean()
    rel = np.abs(r["image"] - g["image"]).max() / g["image"].max()
    print("hitpoints with identical photon counts: %.4f, image max rel diff %.3e" % (same_n, rel))
    assert same_n >= 0.98 and rel < 0.05


def test_tonemap_matches_reference_golden(gpu_ready, orc):
    """Row f2 on the device: gammaCorr + flip equal the reference's own output byte for byte (device pow/exp differ
    from glibc by at most an ulp, which moves an 8-bit result only within ~1e-14 of a rounding boundary)."""
    import cgraytracing_amd as cg
    g = np.load(os.path.join(GOLD, "tonemap.npz"))
    got = cg.tonemap_rgb8(g["image"])
    assert got.shape == g["rgb8"].shape
    assert np.array_equal(got, g["rgb8"])
    rng = np.random.default_rng(11)
    big = 10 ** rng.uniform(-8, 1.5, (768, 1024, 3))  # the reference's frame size
    assert np.array_equal(cg.tonemap_rgb8(big), orc.tonemap(big))
    bad = np.array([[[np.nan, -1.0, np.inf]]])
    assert cg.tonemap_rgb8(bad).tolist() == [[[0, 0, 255]]]


def test_ppm_render_rgb8_and_png(gpu_ready, orc, tmp_path):
    """render() + main()'s PNG loop end to end: the rgb8 plane of cgrt_ppm_render is the tone-mapped, flipped image, and
    equals what the reference's gammaCorr gives for the reference's own gathered image (golden)."""
    import cgraytracing_amd as cg
    name, mk, cam, W, H, spp, nph = make_golden.photon_cases()[0]
    g = np.load(os.path.join(GOLD, "ppm_%s.npz" % name))
    sc = cg.Scene(mk())
    r = sc.ppm_render(W, H, spp, cam(), 5, 12345, nphotons=nph, want_rgb8=True)
    sc.close()
    assert np.array_equal(r["image"], g["image"])
    assert np.array_equal(r["rgb8"], orc.tonemap(g["image"]))
    assert r["rgb8"].max() > 100 and r["n_events"] > nph and r["n_pairs"] > 0
    assert all(v >= 0 for v in r["ms"].values())
    path = str(tmp_path / "test.png")
    cg.write_png(path, r["rgb8"])
    from PIL import Image
    assert np.array_equal(np.asarray(Image.open(path).convert("RGB")), r["rgb8"])


def test_cpp_host_program_full_pipeline(gpu_ready, orc, tmp_path):
    """examples/main_dropin.cpp --photons N --png: render() + main() of the reference end to end through
    include/cgrt_host.hpp (render_ppm, write_png).  The C2 scene at 48x36 with 20 000 photons is the golden case
    ppm_c2_48x36 from the compiled reference: the PNG must hold the reference's gammaCorr of the reference's image."""
    import subprocess
    from PIL import Image
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "cgraytracing_amd", "cgrt_main")
    assert os.path.exists(exe), "build it with make -C cgraytracing_amd/csrc all"
    png = str(tmp_path / "test.png")
    out = subprocess.run([exe, "--scene", "c2", "--width", "48", "--height", "36", "--photons", "20000", "--png", png],
                         capture_output=True, text=True, check=True).stdout
    g = np.load(os.path.join(GOLD, "ppm_c2_48x36.npz"))
    assert "hitpoints: %d" % len(g["hp"]) in out
    assert np.array_equal(np.asarray(Image.open(png).convert("RGB")), orc.tonemap(g["image"]))


def test_ppm_render_shards_by_rows(gpu_ready):
    """Row f1 x row e: a rank that owns only some rows (a band, or block-cyclic stripes) traces all photons but keeps
    only its own hitpoints; its rows must equal the same rows of the full-frame render bit for bit, so the assembled
    multi-GPU frame is the single-GPU frame.  Ranks are played one after another on the one GPU."""
    import torch
    import cgraytracing_amd as cg
    from cgraytracing_amd import dist as cdist
    objs = scenes.planes(scenes.stone_small_texture(True)) + [scenes.Sphere((5, -12, 30), 5, (1, 1, 1), 0.8, 0.5)]
    W, H, spp, nph = 56, 40, 2, 30000
    sc = cg.Scene(objs)
    full = sc.ppm_render(W, H, spp, scenes.cam_dof(), 5, 12345, nphotons=nph)
    band = sc.ppm_render(W, H, spp, scenes.cam_dof(), 5, 12345, nphotons=nph, rows=16, row_offset=8)
    assert np.array_equal(band["image"], full["image"][8:24])
    n, S = 3, 8
    rows_local = cdist.local_rows(H, S, 0, n)
    parts = [sc.ppm_render(W, H, spp, scenes.cam_dof(), 5, 12345, nphotons=nph, rows=rows_local, stripe=(S, r, n))["image"]
             for r in range(n)]
    sc.close()
    frame = cdist.assemble(torch.from_numpy(np.stack(parts)), H, S, n).numpy()
    assert np.array_equal(frame, full["image"])
    assert full["image"].max() > 0.5


@pytest.mark.parametrize("seed", range(int(os.environ.get("CGRT_FUZZ_SEEDS", "3"))))
def test_photon_pass_random_scenes_vs_oracle(gpu_ready, orc, seed):
    """Fuzz for row f1: random triangle soups (opaque / mirror / glass, with duplicate and degenerate triangles) and
    spheres in the box, eye pass + 12 000 photons + gather: the image equals the oracle's serial result bit for bit."""
    import cgraytracing_amd as cg
    from test_gpu_parity import _random_mesh
    rng = np.random.default_rng(500 + seed)
    objs = scenes.planes(scenes.stone_small_texture(True) if seed == 1 else None)
    objs.append(scenes.Sphere((-6.0, -14.0, 30.0), 4.0, (1, 1, 1), 0.8, 0.5 if seed != 2 else 0.0))
    for k in range(2):
        refl, transp = [(0.0, 0.0), (0.8, 0.5), (0.8, 0.0)][(seed + k) % 3]
        tri = _random_mesh(rng, int([60, 200, 11][(seed + k) % 3]), rng.uniform((-8, -16, 24), (8, -4, 34)), 7.0, 0.1)
        objs.append(scenes.TriangleMesh.from_triangles(tri, (0.6, 0.7, 0.8), refl, transp, 0))
    W, H, spp, nph = 40, 30, 1 + seed % 2, 12000
    cam = scenes.cam_dof() if seed % 2 else scenes.cam_pinhole()
    want = BackendScene(orc, objs).ppm(cam, W, H, spp, 5, nphotons=nph)
    sc = cg.Scene(objs)
    got = sc.ppm_render(W, H, spp, cam, 5, 12345, nphotons=nph)
    sc.close()
    assert got["count"] == want["n"]
    assert np.array_equal(got["image"], want["image"])
    assert want["image"].max() > 0.1
