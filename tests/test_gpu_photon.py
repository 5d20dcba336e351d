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
