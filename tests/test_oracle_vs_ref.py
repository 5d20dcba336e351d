"""CPU, build container only: the oracle against the compiled reference run LIVE (oracle/_ref).  Skipped
where the prebuilt harness is absent.  Complements the committed golden vectors with fresh random cases."""
import numpy as np
import pytest

import scenes
from backends import BackendScene, lens_samples


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_random_sphere_scenes(orc, ref, seed):
    rng = np.random.default_rng(seed)
    objs = scenes.wall_spheres()
    for _ in range(5):
        c = (rng.uniform(-15, 15), rng.uniform(-18, 10), rng.uniform(20, 38))
        kind = rng.integers(0, 3)
        refl, transp = [(0, 0), (0.8, 0), (0.8, 0.5)][kind]
        objs.append(scenes.Sphere(c, rng.uniform(2, 6), tuple(rng.uniform(0.2, 1, 3)), refl, transp))
    cam = scenes.cam_dof()
    a = BackendScene(orc, objs).trace_grid(cam, 80, 60, 3, 5, seed=seed, capture=True)
    b = BackendScene(ref, objs).trace_grid(cam, 80, 60, 3, 5, seed=seed, capture=True)
    assert a["nrays"] == b["nrays"]
    assert np.array_equal(a["acc_sum"], b["acc_sum"]) and np.array_equal(a["hp"], b["hp"])


def test_procedural_glass_mesh(orc, ref):
    tris = scenes.procedural_mesh(24, 12, (0, -8, 30), 7.0)
    m = scenes.TriangleMesh.from_triangles(tris, (1, 1, 1), 0.8, 0.5)
    objs = scenes.planes(scenes.chessboard_texture(False)) + [m]
    a = BackendScene(orc, objs).trace_grid(scenes.cam_dof(), 64, 64, 2, 5, capture=True)
    b = BackendScene(ref, objs).trace_grid(scenes.cam_dof(), 64, 64, 2, 5, capture=True)
    assert a["nrays"] == b["nrays"] and np.array_equal(a["hp"], b["hp"])


def test_lens_sampler(orc, ref):
    pix = np.arange(1000, dtype=np.int64) * 7919
    smp = (np.arange(1000) % 64).astype(np.int32)
    assert np.array_equal(lens_samples(orc, 99, pix, smp, 1.5), lens_samples(ref, 99, pix, smp, 1.5))


def test_bezier_function_level(orc, ref):
    bz = scenes.vase_bezier()
    rng = np.random.default_rng(77)
    n = 600
    org = np.tile(np.array([0, 0, -10.0]), (n, 1)) + rng.normal(size=(n, 3))
    tgt = np.array([15, -10.1, 35.0]) + (rng.random((n, 3)) - 0.5) * np.array([10, 22, 10])
    d = tgt - org
    d /= np.linalg.norm(d, axis=1)[:, None]
    keys = rng.integers(0, 2 ** 63, n, dtype=np.uint64)
    h1, l1, n1 = BackendScene(orc, [bz]).intersect_batch(0, org, d, keys)
    h2, l2, n2 = BackendScene(ref, [bz]).intersect_batch(0, org, d, keys)
    assert np.array_equal(h1, h2) and np.array_equal(l1, l2) and np.array_equal(n1[h1 == 1], n2[h2 == 1])
