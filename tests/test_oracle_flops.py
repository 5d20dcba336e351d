"""CPU: the instrumented build of the oracle (oracle/liborc_flops.so: `real` counts its arithmetic, oracle/cgrt_flopcount.h) --
the exact algorithmic flop count SURVEY.md section 8d asks for.  It must render what the oracle renders, and its counts must
behave like counts."""
import numpy as np

import scenes
from backends import Backend, BackendScene


def _count(be, objs, cam, W, H, spp, depth=5, **kw):
    sc = BackendScene(be, objs)
    be.flop_reset()
    r = sc.trace_grid(cam, W, H, spp, depth, seed=12345, **kw)
    c = be.flop_counts()
    sc.close()
    return r, c


def test_instrumented_build_renders_the_oracles_image(orc):
    be = Backend("flops")
    for objs, cam in ((scenes.scene_c2(), scenes.cam_dof()), (scenes.scene_c3(True), scenes.cam_pinhole()),
                      (scenes.scene_c5(scenes.stone_small_texture(True)), scenes.cam_dof())):
        got, c = _count(be, objs, cam, 48, 36, 2)
        o = BackendScene(orc, objs)
        want = o.trace_grid(cam, 48, 36, 2, 5, seed=12345)
        o.close()
        assert got["nrays"] == want["nrays"] and np.array_equal(got["nhit"], want["nhit"])
        assert np.array_equal(got["acc_sum"], want["acc_sum"])  # fp64 sums, bit for bit: counting changes no value
        assert c["flops"] > 100 * got["nrays"]


def test_counts_are_additive_and_pinned():
    be = Backend("flops")
    objs, cam = scenes.scene_c1(), scenes.cam_pinhole()
    r1, c1 = _count(be, objs, cam, 32, 32, 1, depth=1)
    r3, c3 = _count(be, objs, cam, 32, 32, 3, depth=1)
    # pinhole samples of a pixel are the same ray: three samples cost exactly three times one, minus the per-pixel camera set-up
    top, ctop = _count(be, objs, cam, 32, 32, 1, depth=1, row0=0, nrows=16)
    bot, cbot = _count(be, objs, cam, 32, 32, 1, depth=1, row0=16, nrows=16)
    for k in ("add_sub", "mul", "div", "sqrt", "transcendental_calls"):
        assert ctop[k] + cbot[k] == c1[k], k  # row ranges add up exactly
        per_sample = (c3[k] - c1[k]) // 2
        assert c3[k] - c1[k] == 2 * per_sample and per_sample <= c1[k]
    assert c1["transcendental_calls"] == 0  # spheres: no pow / sin / cos anywhere on the path
    # regression pin of the C1 restatement (6 spheres, depth 1, 32 x 32 primary rays): operations per ray
    per_ray = c1["flops"] / r1["nrays"]
    print("C1: %.2f flops per ray (%s)" % (per_ray, c1))
    assert 120 < per_ray < 260
