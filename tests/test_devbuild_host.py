"""CPU: the host-visible side of row f3 (structures built on the device, cgrt_scene_set_build): the build-mode API without a
GPU, and the exp of the device-side height field against this machine's libm."""
import os
import subprocess

import numpy as np
import pytest

import scenes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_dd_exp_against_libm(tmp_path):
    """cgrt_dd::exp_dd rounds a ~100-bit value once.  libm's exp is within 0.52 ulp, i.e. NOT always correctly rounded, and
    which of its variants runs depends on the CPU (glibc picks an FMA or a non-FMA build at load time) -- the reference's
    heights are defined up to that.  Over every 4th green/blue level of all 2^24 RGB triples: never more than 1 ulp apart,
    fewer than 0.2 % differ at all, and where they differ expl sides with exp_dd."""
    exe = str(tmp_path / "ddexp_check")
    subprocess.check_call(["g++", "-O2", "-ffp-contract=off", os.path.join(ROOT, "tests", "native", "ddexp_check.cpp"), "-o", exe])
    n, differ, max_ulp, dd_nearer = map(int, subprocess.check_output([exe, "4"]).split())
    print("exp_dd vs libm exp: %d of %d arguments differ (max %d ulp); exp_dd is the nearer one in %d of them" % (differ, n, max_ulp, dd_nearer))
    assert n == 256 * 64 * 64 and max_ulp <= 1
    assert differ < 2e-3 * n
    assert dd_nearer >= 0.98 * differ


def test_build_mode_api_without_gpu():
    """set_build before the objects; opaque owners are deferred to the commit (nothing built on the host), transparent ones
    are built on the host as always; stats report the reference's counts either way."""
    from cgraytracing_amd import _capi
    from cgraytracing_amd.engine import Scene
    tris = scenes.bunny_tris()
    for transp, deferred in ((0.0, True), (0.5, False)):
        host = Scene([scenes.TriangleMesh.from_triangles(tris, (1, 1, 1), 0.8, transp)], commit=False)
        dev = Scene([scenes.TriangleMesh.from_triangles(tris, (1, 1, 1), 0.8, transp)], commit=False, build="device")
        sh, sd = host.stats(), dev.stats()
        assert sh["n_triangles"] == sd["n_triangles"] == len(tris)
        assert sh["n_nodes"] == sd["n_nodes"] == 255 and sh["scene_bytes_fp64"] == sd["scene_bytes_fp64"]
        nodes_d = dev.tree_dump(0)[0]
        assert (len(nodes_d) == 0) == deferred
        assert dev.build_info()["mode"] == _capi.BUILD_DEVICE and host.build_info()["mode"] == _capi.BUILD_HOST
        if deferred:
            assert dev.build_info()["ms_host_build"] == 0.0 and host.build_info()["ms_host_build"] > 0.0
        host.close()
        dev.close()
    floor = Scene(scenes.planes(scenes.stone_small_texture(True)), commit=False, build="device")
    st = floor.stats()
    ref = Scene(scenes.planes(scenes.stone_small_texture(True)), commit=False)
    assert st["n_triangles"] == ref.stats()["n_triangles"] and st["n_nodes"] == ref.stats()["n_nodes"]
    floor.close()
    ref.close()
    # an unknown mode is refused; the mode cannot change after the commit (no GPU here: commit is never reached)
    L = _capi.lib()
    import ctypes as C
    h = C.c_void_p()
    assert L.cgrt_scene_create(C.byref(h)) == 0
    assert L.cgrt_scene_set_build(h, 7) < 0
    L.cgrt_scene_destroy(h)


def test_env_default(monkeypatch):
    from cgraytracing_amd import _capi
    from cgraytracing_amd.engine import Scene
    monkeypatch.setenv("CGRT_BUILD", "device")
    sc = Scene(scenes.scene_c1(), commit=False)
    assert sc.build_info()["mode"] == _capi.BUILD_DEVICE
    sc.close()
    monkeypatch.setenv("CGRT_BUILD", "host")
    sc = Scene(scenes.scene_c1(), commit=False)
    assert sc.build_info()["mode"] == _capi.BUILD_HOST
    sc.close()
