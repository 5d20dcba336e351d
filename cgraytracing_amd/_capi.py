"""ctypes binding of libcgrt.so (include/cgrt.h).  There is no fallback: if the HIP library has not been
built (``python -c "import __graft_entry__ as g; g.build()"`` or ``make -C cgraytracing_amd/csrc``) importing
this module raises."""
from __future__ import annotations

import ctypes as C
import os

import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
CGRT_VERSION = 112  # include/cgrt.h: the ABI this binding was written against
# Development hook: CGRT_LIB names another build of the same ABI (make exp / make fma: other occupancy or contraction choices)
# and is honoured ONLY together with CGRT_DEV_LIBS=1 -- a stray variable must not swap the product's library.
LIB_PATH = os.path.join(_HERE, "libcgrt.so")
if os.environ.get("CGRT_LIB"):
    if os.environ.get("CGRT_DEV_LIBS") == "1":
        LIB_PATH = os.environ["CGRT_LIB"]
        print("cgraytracing_amd: DEVELOPMENT library %s (CGRT_LIB + CGRT_DEV_LIBS=1)" % LIB_PATH, file=sys.stderr)
    else:
        print("cgraytracing_amd: CGRT_LIB ignored (set CGRT_DEV_LIBS=1 to load a development build)", file=sys.stderr)

CGRT_OK = 0
CGRT_NCOUNTERS = 8
CNT_RAYS, CNT_HITPOINTS, CNT_WAVE_ITERS, CNT_NODE_TESTS, CNT_TRI_TESTS = 0, 1, 2, 3, 4


class CgrtError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libcgrt error %d: %s" % (code, msg))
        self.code = code


class Camera(C.Structure):
    _fields_ = [("cam", C.c_double * 3), ("half_width", C.c_double), ("focus_plane", C.c_double),
                ("lens_radius", C.c_double)]


class Grid(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("rows", C.c_int32), ("row_offset", C.c_int32),
                ("stripe_rows", C.c_int32), ("stripe_rank", C.c_int32), ("stripe_nranks", C.c_int32),
                ("spp", C.c_int32), ("sample_offset", C.c_int32), ("spp_total", C.c_int32),
                ("max_depth", C.c_int32), ("flags", C.c_int32), ("seed", C.c_uint64)]


class Photons(C.Structure):
    _fields_ = [("light", C.c_double * 3), ("jitter", C.c_double), ("power", C.c_double), ("alpha", C.c_double),
                ("nphotons", C.c_int64), ("hashsize", C.c_int32), ("batch", C.c_int32), ("seed", C.c_uint64),
                ("initial_radius", C.c_double), ("pair_cap", C.c_int64)]


class PpmResult(C.Structure):
    _fields_ = [("image", C.c_void_p), ("rgb8", C.c_void_p), ("hp16", C.c_void_p), ("hp_cap", C.c_uint64),
                ("hp_count", C.c_uint64), ("n_events", C.c_uint64), ("n_pairs", C.c_uint64), ("n_batch_halvings", C.c_uint64),
                ("ms_eye", C.c_double),
                ("ms_table", C.c_double), ("ms_photons", C.c_double), ("ms_gather", C.c_double)]


class SceneStats(C.Structure):
    _fields_ = [("n_objects", C.c_int32), ("n_spheres", C.c_int32), ("n_planes", C.c_int32),
                ("n_meshes", C.c_int32), ("n_beziers", C.c_int32), ("n_textures", C.c_int32),
                ("n_trees", C.c_int32), ("committed", C.c_int32), ("n_triangles", C.c_int64),
                ("n_nodes", C.c_int64), ("device_bytes", C.c_int64), ("scene_bytes_fp64", C.c_int64)]


class BuildInfo(C.Structure):
    _fields_ = [("mode", C.c_int32), ("n_device_trees", C.c_int32), ("ms_host_build", C.c_double),
                ("ms_device_build", C.c_double), ("ms_commit", C.c_double)]


BUILD_HOST, BUILD_DEVICE = 0, 1

# every symbol include/cgrt.h declares, with its signature
_DP = C.POINTER(C.c_double)
SIGNATURES = {
    "cgrt_version": (C.c_int, []),
    "cgrt_last_error": (C.c_char_p, []),
    "cgrt_scene_create": (C.c_int, [C.POINTER(C.c_void_p)]),
    "cgrt_scene_destroy": (None, [C.c_void_p]),
    "cgrt_scene_set_build": (C.c_int, [C.c_void_p, C.c_int]),
    "cgrt_scene_build_info": (C.c_int, [C.c_void_p, C.POINTER(BuildInfo)]),
    "cgrt_scene_add_sphere": (C.c_int, [C.c_void_p, _DP, C.c_double, _DP, C.c_double, C.c_double]),
    "cgrt_scene_add_texture": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, _DP, _DP, C.c_double,
                                         C.c_double, C.c_int]),
    "cgrt_scene_add_plane": (C.c_int, [C.c_void_p, _DP, _DP, _DP, C.c_double, C.c_double, C.c_int]),
    "cgrt_scene_add_mesh_file": (C.c_int, [C.c_void_p, C.c_char_p, C.c_double, _DP, _DP, C.c_double, C.c_double,
                                           C.c_int]),
    "cgrt_scene_add_mesh_triangles": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, _DP, C.c_double, C.c_double,
                                                C.c_int]),
    "cgrt_scene_add_bezier": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, _DP, _DP, C.c_double, C.c_double]),
    "cgrt_scene_commit": (C.c_int, [C.c_void_p, C.c_int]),
    "cgrt_scene_get_stats": (C.c_int, [C.c_void_p, C.POINTER(SceneStats)]),
    "cgrt_scene_tree_sizes": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                        C.POINTER(C.c_int32)]),
    "cgrt_scene_tree_dump": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "cgrt_trace_grid": (C.c_int, [C.c_void_p, C.POINTER(Camera), C.POINTER(Grid), C.c_void_p, C.c_void_p,
                                  C.c_void_p, C.c_void_p]),
    "cgrt_unpermute_stripes": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                         C.c_void_p]),
    "cgrt_trace_grid_host": (C.c_int, [C.c_void_p, C.POINTER(Camera), C.POINTER(Grid), C.c_void_p, C.c_void_p,
                                       C.c_void_p]),
    "cgrt_trace_grid_hitpoints": (C.c_int, [C.c_void_p, C.POINTER(Camera), C.POINTER(Grid), C.c_void_p, C.c_uint64,
                                            C.POINTER(C.c_uint64)]),
    "cgrt_ppm_render": (C.c_int, [C.c_void_p, C.POINTER(Camera), C.POINTER(Grid), C.POINTER(Photons),
                                  C.POINTER(PpmResult)]),
    "cgrt_tonemap_rgb8": (C.c_int, [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "cgrt_write_png": (C.c_int, [C.c_char_p, C.c_int, C.c_int, C.c_void_p]),
    "cgrt_photon_events": (C.c_int, [C.c_void_p, C.POINTER(Photons), C.c_int, C.c_int64, C.c_int32, C.c_void_p,
                                     C.c_void_p]),
    "cgrt_scene_bvh_dump": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_int32), C.c_void_p, C.c_void_p]),
    "cgrt_scene_bvh_order": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_int32), C.c_void_p]),
    "cgrt_scene_wide_dump": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_void_p, C.c_void_p]),
    "cgrt_lens_samples": (C.c_int, [C.c_uint64, C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_void_p]),
    "cgrt_surface_colors": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    "cgrt_trace_grid_variant": (C.c_int, [C.c_void_p, C.POINTER(Camera), C.POINTER(Grid), C.c_char_p, C.c_size_t]),
    "cgrt_intersect_rays": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                      C.c_void_p, C.c_void_p, C.c_void_p]),
}

_lib = None


def lib():
    """The loaded library; raises if libcgrt.so is missing (no CPU fallback exists)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "cgraytracing_amd: %s not found -- build the HIP library first "
                "(make -C cgraytracing_amd/csrc, or __graft_entry__.build()). There is no CPU fallback." % LIB_PATH)
        # Load order matters where PyTorch is in the process: libtorch_hip needs "libamdhip64.so" (its bundled copy, found by
        # RPATH) while libcgrt.so needs "libamdhip64.so.7".  With torch first, our NEEDED entry matches the SONAME of the copy
        # already loaded and the process has ONE HIP runtime; with libcgrt.so first, torch's name matches nothing loaded, a
        # second runtime comes in and one of the two then finds "no ROCm-capable device".  So torch, when importable, goes first
        # -- deliberately, also for the torch-free entry points (trace_grid_host, ppm_render): the cost is torch's import time.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)  # AttributeError if a declared symbol is not exported
            fn.restype = res
            fn.argtypes = args
        if L.cgrt_version() != CGRT_VERSION:  # same symbols, other struct layouts: refuse rather than corrupt memory
            raise ImportError("cgraytracing_amd: %s is ABI version %d, this binding needs %d -- rebuild it (make -C cgraytracing_amd/csrc)"
                              % (LIB_PATH, L.cgrt_version(), CGRT_VERSION))
        _lib = L
    return _lib


def check(rc):
    if rc < 0:
        raise CgrtError(rc, lib().cgrt_last_error().decode("utf-8", "replace"))
    return rc
