"""Host-side `render(objs)` over the C ABI: flattens a list of scene objects into a cgrt_scene handle and
launches the eye pass (the loop nest of main.cpp:185-219) on the GPU."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import _capi
from ._capi import Camera as _CCamera
from ._capi import Grid as _CGrid
from ._capi import check
from .scene import Camera


def _d3(v):
    return (C.c_double * 3)(*[float(x) for x in v])


class Scene:
    """Owns a cgrt_scene handle.  `objs` order is the reference's `objs` order.  commit=False keeps the scene
    on the host only (mesh loading / tree build can then be inspected without a GPU)."""

    def __init__(self, objs, device=0, commit=True, build=None):
        """build: None (the library's default: host, or what CGRT_BUILD says), "host" or "device" (cgrt_scene_set_build: opaque
        owners' structures built on the GPU at commit; tolerance-class parity, see include/cgrt.h)."""
        L = _capi.lib()
        h = C.c_void_p()
        check(L.cgrt_scene_create(C.byref(h)))
        if build is not None:
            check(L.cgrt_scene_set_build(h, {"host": _capi.BUILD_HOST, "device": _capi.BUILD_DEVICE}[build]))
        self._h = h
        self._L = L
        self.device = int(device)
        self.obj_index = []
        tex_ids = {}
        try:
            for o in objs:
                k = o.kind
                if k == "sphere":
                    i = L.cgrt_scene_add_sphere(h, _d3(o.center), o.radius, _d3(o.surfaceColor), o.reflection,
                                                o.transparency)
                elif k == "plane":
                    tid = -1
                    t = o.texture
                    if t is not None:
                        if id(t) not in tex_ids:
                            tex_ids[id(t)] = check(L.cgrt_scene_add_texture(
                                h, t.data.ctypes.data, t.data.shape[0], t.data.shape[1], _d3(t.normal),
                                _d3(t.position), t.lenx, t.leny, int(t.isbump)))
                        tid = tex_ids[id(t)]
                    i = L.cgrt_scene_add_plane(h, _d3(o.position), _d3(o.normal), _d3(o.surfaceColor), o.reflection,
                                               o.transparency, tid)
                elif k == "mesh":
                    if o.triangles is not None:
                        i = L.cgrt_scene_add_mesh_triangles(h, o.triangles.ctypes.data, len(o.triangles),
                                                            _d3(o.surfaceColor), o.reflection, o.transparency,
                                                            o.typeofdata)
                    else:
                        i = L.cgrt_scene_add_mesh_file(h, o.filename.encode(), o.a, _d3(o.b), _d3(o.surfaceColor),
                                                       o.reflection, o.transparency, o.typeofdata)
                elif k == "bezier":
                    i = L.cgrt_scene_add_bezier(h, o.cpoints.ctypes.data, len(o.cpoints), _d3(o.position),
                                                _d3(o.surfaceColor), o.reflection, o.transparency)
                else:
                    raise TypeError("not a scene object: %r" % (o,))
                self.obj_index.append(check(i))
            if commit:
                check(L.cgrt_scene_commit(h, self.device))
        except Exception:
            self.close()
            raise

    def close(self):
        if getattr(self, "_h", None):
            self._L.cgrt_scene_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    # ---- introspection ----
    def stats(self):
        st = _capi.SceneStats()
        check(self._L.cgrt_scene_get_stats(self._h, C.byref(st)))
        return {f: getattr(st, f) for f, _ in st._fields_}

    def build_info(self):
        bi = _capi.BuildInfo()
        check(self._L.cgrt_scene_build_info(self._h, C.byref(bi)))
        return {f: getattr(bi, f) for f, _ in bi._fields_}

    def tree_dump(self, t=0):
        nn, nl, nt = C.c_int32(), C.c_int32(), C.c_int32()
        check(self._L.cgrt_scene_tree_sizes(self._h, t, C.byref(nn), C.byref(nl), C.byref(nt)))
        nodes = np.zeros((nn.value, 3), np.int32)
        leaf = np.zeros((nl.value,), np.int32)
        bbox = np.zeros((nn.value, 6), np.float64)
        tris = np.zeros((nt.value, 9), np.float64)
        check(self._L.cgrt_scene_tree_dump(self._h, t, nodes.ctypes.data, leaf.ctypes.data, bbox.ctypes.data,
                                           tris.ctypes.data))
        return nodes, leaf, bbox, tris

    def bvh_dump(self, t=0):
        """The hierarchy the device traverses (cgrt_scene_bvh_dump): boxes [8, n, 6] float32, skip [8, n], leaf [8, n]."""
        nn = C.c_int32()
        check(self._L.cgrt_scene_bvh_dump(self._h, t, C.byref(nn), None, None))
        n = nn.value
        box = np.zeros((8, n, 6), np.float32)
        sl = np.zeros((8, n, 2), np.int32)
        check(self._L.cgrt_scene_bvh_dump(self._h, t, C.byref(nn), box.ctypes.data, sl.ctypes.data))
        return box, sl[:, :, 0].copy(), sl[:, :, 1].copy()

    def bvh_order(self, t=0):
        """(tri_level, order): see cgrt_scene_bvh_order."""
        nn, nl, nt = C.c_int32(), C.c_int32(), C.c_int32()
        check(self._L.cgrt_scene_tree_sizes(self._h, t, C.byref(nn), C.byref(nl), C.byref(nt)))
        lvl = C.c_int32()
        order = np.zeros(max(nt.value, 1), np.int32)
        check(self._L.cgrt_scene_bvh_order(self._h, t, C.byref(lvl), order.ctypes.data))
        return bool(lvl.value), order[: nt.value]

    def wide_dump(self, t=0):
        """(box[nwide, 4, 6], ref[nwide, 4], stack_need): see cgrt_scene_wide_dump."""
        n, need = C.c_int32(), C.c_int32()
        check(self._L.cgrt_scene_wide_dump(self._h, t, C.byref(n), C.byref(need), None, None))
        box = np.zeros((max(n.value, 1), 4, 6), np.float32)
        ref = np.zeros((max(n.value, 1), 4), np.int32)
        check(self._L.cgrt_scene_wide_dump(self._h, t, C.byref(n), C.byref(need), box.ctypes.data, ref.ctypes.data))
        return box[: n.value], ref[: n.value], need.value

    # ---- the hot path ----
    def _structs(self, camera, width, height, rows, spp, max_depth, seed, row_offset, stripe, sample_offset,
                 spp_total, flags):
        camera = camera or Camera()
        cc = _CCamera(_d3(camera.cam), camera.half_width, camera.focus_plane, camera.lens_radius)
        s_rows, s_rank, s_n = stripe if stripe else (0, 0, 1)
        g = _CGrid(width, height, rows, row_offset, s_rows, s_rank, s_n, spp, sample_offset,
                   spp_total if spp_total else spp, max_depth, flags, seed)
        return cc, g

    def trace_grid(self, width, height, spp=1, camera=None, max_depth=5, seed=12345, rows=None, row_offset=0,
                   stripe=None, sample_offset=0, spp_total=None, out=None, nhit=None, counters=None, stream=None,
                   stats=False, accumulate=False, split_samples=False, reorder=True, force_reorder=False):
        """Asynchronous launch on torch's current stream (or `stream`).  reorder=False: CGRT_GRID_NO_REORDER (tiles in image
        order instead of heaviest-first; same image).  split_samples: CGRT_GRID_SPLIT_SAMPLES (several
        workgroups share a tile's samples; reproducible, fp64 summation order differs from the sample-by-sample sum).  Returns (rgb, nhit, counters) torch
        tensors on the scene's device: float32 [rows,width,3], int32 [rows,width] (bit pattern uint32),
        int64 [8] (counters are ADDED to)."""
        import torch

        rows = height - row_offset if rows is None else rows
        dev = torch.device("cuda", self.device)
        if out is None:
            out = torch.zeros((rows, width, 3), dtype=torch.float32, device=dev)
        if nhit is None:
            nhit = torch.zeros((rows, width), dtype=torch.int32, device=dev)
        elif nhit is False:  # skip the optional per-pixel hitpoint-count plane
            nhit = None
        if counters is None:
            counters = torch.zeros((_capi.CGRT_NCOUNTERS,), dtype=torch.int64, device=dev)
        assert out.is_contiguous() and out.dtype == torch.float32 and tuple(out.shape) == (rows, width, 3)
        cc, g = self._structs(camera, width, height, rows, spp, max_depth, seed, row_offset, stripe, sample_offset,
                              spp_total, (1 if stats else 0) | (2 if accumulate else 0) | (4 if split_samples else 0) |
                              (0 if reorder else 8) | (16 if force_reorder else 0))
        st = stream if stream is not None else torch.cuda.current_stream(dev).cuda_stream
        check(self._L.cgrt_trace_grid(self._h, C.byref(cc), C.byref(g), out.data_ptr(),
                                      nhit.data_ptr() if nhit is not None else None,
                                      counters.data_ptr(), C.c_void_p(st)))
        return out, nhit, counters

    def trace_grid_host(self, width, height, spp=1, camera=None, max_depth=5, seed=12345, rows=None, row_offset=0,
                        stripe=None, sample_offset=0, spp_total=None, stats=False, split_samples=False, reorder=True,
                        force_reorder=False):
        """Synchronous form with numpy outputs (no torch needed): dict(rgb, nhit, counters)."""
        rows = height - row_offset if rows is None else rows
        rgb = np.zeros((rows, width, 3), np.float32)
        nhit = np.zeros((rows, width), np.uint32)
        cnt = np.zeros((_capi.CGRT_NCOUNTERS,), np.uint64)
        cc, g = self._structs(camera, width, height, rows, spp, max_depth, seed, row_offset, stripe, sample_offset,
                              spp_total, (1 if stats else 0) | (4 if split_samples else 0) | (0 if reorder else 8) |
                              (16 if force_reorder else 0))
        check(self._L.cgrt_trace_grid_host(self._h, C.byref(cc), C.byref(g), rgb.ctypes.data, nhit.ctypes.data,
                                           cnt.ctypes.data))
        return dict(rgb=rgb, nhit=nhit, counters=cnt, nrays=int(cnt[_capi.CNT_RAYS]),
                    nhp=int(cnt[_capi.CNT_HITPOINTS]))

    def trace_grid_hitpoints(self, width, height, spp=1, camera=None, max_depth=5, seed=12345, rows=None,
                             row_offset=0, cap=None):
        """The reference's Hitpoint records for the grid (unordered): dict(hp [n,9] = f,pos,normal; pix [n];
        smp [n]; count)."""
        rows = height - row_offset if rows is None else rows
        cap = int(cap if cap is not None else rows * width * spp * 16)
        rec = np.zeros((max(cap, 1), 10), np.float64)
        n = C.c_uint64(0)
        cc, g = self._structs(camera, width, height, rows, spp, max_depth, seed, row_offset, None, 0, None, 0)
        check(self._L.cgrt_trace_grid_hitpoints(self._h, C.byref(cc), C.byref(g), rec.ctypes.data, cap, C.byref(n)))
        m = min(int(n.value), cap)
        lab = rec[:m, 9].astype(np.int64)
        seq, lab = lab & 15, lab >> 4
        return dict(hp=rec[:m, :9].copy(), pix=lab % (rows * width), smp=lab // (rows * width), seq=seq,
                    count=int(n.value))

    def ppm_render(self, width, height, spp=1, camera=None, max_depth=5, seed=12345, nphotons=100000, photon_seed=777,
                   hashsize=1000001, light=(0.0, 19.999, 20.0), jitter=2.0, power=700.0, alpha=0.7, batch=0,
                   want_hitpoints=False, want_rgb8=False, rows=None, row_offset=0, stripe=None, initial_radius=0.0,
                   pair_cap=0):
        """Eye pass + photon pass + final gather (+ tone map): render() main.cpp:169-258 with the serial photon
        semantics, and the PNG pixel loop of main.cpp:403-412.
        rows / row_offset / stripe select this rank's share of the frame exactly as in trace_grid (every rank traces
        all photons and owns only its rows' hitpoints; the rows equal those of a full-frame call bit for bit).
        initial_radius: the reference's 200/height of main.cpp:84,183 (0 = 200/768, its committed height); pair_cap: size
        of the per-batch pair buffer (0 = automatic; the result does not depend on it).
        Returns dict(image [rows,W,3] float64 (row 0 = bottom), count, n_events, n_pairs, ms (stage times); with
        want_hitpoints: hp [n,16]; with want_rgb8 (contiguous rows only): rgb8 [rows,W,3] uint8, top row first)."""
        rows = height if rows is None else rows
        cc, g = self._structs(camera, width, height, rows, spp, max_depth, seed, row_offset, stripe, 0, None, 0)
        ph = _capi.Photons(_d3(light), jitter, power, alpha, nphotons, hashsize, batch, photon_seed, initial_radius, pair_cap)
        height = rows
        img = np.zeros((height, width, 3), np.float64)
        cap = height * width * spp * 16 if want_hitpoints else 0
        hp = np.zeros((max(cap, 1), 16), np.float64)
        rgb8 = np.zeros((height, width, 3), np.uint8)
        res = _capi.PpmResult(img.ctypes.data, rgb8.ctypes.data if want_rgb8 else None,
                              hp.ctypes.data if want_hitpoints else None, cap)
        check(self._L.cgrt_ppm_render(self._h, C.byref(cc), C.byref(g), C.byref(ph), C.byref(res)))
        out = dict(image=img, count=int(res.hp_count), n_events=int(res.n_events), n_pairs=int(res.n_pairs),
                   n_batch_halvings=int(res.n_batch_halvings),
                   ms=dict(eye=res.ms_eye, table=res.ms_table, photons=res.ms_photons, gather=res.ms_gather))
        if want_hitpoints:
            out["hp"] = hp[: int(res.hp_count)]
        if want_rgb8:
            out["rgb8"] = rgb8
        return out

    def photon_events(self, first, count, max_depth=5, photon_seed=777, light=(0.0, 19.999, 20.0), jitter=2.0,
                      power=700.0):
        """Verification probe: diffuse hits of photons [first, first+count): [n,10] = photon, P, n, flux in serial
        order (slot order)."""
        ph = _capi.Photons(_d3(light), jitter, power, 0.7, count, 1000001, 0, photon_seed, 0.0, 0)
        ev = np.zeros((count * 8, 9), np.float64)
        va = np.zeros(count * 8, np.uint8)
        check(self._L.cgrt_photon_events(self._h, C.byref(ph), max_depth, first, count, ev.ctypes.data, va.ctypes.data))
        idx = np.nonzero(va)[0]
        return np.concatenate([(first + idx // 8)[:, None].astype(np.float64), ev[idx]], axis=1)

    def surface_colors(self, obj, pts):
        """objs[obj]->getSurfaceColor(P) on the device for each row of pts [n,3] (function-level probe)."""
        pts = np.ascontiguousarray(pts, np.float64)
        out = np.zeros_like(pts)
        check(self._L.cgrt_surface_colors(self._h, self.obj_index[obj], pts.ctypes.data, len(pts), out.ctypes.data))
        return out

    def kernel_variant(self, width, height, spp=1, camera=None, max_depth=5, rows=None, stripe=None, flags=0):
        """Name of the trace_grid_kernel instantiation this grid launches (what a rocprofv3 kernel trace shows)."""
        rows = height if rows is None else rows
        cc, g = self._structs(camera, width, height, rows, spp, max_depth, 0, 0, stripe, 0, None, flags)
        buf = C.create_string_buffer(160)
        check(self._L.cgrt_trace_grid_variant(self._h, C.byref(cc), C.byref(g), buf, len(buf)))
        return buf.value.decode()

    def intersect_rays(self, obj, org, dirs, keys=None):
        org = np.ascontiguousarray(org, np.float64)
        dirs = np.ascontiguousarray(dirs, np.float64)
        n = len(org)
        hit = np.zeros(n, np.int32)
        ln = np.zeros(n, np.float64)
        nv = np.zeros((n, 3), np.float64)
        kp = None
        if keys is not None:
            keys = np.ascontiguousarray(keys, np.uint64)
            kp = keys.ctypes.data
        check(self._L.cgrt_intersect_rays(self._h, self.obj_index[obj], org.ctypes.data, dirs.ctypes.data, kp, n,
                                          hit.ctypes.data, ln.ctypes.data, nv.ctypes.data))
        return hit, ln, nv


def render(objs, width=1024, height=768, num_of_samples=1, camera=None, max_depth=5, seed=12345, device=0):
    """Drop-in for the eye pass of render(objs) (main.cpp:169-219): returns the per-pixel accumulator
    float32 [height, width, 3] (row 0 = bottom) as a numpy array."""
    sc = Scene(objs, device)
    try:
        return sc.trace_grid_host(width, height, num_of_samples, camera, max_depth, seed)["rgb"]
    finally:
        sc.close()


def tonemap_rgb8(image, device=0):
    """gammaCorr + vertical flip (util.h:45-47, main.cpp:403-412) on the device: [H,W,3] float64, row 0 = bottom ->
    [H,W,3] uint8, top row first."""
    image = np.ascontiguousarray(image, np.float64)
    h, w = image.shape[:2]
    out = np.zeros((h, w, 3), np.uint8)
    check(_capi.lib().cgrt_tonemap_rgb8(device, image.ctypes.data, w, h, out.ctypes.data))
    return out


def write_png(path, rgb8):
    """stbi_write_png's role at main.cpp:412: [H,W,3] uint8, top row first."""
    rgb8 = np.ascontiguousarray(rgb8, np.uint8)
    h, w = rgb8.shape[:2]
    check(_capi.lib().cgrt_write_png(os.fsencode(path), w, h, rgb8.ctypes.data))
