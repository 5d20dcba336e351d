#!/usr/bin/env python3
"""Development aid: stage times of the reference's main() configuration (1024x768, spp 1, stone.jpg bump floor + dragon,
20 480 000 photons) through cgrt_ppm_render.  python tools/ppm_probe.py [nphotons]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import cgraytracing_amd as cg, scenes
nph = int(sys.argv[1]) if len(sys.argv) > 1 else 20480000
objs = scenes.planes(scenes.stone_texture()) + [scenes.TriangleMesh.from_triangles(scenes.dragon_tris(), (0.25, 0.25, 0.5), 0.0, 0.0, 1)]
with cg.Scene(objs) as sc:
    sc.ppm_render(64, 48, 1, scenes.cam_pinhole(), 5, 12345, nphotons=1000)
    for _ in range(2):
        r = sc.ppm_render(1024, 768, 1, scenes.cam_pinhole(), 5, 12345, nphotons=nph)
        print({k: round(v, 2) for k, v in r["ms"].items()}, "total %.1f ms" % sum(r["ms"].values()), "events", r["n_events"], flush=True)
