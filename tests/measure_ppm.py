"""Measurement driver (not a test): the reference's COMMITTED configuration end to end on one MI355X --
main.cpp:28-29,177,223-224,292,320,348-353: 1024x768, 1 sample per pixel, planes + stone bump floor + diffuse dragon,
2 560 000 x 8 = 20 480 000 photons -- through cgrt_ppm_render (eye pass, hash table order, serial-semantics photon
pass, final gather, tone map) and cgrt_write_png.  The stone texture is the seeded stand-in of stone.jpg's size
(the JPEG is not shipped).  Beside it: the oracle (CPU port of the same serial semantics, 1 thread) on a bounded
photon sample of the same scene, and a bit-for-bit comparison of that sample's image with the GPU's.

    python tests/measure_ppm.py [--photons N] [--cpu-photons M] [--png gpurun_out/ppm_reference_config.png]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

import cgraytracing_amd as cg
import scenes
from backends import Backend, BackendScene


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--photons", type=int, default=20480000)
    ap.add_argument("--cpu-photons", type=int, default=40000)
    ap.add_argument("--width", type=int, default=1024)
    ap.add_argument("--height", type=int, default=768)
    ap.add_argument("--png", default="")
    ap.add_argument("--scene", default="committed", choices=["committed", "c2", "bunny"])
    args = ap.parse_args()
    W, H = args.width, args.height
    if args.scene == "committed":
        tex = scenes.stone_texture()
        objs = scenes.planes(tex) + [scenes.TriangleMesh.from_triangles(scenes.dragon_tris(), (0.25, 0.25, 0.5), 0.0, 0.0, 1)]
    elif args.scene == "c2":
        objs = scenes.scene_c2()
    else:
        objs = scenes.scene_c3(True)
    cam = scenes.cam_pinhole()
    t0 = time.time()
    sc = cg.Scene(objs)
    t_build = time.time() - t0
    sc.ppm_render(64, 48, 1, cam, 5, 12345, nphotons=1000)  # warm-up (module load, allocator)
    t0 = time.time()
    r = sc.ppm_render(W, H, 1, cam, 5, 12345, nphotons=args.photons, want_rgb8=True)
    wall = time.time() - t0
    doc = {
        "workload": "%s scene %dx%d spp 1, %d photons (main.cpp committed configuration: 1024x768, 20 480 000)" %
                    (args.scene, W, H, args.photons),
        "scene_build_s": round(t_build, 3),
        "gpu": {"wall_s": round(wall, 3), "stage_ms": {k: round(v, 2) for k, v in r["ms"].items()},
                "hitpoints": r["count"], "photon_events": r["n_events"], "pairs_replayed": r["n_pairs"],
                "photons_per_s": round(args.photons / (r["ms"]["photons"] / 1e3), 0),
                "image_mean": float(r["image"].mean()), "rgb8_mean": float(r["rgb8"].mean())},
    }
    if args.png:
        cg.write_png(args.png, r["rgb8"])
        doc["png"] = args.png
    if args.cpu_photons > 0:
        m = args.cpu_photons
        g = sc.ppm_render(W, H, 1, cam, 5, 12345, nphotons=m)
        be = Backend("orc")
        be.set_threads(1)
        o = BackendScene(be, objs)
        t0 = time.time()
        want = o.ppm(cam, W, H, 1, 5, nphotons=m)
        cpu_s = time.time() - t0
        doc["cpu_port_1_thread"] = {
            "sample": "same scene and grid, first %d photons (eye pass included)" % m, "seconds": round(cpu_s, 2),
            "gpu_same_sample_wall_ms": round(sum(g["ms"].values()), 2),
            "image_bit_identical": bool(np.array_equal(g["image"], want["image"])),
            "max_abs_diff": float(np.abs(g["image"] - want["image"]).max()),
        }
    sc.close()
    print(json.dumps(doc))


if __name__ == "__main__":
    main()
