#!/usr/bin/env python3
"""Row f3 measurement (not a test): what building the acceleration structures on the device buys and costs.
  python tools/devbuild_probe.py [out.json]
For the dragon (100 000 triangles), the stone.jpg bump floor (146 744 triangles) and the reference's main() scene (both): wall
clock of scene construction + commit under CGRT_BUILD_HOST and CGRT_BUILD_DEVICE (three commits each, first and best), the
library's own breakdown (cgrt_scene_build_info), and the frame time of the same view with either structure -- the host
build's binned-SAH hierarchy against the device's radix tree -- plus whether the frames are bit-equal."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch

import cgraytracing_amd as cg
import scenes


def frame_ms(sc, W, H, spp, cam, reps=3):
    out = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
    cnt = torch.zeros(8, dtype=torch.int64, device="cuda")
    sc.trace_grid(W, H, spp, cam, 5, 12345, out=out, nhit=False, counters=cnt)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        sc.trace_grid(W, H, spp, cam, 5, 12345, out=out, nhit=False, counters=cnt)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps, out


def main():
    dragon = scenes.dragon_tris()
    stone = scenes.stone_texture()
    cases = {
        "dragon (100 000 triangles) + 5 planes": (lambda: scenes.planes() + [scenes.TriangleMesh.from_triangles(dragon, (0.25, 0.25, 0.5), 0.0, 0.0, 1)], 4096, 4096, 64),
        "stone.jpg bump floor (146 744 triangles) + 4 planes": (lambda: scenes.planes(stone), 2048, 2048, 16),
        "reference main() scene: stone.jpg bump floor + dragon": (lambda: scenes.planes(stone) + [scenes.TriangleMesh.from_triangles(dragon, (0.25, 0.25, 0.5), 0.0, 0.0, 1)], 2048, 2048, 16),
    }
    cam = scenes.cam_dof()
    res = {}
    torch.zeros(1, device="cuda")
    for name, (mk, W, H, spp) in cases.items():
        objs = mk()
        rec = {}
        frames = {}
        for mode in ("host", "device"):
            walls, infos = [], []
            for k in range(3):
                t0 = time.perf_counter()
                sc = cg.Scene(objs, build=mode)
                torch.cuda.synchronize()
                walls.append((time.perf_counter() - t0) * 1e3)
                infos.append(sc.build_info())
                if k < 2:
                    sc.close()
            ms, frames[mode] = frame_ms(sc, W, H, spp, cam)
            sc.close()
            rec[mode] = {"scene_construction_plus_commit_ms": {"first": round(walls[0], 2), "best_of_3": round(min(walls), 2)},
                         "library_breakdown_ms_best": {k: round(min(i[k] for i in infos), 3) for k in ("ms_host_build", "ms_device_build", "ms_commit")},
                         "frame_ms_%dx%d_spp%d" % (W, H, spp): round(ms, 3)}
        rec["frames_bit_equal"] = bool(torch.equal(frames["host"], frames["device"]))
        rec["frames_linf"] = float((frames["host"] - frames["device"]).abs().max())
        rec["pixels_not_bit_equal"] = int((frames["host"] != frames["device"]).any(dim=2).sum())
        res[name] = rec
        print(name, json.dumps(rec), flush=True)
    out = {"what": "tools/devbuild_probe.py on one MI355X: CGRT_BUILD_HOST vs CGRT_BUILD_DEVICE (row f3)", "host_cpus": os.cpu_count(), "cases": res}
    if len(sys.argv) > 1:
        json.dump(out, open(sys.argv[1], "w"), indent=1)


if __name__ == "__main__":
    main()
