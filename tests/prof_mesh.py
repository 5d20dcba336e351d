"""Profiling driver (not a test): one mesh workload, a few launches.  rocprofv3 ... -- python3 tests/prof_mesh.py [dragon|bunny|bump]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import perf_probe as pp, scenes
which = sys.argv[1] if len(sys.argv) > 1 else "dragon"
if which == "dragon":
    pp.run("dragon 2048 spp16", scenes.scene_dragon(), scenes.cam_dof(), 2048, 2048, 16, 5, reps=2)
elif which == "bunny":
    pp.run("bunny 2048 spp16", scenes.scene_c3(True), scenes.cam_dof(), 2048, 2048, 16, 5, reps=2)
elif which == "c5band":
    tex = scenes.stone_texture()
    pp.run("c5 band rows 3000..3255", scenes.scene_c5(tex), scenes.cam_dof(), 8192, 8192, 16, 5, reps=1, rows=256, row_offset=3000)
elif which == "c5quarter":  # a quarter of C5's per-GPU share in samples: every workgroup slot is busy with vase tiles
    tex = scenes.stone_texture()
    pp.run("c5 share spp 256", scenes.scene_c5(tex), scenes.cam_dof(), 8192, 8192, 256, 5, reps=1, rows=1024, row_offset=3584)
elif which == "vase":
    pp.run("vase 1536 spp8", scenes.planes() + [scenes.vase_bezier()], scenes.cam_dof(), 1536, 1536, 8, 5, reps=2)
else:
    tex = scenes.stone_texture()
    pp.run("bump 1024 spp4", scenes.planes(tex), scenes.cam_dof(), 1024, 1024, 4, 5, reps=2)
