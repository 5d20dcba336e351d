"""Scene definitions used by tests, golden generation and bench.py.  Geometry and materials follow
SURVEY.md §8d, which in turn cites the (partly commented-out) scene code of main.cpp:281-376."""
from __future__ import annotations

import os

import numpy as np

from cgraytracing_amd.scene import Bezier, Camera, Plane, Sphere, Texture, TriangleMesh

HERE = os.path.dirname(os.path.abspath(__file__))
ASSETS = os.path.join(HERE, "golden", "assets")


def wall_spheres():
    # main.cpp:281-285
    return [
        Sphere((0.0, -10020, 0), 10000, (0.25, 0.25, 0.25), 0.0, 0.0),
        Sphere((10020, 0.0, 0), 10000, (0.25, 0.75, 0.25), 0.0, 0.0),
        Sphere((-10020, 0.0, 0), 10000, (0.75, 0.25, 0.25), 0.0, 0.0),
        Sphere((0.0, 0.0, 10040), 10000, (0.25, 0.25, 0.25), 0.0, 0.0),
        Sphere((0.0, 10020, 0), 10000, (0.25, 0.25, 0.25), 0.0, 0.0),
    ]


def scene_c1():
    """C1: six diffuse spheres (main.cpp:281-285,288)."""
    return wall_spheres() + [Sphere((-15.0, -20.0, 60), 10, (0.3, 0.3, 0.3), 0.0, 0.0)]


def scene_c2():
    """C2: walls + diffuse + mirror + glass spheres (materials main.cpp:288-290; centres per SURVEY §8d)."""
    return wall_spheres() + [
        Sphere((-15.0, -20.0, 60), 10, (0.3, 0.3, 0.3), 0.0, 0.0),
        Sphere((10.0, -13.0, 30), 7, (1.0, 1.0, 1.0), 0.8, 0.0),
        Sphere((-8.0, -13.0, 25), 7, (1.0, 1.0, 1.0), 0.8, 0.5),
    ]


def planes(floor_tex=None):
    # main.cpp:348-353
    return [
        Plane((0.0, -20, 0), (0, 1, 0), (0.15, 0.15, 0.15), 0.0, 0.0, floor_tex),
        Plane((20, 0.0, 0), (-1, 0, 0), (0.15, 0.50, 0.15), 0.0, 0.0),
        Plane((-20, 0.0, 0), (1, 0, 0), (0.50, 0.15, 0.15), 0.0, 0.0),
        Plane((0.0, 0.0, 40), (0, 0, -1), (0.15, 0.15, 0.15), 0.0, 0.0),
        Plane((0.0, 20, 0), (0, -1, 0), (0.15, 0.15, 0.15), 0.0, 0.0),
    ]


def load_asset(name):
    return np.load(os.path.join(ASSETS, name))


def chessboard_texture(bump=False):
    d = load_asset("chessboard_rgb.npz")["rgb"]
    return Texture(d, (0, 1, 0), (-21, 0, 0), 42, 40, bump)  # main.cpp:320 geometry


def stone_small_texture(bump=True):
    d = load_asset("stone_small_rgb.npz")["rgb"]
    return Texture(d, (0, 1, 0), (-21, 0, 0), 42, 40, bump)


def stone_texture(bump=True):
    """texture/stone.jpg at its full 1000x667 as the reference's stb decoder returns it, placed as main.cpp:320 does:
    Texture(tdata, (0,1,0), (-21,0,0), 42, 40, true).  As a bump map: 146 744 triangles, 32 767 nodes."""
    d = load_asset("stone_rgb.npz")["rgb"]
    return Texture(d, (0, 1, 0), (-21, 0, 0), 42, 40, bump)


def procedural_stone(rows=667, cols=1000, seed=7):
    """Seeded stand-in of texture/stone.jpg's size (kept for fuzz-style tests; the configurations use stone_texture())."""
    rng = np.random.default_rng(seed)
    base = rng.random((rows // 8 + 2, cols // 8 + 2))
    up = np.kron(base, np.ones((8, 8)))[:rows, :cols]
    fine = rng.random((rows, cols))
    g = np.clip(0.25 + 0.5 * up + 0.25 * fine, 0, 1)
    rgb = np.stack([g * 200 + 20, g * 190 + 20, g * 170 + 20], -1).astype(np.uint8)
    return np.ascontiguousarray(rgb)


def bunny_tris():
    """model/lowpolybunny.txt through the type-0 loader transform a=10, b=(0,-15,40) (main.cpp:293)."""
    return load_asset("bunny_tris.npz")["tris"]


def scene_c3(glass=True):
    """C3: 5 planes (chessboard floor) + glass bunny (main.cpp:293,348-353)."""
    floor = chessboard_texture(False)
    m = TriangleMesh.from_triangles(bunny_tris(), (1.0, 1.0, 1.0), 0.8 if glass else 0.0, 0.5 if glass else 0.0)
    return planes(floor) + [m]


def pyramid_tris(a=1.0, b=(0.0, -5.0, 30.0)):
    """The 6-triangle square pyramid of model/tri.txt (type-1 file, 5 vertices), transformed."""
    v = np.array([[5, 0, 5], [5, 0, -5], [-5, 0, 5], [-5, 0, -5], [0, -10, 0]], np.float64)
    f = np.array([[1, 2, 5], [1, 3, 5], [2, 4, 5], [3, 4, 5], [1, 2, 4], [1, 3, 4]]) - 1
    v = v * np.array([1, 1, -1.0]) * a + np.asarray(b, np.float64)
    return v[f].reshape(-1, 9)


def scene_pyramid(glass=False):
    m = TriangleMesh.from_triangles(pyramid_tris(1.0, (0.0, -5.0, 30.0)), (0.6, 0.7, 0.9),
                                    0.8 if glass else 0.0, 0.5 if glass else 0.0)
    return planes() + [m]


def procedural_mesh(n_u=40, n_v=20, center=(-5.0, -10.0, 30.0), radius=8.0, seed=3, wobble=0.15):
    """Closed lumpy sphere mesh with 2*n_u*(n_v-1) triangles (seeded), for mesh configs whose
    original model file is not shipped."""
    rng = np.random.default_rng(seed)
    th = np.linspace(0, np.pi, n_v + 1)
    ph = np.linspace(0, 2 * np.pi, n_u, endpoint=False)
    r = radius * (1 + wobble * (rng.random((n_v + 1, n_u)) - 0.5))
    r[0, :] = r[0, 0]
    r[-1, :] = r[-1, 0]
    P = np.zeros((n_v + 1, n_u, 3))
    P[..., 0] = r * np.sin(th)[:, None] * np.cos(ph)[None, :]
    P[..., 1] = r * np.cos(th)[:, None]
    P[..., 2] = r * np.sin(th)[:, None] * np.sin(ph)[None, :]
    P += np.asarray(center, np.float64)
    tris = []
    for i in range(n_v):
        for j in range(n_u):
            j2 = (j + 1) % n_u
            a, b, c, d = P[i, j], P[i, j2], P[i + 1, j], P[i + 1, j2]
            if i > 0:
                tris.append([a, b, c])
            if i < n_v - 1:
                tris.append([d, b, c])
    return np.asarray(tris, np.float64).reshape(-1, 9)


def scene_c4(n_u=320, n_v=158):
    """C4 stand-in: untextured planes + a ~100k-triangle diffuse procedural mesh placed where the
    dragon sits (main.cpp:292: scale 1.5, offset (-5,-20,30), colour (0.25,0.25,0.5))."""
    m = TriangleMesh.from_triangles(procedural_mesh(n_u, n_v, (-5.0, -10.0, 30.0), 9.0), (0.25, 0.25, 0.5), 0.0, 0.0, 1)
    return planes() + [m]


def vase_bezier(refl=0.5, transp=0.0):
    # main.cpp:371-376
    cp = [(0, -10, 4), (0, 2, 4), (0, -2, 0), (0, 10, 2)]
    return Bezier(cp, (15, -10.1, 35), (1.0, 1.0, 1.0), refl, transp)


def scene_c5(tex=None):
    """C5: planes with bump floor + Bezier vase."""
    return planes(tex) + [vase_bezier()]


def textured_walls():
    """Textures in all three orientations of Texture::color (texture.h:39-72): the chessboard on the floor (|d.y| branch),
    on the back wall (|d.z| branch, vertically flipped) and on the right-hand wall (|d.x| branch, tested first), sized so
    that part of every wall lies outside the texture rectangle (flat colour there, objects.h:533-539)."""
    chess = load_asset("chessboard_rgb.npz")["rgb"]
    floor = Texture(chess, (0, 1, 0), (-21, -20, 0), 42, 40, False)
    back = Texture(chess, (0, 0, -1), (-15, -18, 40), 30, 25, False)
    side = Texture(chess, (-1, 0, 0), (20, -16, 5), 30, 28, False)
    return [
        Plane((0.0, -20, 0), (0, 1, 0), (0.15, 0.15, 0.15), 0.0, 0.0, floor),
        Plane((20, 0.0, 0), (-1, 0, 0), (0.15, 0.50, 0.15), 0.0, 0.0, side),
        Plane((-20, 0.0, 0), (1, 0, 0), (0.50, 0.15, 0.15), 0.0, 0.0),
        Plane((0.0, 0.0, 40), (0, 0, -1), (0.15, 0.15, 0.15), 0.0, 0.0, back),
        Plane((0.0, 20, 0), (0, -1, 0), (0.15, 0.15, 0.15), 0.0, 0.0),
    ]


def scene_textured_walls():
    """textured_walls() + a mirror and a glass sphere, so reflected and refracted rays reach the walls at all angles."""
    return textured_walls() + [Sphere((9.0, -13.0, 30), 7, (1.0, 1.0, 1.0), 0.8, 0.0),
                               Sphere((-8.0, -13.0, 25), 7, (1.0, 1.0, 1.0), 0.8, 0.5)]


def scene_glass_bump_floor():
    """A TRANSPARENT bump-mapped floor: its displacement mesh goes through the tree (the improvement counter is observable
    for a refracting owner), not the height-field walk of opaque floors; a second textured floor below catches
    the refracted rays."""
    pl = planes(None)
    pl[0] = Plane((0.0, -20, 0), (0, 1, 0), (0.9, 0.9, 0.9), 0.8, 0.5, chessboard_texture(True))
    chess = load_asset("chessboard_rgb.npz")["rgb"]
    below = Plane((0.0, -24, 0), (0, 1, 0), (0.2, 0.2, 0.3), 0.0, 0.0, Texture(chess, (0, 1, 0), (-30, -24, 0), 60, 60, False))
    return pl + [below]


def cam_pinhole():
    return Camera()


def cam_dof():
    return Camera(lens_radius=1.5)  # main.cpp:178-181


def dragon_tris(a=1.5, b=(-5.0, -20.0, 30.0)):
    """model/dragon.txt (type-1 file) through the loader transform (x,y,-z)*a+b (objects.h:365,371)."""
    d = load_asset("dragon_mesh.npz")
    v = d["v1e4"].astype(np.float64) / 1e4
    v = v * np.array([1.0, 1.0, -1.0])
    v = v * a + np.asarray(b, np.float64)
    return v[d["faces"] - 1].reshape(-1, 9)


def scene_dragon():
    """The committed main() scene minus the bump floor: planes + diffuse dragon (main.cpp:292,348-353)."""
    m = TriangleMesh.from_triangles(dragon_tris(), (0.25, 0.25, 0.5), 0.0, 0.0, 1)
    return planes() + [m]
