"""Host-side scene description mirroring the reference's object API.

The classes keep the reference's constructor argument order and meaning so that a scene
written for the reference reads the same here:

    Sphere(c, r, sc, refl=0, transp=0)                         objects.h:28-38
    Plane(p, n, sc, refl=0, transp=0, tx=None)                 objects.h:480
    TriangleMesh(filename, a, b, sc, refl=0, transp=0, typeofdata=0)   objects.h:338-340
    Bezier(points, pos, sc, refl=0, transp=0)                  bezier.h:44-45
    Texture(data, n, p, lx, ly, flag=False)                    texture.h:19

They are inert parameter holders: all geometry processing (mesh loading, tree build, bump-mesh
construction) and all tracing happens behind the C ABI in libcgrt.so (include/cgrt.h).
`objs` is an ordered list, exactly like `vector<Object*> objs` (main.cpp:277,355-366): the first
object wins equal-distance ties (main.cpp:57).
"""
from __future__ import annotations

import numpy as np


def _v3(v):
    a = np.asarray(v, dtype=np.float64).reshape(-1)
    if a.shape != (3,):
        raise ValueError("expected a 3-vector, got shape %r" % (a.shape,))
    return a


class Vec3(tuple):
    """3-vector of doubles (vec3.h:11-30); a tuple so it can be passed wherever a 3-sequence is."""

    def __new__(cls, x=0.0, y=0.0, z=0.0):
        return super().__new__(cls, (float(x), float(y), float(z)))

    x = property(lambda s: s[0])
    y = property(lambda s: s[1])
    z = property(lambda s: s[2])


class Object:
    """Base of the scene objects (objects.h:17-24)."""

    kind = "object"


class Texture:
    """Planar texture / bump map (texture.h:19-38).

    `data` is rows x cols x 3 uint8 exactly as the image decoder returns it; the reference divides
    bytes by 256 (main.cpp:303-316) and the library does the same on the device.
    """

    def __init__(self, data, n, p, lx, ly, flag=False):
        d = np.ascontiguousarray(np.asarray(data))
        if d.dtype != np.uint8 or d.ndim != 3 or d.shape[2] != 3:
            raise ValueError("Texture data must be uint8 [rows, cols, 3]")
        self.data = d
        self.normal = _v3(n)
        self.position = _v3(p)
        self.lenx = float(lx)
        self.leny = float(ly)
        self.isbump = bool(flag)


class Sphere(Object):
    kind = "sphere"

    def __init__(self, c, r, sc, refl=0.0, transp=0.0):
        self.center = _v3(c)
        self.radius = float(r)
        self.surfaceColor = _v3(sc)
        self.reflection = float(refl)
        self.transparency = float(transp)


class Plane(Object):
    kind = "plane"

    def __init__(self, p, n, sc, refl=0.0, transp=0.0, tx=None):
        self.position = _v3(p)
        self.normal = _v3(n)
        self.surfaceColor = _v3(sc)
        self.reflection = float(refl)
        self.transparency = float(transp)
        self.texture = tx


class TriangleMesh(Object):
    """Triangle mesh from one of the reference's three text formats (typeofdata 0/1/2), vertices
    mapped to (x, y, -z) * a + b (objects.h:348,365,384); or, via `from_triangles`, from an
    [ntri, 3, 3] array of already transformed vertices."""

    kind = "mesh"

    def __init__(self, filename, a, b, sc, refl=0.0, transp=0.0, typeofdata=0):
        self.filename = None if filename is None else str(filename)
        self.a = float(a)
        self.b = _v3(b)
        self.surfaceColor = _v3(sc)
        self.reflection = float(refl)
        self.transparency = float(transp)
        self.typeofdata = int(typeofdata)
        self.triangles = None

    @classmethod
    def from_triangles(cls, tris, sc, refl=0.0, transp=0.0, typeofdata=0):
        m = cls(None, 1.0, (0, 0, 0), sc, refl, transp, typeofdata)
        t = np.ascontiguousarray(np.asarray(tris, dtype=np.float64)).reshape(-1, 9)
        m.triangles = t
        return m


class Bezier(Object):
    kind = "bezier"

    def __init__(self, points, pos, sc, refl=0.0, transp=0.0):
        pts = np.ascontiguousarray(np.asarray(points, dtype=np.float64)).reshape(-1, 3)
        if not (1 <= len(pts) <= 6):
            raise ValueError("Bezier takes 1..6 control points (bezier.h:46)")
        self.cpoints = pts
        self.position = _v3(pos)
        self.surfaceColor = _v3(sc)
        self.reflection = float(refl)
        self.transparency = float(transp)


class Camera:
    """Camera constants of render() (main.cpp:178-181,188-206).  lens_radius == 0 selects the
    pinhole call (main.cpp:209), > 0 the thin-lens call (main.cpp:207)."""

    def __init__(self, cam=(0.0, 0.0, -10.0), half_width=10.0, focus_plane=20.0, lens_radius=0.0):
        self.cam = _v3(cam)
        self.half_width = float(half_width)
        self.focus_plane = float(focus_plane)
        self.lens_radius = float(lens_radius)
