"""cgraytracing_amd -- MI355X-native eye-ray pass of haoyuzhao123/CGRayTracing."""
from .scene import Bezier, Camera, Object, Plane, Sphere, Texture, TriangleMesh, Vec3  # noqa: F401

__all__ = ["Bezier", "Camera", "Object", "Plane", "Sphere", "Texture", "TriangleMesh", "Vec3", "Scene", "render", "tonemap_rgb8", "write_png"]


def __getattr__(name):
    # Scene / render need libcgrt.so; importing them lazily keeps `import cgraytracing_amd.scene` usable for
    # tools that only describe scenes.
    if name in ("Scene", "render", "tonemap_rgb8", "write_png"):
        import importlib
        _r = importlib.import_module(__name__ + ".engine")
        return getattr(_r, name)
    raise AttributeError(name)
