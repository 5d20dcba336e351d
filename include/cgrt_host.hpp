// cgrt_host.hpp -- header-only C++ host layer over the C ABI (cgrt.h).
//
// Keeps the reference's host-side API surface -- Vec3 (headers/vec3.h), Texture (headers/texture.h), Object /
// Sphere / Plane / TriangleMesh (headers/objects.h), Bezier (headers/bezier.h) with the SAME constructor
// signatures, and a render(objs) entry shaped like main.cpp:169 -- so that the scene-building part of a
// main() written against the reference compiles against this header unchanged.  Nothing is traced on the host
// (Object::intersect, too, runs on the device):
// render() flattens `objs` through cgrt_scene_add_* and launches the eye pass on the GPU with
// cgrt_trace_grid_host().  This file is original code; it mirrors interfaces, not implementations.
#ifndef CGRT_HOST_HPP
#define CGRT_HOST_HPP

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

#include "cgrt.h"

namespace cgrt_host {

// ---- vec3.h:11-119 ------------------------------------------------------------------------------
class Vec3 {
  public:
    double x, y, z;
    Vec3(double x_ = 0, double y_ = 0, double z_ = 0) : x(x_), y(y_), z(z_) {}
    double norm() const { return std::sqrt(x * x + y * y + z * z); }
    Vec3 normalize() {  // mutates and returns a copy, like the reference (vec3.h:36-44)
        double len = norm();
        if (len > 0) { x *= 1 / len; y *= 1 / len; z *= 1 / len; }
        return *this;
    }
    Vec3 copy() const { return Vec3(x, y, z); }
    Vec3 operator*(const double &f) const { return Vec3(x * f, y * f, z * f); }
    Vec3 operator*(const Vec3 &v) const { return Vec3(x * v.x, y * v.y, z * v.z); }
    Vec3 mul(const Vec3 &v) const { return Vec3(x * v.x, y * v.y, z * v.z); }
    double dot(const Vec3 &v) const { return x * v.x + y * v.y + z * v.z; }
    Vec3 operator+(const Vec3 &v) const { return Vec3(x + v.x, y + v.y, z + v.z); }
    Vec3 operator+(double b) const { return Vec3(x + b, y + b, z + b); }
    Vec3 operator-(const Vec3 &v) const { return Vec3(x - v.x, y - v.y, z - v.z); }
    Vec3 operator-(double b) const { return Vec3(x - b, y - b, z - b); }
    Vec3 operator-() const { return Vec3(-x, -y, -z); }
    Vec3 cross(const Vec3 &b) const { return Vec3(y * b.z - z * b.y, z * b.x - x * b.z, x * b.y - y * b.x); }
    void print() const { std::printf("x: %.6lf, y: %.6lf, z: %.6lf\n", x, y, z); }
    void get(double out[3]) const { out[0] = x; out[1] = y; out[2] = z; }
};
inline double det(const Vec3 &a, const Vec3 &b, const Vec3 &c) {
    return (a.x * b.y * c.z + b.x * c.y * a.z + c.x * a.y * b.z - a.x * c.y * b.z - b.x * a.y * c.z - c.x * b.y * a.z);
}
inline Vec3 matrixVectorProduct(const Vec3 &a, const Vec3 &b, const Vec3 &c, const Vec3 &d) {
    return a * d.x + b * d.y + c * d.z;
}
// adjugate / determinant, "singular" when |det| < 1e-4 (vec3.h:103-119)
inline bool inv(const Vec3 &a, const Vec3 &b, const Vec3 &c, Vec3 &resa, Vec3 &resb, Vec3 &resc) {
    const double d = det(a, b, c);
    if (d < 1e-4 && d > -1e-4) return false;
    resa = Vec3((b.y * c.z - b.z * c.y) / d, (c.y * a.z - c.z * a.y) / d, (a.y * b.z - a.z * b.y) / d);
    resb = Vec3((c.x * b.z - c.z * b.x) / d, (a.x * c.z - a.z * c.x) / d, (b.x * a.z - b.z * a.x) / d);
    resc = Vec3((b.x * c.y - c.x * b.y) / d, (c.x * a.y - c.y * a.x) / d, (a.x * b.y - a.y * b.x) / d);
    return true;
}

struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string &m) : std::runtime_error(m), code(c) {}
};
inline int check(int rc) {
    if (rc < 0) throw Error(rc, std::string("libcgrt: ") + cgrt_last_error());
    return rc;
}

// ---- texture.h:14-83 ------------------------------------------------------------------------------
// The reference builds `data` from 8-bit image bytes as byte/256 (main.cpp:303-316); the device stores the
// bytes, so every texel must be k/256 for an integer k in 0..255 (anything else is rejected).
class Texture {
  public:
    Texture() : rows(0), cols(0), lenx(0), leny(0), isbump(false), htexture(false) {}
    Texture(const std::vector<std::vector<Vec3> > &d, const Vec3 &n, const Vec3 &p, double lx, double ly,
            bool flag = false)
        : rows((int)d.size()), cols(d.empty() ? 0 : (int)d[0].size()), normal(n), position(p), lenx(lx), leny(ly),
          isbump(flag), htexture(true) {
        rgb.resize((size_t)rows * cols * 3);
        for (int i = 0; i < rows; i++)
            for (int j = 0; j < cols; j++) {
                const double c[3] = {d[i][j].x, d[i][j].y, d[i][j].z};
                for (int k = 0; k < 3; k++) {
                    const double b = c[k] * 256.0;
                    if (!(b >= 0 && b <= 255 && b == std::floor(b)))
                        throw Error(CGRT_ERR_INVALID, "Texture: texel is not byte/256");
                    rgb[3 * ((size_t)i * cols + j) + k] = (uint8_t)b;
                }
            }
    }
    // same object straight from decoder output (rows*cols*3 bytes)
    Texture(const uint8_t *bytes, int rows_, int cols_, const Vec3 &n, const Vec3 &p, double lx, double ly,
            bool flag = false)
        : rows(rows_), cols(cols_), rgb(bytes, bytes + (size_t)rows_ * cols_ * 3), normal(n), position(p), lenx(lx),
          leny(ly), isbump(flag), htexture(true) {}
    int rows, cols;
    std::vector<uint8_t> rgb;
    Vec3 normal, position;
    double lenx, leny;
    bool isbump, htexture;
};

// ---- objects.h:17-24 ------------------------------------------------------------------------------
class SceneBuilder;
class Object {
  public:
    Object() : probe_(nullptr) {}
    Object(const Object &) : probe_(nullptr) {}  // a copy builds its own probe scene when first asked
    Object &operator=(const Object &);           // the assigned-to object's cached probe scene is dropped
    virtual ~Object();
    // objects.h:20: nearest intersection of one ray with this object.  Runs ON THE DEVICE (this object alone, committed
    // once on `intersect_device` and cached; the function-level probe cgrt_intersect_rays) -- there is no host tracer.
    // len and normalvec are written only on a hit.  For many rays use intersect_batch.
    bool intersect(const Vec3 &rayorig, const Vec3 &raydir, double &len, Vec3 &normalvec) const;
    // n rays at once: org3 / dir3 are n x 3 doubles; hit, len, normal3 receive n, n and n x 3 values
    void intersect_batch(const double *org3, const double *dir3, int n, int32_t *hit, double *len, double *normal3) const;
    virtual double getTransparency() const = 0;
    virtual double getReflection() const = 0;
    virtual Vec3 getSurfaceColor(const Vec3 &) const = 0;  // flat colour; textured lookups happen on the device
    virtual int add_to(SceneBuilder &sb) const = 0;         // appends this object to a cgrt_scene
    int intersect_device = 0;

  private:
    mutable SceneBuilder *probe_;
};

class SceneBuilder {
  public:
    cgrt_scene *scene;
    std::map<const Texture *, int> tex_ids;
    SceneBuilder() : scene(nullptr) { check(cgrt_scene_create(&scene)); }
    ~SceneBuilder() { cgrt_scene_destroy(scene); }
    SceneBuilder(const SceneBuilder &) = delete;
    SceneBuilder &operator=(const SceneBuilder &) = delete;
    int texture(const Texture &t) {
        if (!t.htexture) return -1;
        auto it = tex_ids.find(&t);
        if (it != tex_ids.end()) return it->second;
        double n[3], p[3];
        t.normal.get(n);
        t.position.get(p);
        int id = check(cgrt_scene_add_texture(scene, t.rgb.data(), t.rows, t.cols, n, p, t.lenx, t.leny, t.isbump));
        tex_ids[&t] = id;
        return id;
    }
};

inline Object::~Object() { delete probe_; }
inline Object &Object::operator=(const Object &) {
    delete probe_;
    probe_ = nullptr;
    return *this;
}
inline void Object::intersect_batch(const double *org3, const double *dir3, int n, int32_t *hit, double *len,
                                    double *normal3) const {
    if (!probe_) {
        SceneBuilder *sb = new SceneBuilder();
        try {
            add_to(*sb);
            check(cgrt_scene_commit(sb->scene, intersect_device));
        } catch (...) {
            delete sb;
            throw;
        }
        probe_ = sb;
    }
    check(cgrt_intersect_rays(probe_->scene, 0, org3, dir3, nullptr, n, hit, len, normal3));
}
inline bool Object::intersect(const Vec3 &rayorig, const Vec3 &raydir, double &len, Vec3 &normalvec) const {
    double o[3], d[3], l = 0, nv[3] = {0, 0, 0};
    int32_t h = 0;
    rayorig.get(o);
    raydir.get(d);
    intersect_batch(o, d, 1, &h, &l, nv);
    if (h) {
        len = l;
        normalvec = Vec3(nv[0], nv[1], nv[2]);
    }
    return h != 0;
}

// Sphere(c, r, sc, refl, transp, ec)   objects.h:28-38
class Sphere : public Object {
  public:
    Sphere(const Vec3 &c, const double &r, const Vec3 &sc, const double &refl = 0, const double &transp = 0,
           const Vec3 & /*ec*/ = 0)
        : center(c), radius(r), surfaceColor(sc), transparency(transp), reflection(refl) {}
    double getTransparency() const override { return transparency; }
    double getReflection() const override { return reflection; }
    Vec3 getSurfaceColor(const Vec3 &) const override { return surfaceColor.copy(); }
    int add_to(SceneBuilder &sb) const override {
        double c[3], s[3];
        center.get(c);
        surfaceColor.get(s);
        return check(cgrt_scene_add_sphere(sb.scene, c, radius, s, reflection, transparency));
    }
  private:
    Vec3 center;
    double radius;
    Vec3 surfaceColor;
    double transparency, reflection;
};

// Plane(p, n, sc, refl, transp, tx, ec)   objects.h:480
class Plane : public Object {
  public:
    Plane(const Vec3 &p, const Vec3 &n, const Vec3 &sc, const double &refl = 0, const double &transp = 0,
          const Texture &tx = Texture(), const Vec3 & /*ec*/ = 0)
        : position(p), normal(n), surfaceColor(sc), transparency(transp), reflection(refl), texture(tx) {}
    double getTransparency() const override { return transparency; }
    double getReflection() const override { return reflection; }
    Vec3 getSurfaceColor(const Vec3 &) const override { return surfaceColor.copy(); }
    int add_to(SceneBuilder &sb) const override {
        double p[3], n[3], s[3];
        position.get(p);
        normal.get(n);
        surfaceColor.get(s);
        return check(cgrt_scene_add_plane(sb.scene, p, n, s, reflection, transparency, sb.texture(texture)));
    }
  private:
    Vec3 position, normal, surfaceColor;
    double transparency, reflection;
    Texture texture;
};

// TriangleMesh(filename, a, b, sc, refl, transp, typeofdata, ec)   objects.h:338-340
class TriangleMesh : public Object {
  public:
    TriangleMesh(const char *filename, double a_, const Vec3 &b_, const Vec3 &sc, const double &refl = 0,
                 const double &transp = 0, int typeofdata = 0, const Vec3 & /*ec*/ = 0)
        : file(filename ? filename : ""), a(a_), b(b_), surfaceColor(sc), transparency(transp), reflection(refl),
          objtype(typeofdata) {}
    double getTransparency() const override { return transparency; }
    double getReflection() const override { return reflection; }
    Vec3 getSurfaceColor(const Vec3 &) const override { return surfaceColor.copy(); }
    int add_to(SceneBuilder &sb) const override {
        double bb[3], s[3];
        b.get(bb);
        surfaceColor.get(s);
        return check(cgrt_scene_add_mesh_file(sb.scene, file.c_str(), a, bb, s, reflection, transparency, objtype));
    }
  private:
    std::string file;
    double a;
    Vec3 b, surfaceColor;
    double transparency, reflection;
    int objtype;
};

// Bezier(points, pos, sc, refl, transp, typeofdata, ec)   bezier.h:44-45
class Bezier : public Object {
  public:
    Bezier(std::vector<Vec3> points, const Vec3 &pos, const Vec3 &sc, const double &refl = 0,
           const double &transp = 0, int /*typeofdata*/ = 0, const Vec3 & /*ec*/ = 0)
        : cpoints(points), position(pos), surfaceColor(sc), transparency(transp), reflection(refl) {}
    double getTransparency() const override { return transparency; }
    double getReflection() const override { return reflection; }
    Vec3 getSurfaceColor(const Vec3 &) const override { return surfaceColor.copy(); }
    int add_to(SceneBuilder &sb) const override {
        std::vector<double> cp;
        for (const Vec3 &v : cpoints) { cp.push_back(v.x); cp.push_back(v.y); cp.push_back(v.z); }
        double p[3], s[3];
        position.get(p);
        surfaceColor.get(s);
        return check(cgrt_scene_add_bezier(sb.scene, cp.data(), (int)cpoints.size(), p, s, reflection, transparency));
    }
  private:
    std::vector<Vec3> cpoints;
    Vec3 position, surfaceColor;
    double transparency, reflection;
};

// ---- render(), main.cpp:169-219 -------------------------------------------------------------------
// The reference's compile-time constants and render() locals, as runtime fields with the same defaults.
struct RenderParams {
    int width = 1024, height = 768;       // main.cpp:28-29
    int num_of_samples = 1;               // main.cpp:177
    double focus_plane = 20.0;            // main.cpp:178
    double radius = 1.5;                  // main.cpp:179 (lens radius)
    Vec3 camorg = Vec3(0, 0, -10);        // main.cpp:181
    int max_depth = 5;                    // MAX_DEPTH, main.cpp:35
    bool depth_of_field = false;          // false: the committed pinhole call (main.cpp:209); true: main.cpp:207
    uint64_t seed = 12345;
    int device = 0;
};
struct RenderStats {
    uint64_t rays = 0, hitpoints = 0;
};

// Eye pass of render(objs): image[h][w] (row 0 = bottom, main.cpp:185-189) receives, per channel, the sum of
// hp.f over the pixel's hitpoints divided by num_of_samples, as float RGB.  (For the photon pass and tone mapping
// that follow in the reference, main.cpp:223-258, see render_ppm below.)
inline void render(const std::vector<Object *> &objs, const RenderParams &rp, std::vector<float> &image,
                   RenderStats *stats = nullptr) {
    SceneBuilder sb;
    for (const Object *o : objs) o->add_to(sb);
    check(cgrt_scene_commit(sb.scene, rp.device));
    cgrt_camera cam;
    rp.camorg.get(cam.cam);
    cam.half_width = 10.0;
    cam.focus_plane = rp.focus_plane;
    cam.lens_radius = rp.depth_of_field ? rp.radius : 0.0;
    cgrt_grid g{};
    g.width = rp.width;
    g.height = rp.height;
    g.rows = rp.height;
    g.stripe_nranks = 1;
    g.spp = rp.num_of_samples;
    g.spp_total = rp.num_of_samples;
    g.max_depth = rp.max_depth;
    g.seed = rp.seed;
    image.assign((size_t)rp.width * rp.height * 3, 0.f);
    uint64_t cnt[CGRT_NCOUNTERS] = {0};
    check(cgrt_trace_grid_host(sb.scene, &cam, &g, image.data(), nullptr, cnt));
    if (stats) {
        stats->rays = cnt[CGRT_CNT_RAYS];
        stats->hitpoints = cnt[CGRT_CNT_HITPOINTS];
    }
}

// ---- the whole of render() + main()'s PNG loop: main.cpp:169-258, 403-412 ---------------------------------------
// Photon-pass constants of the reference as runtime fields with the same defaults.
struct PhotonParams {
    Vec3 lightorg = Vec3(0, 19.999, 20);   // main.cpp:180
    double jitter = 2.0;                   // main.cpp:240-241 (u*4-2)
    double power = 700.0;                  // main.cpp:246
    double alpha = 0.7;                    // main.cpp:36
    long long num_photon = 2560000;        // main.cpp:223
    int num_threads = 8;                   // main.cpp:224: total photons = num_photon * num_threads
    int hashsize = 1000001;                // main.cpp:184
    uint64_t seed = 777;
    double initial_radius = 0.0;           // main.cpp:84,183: 200.0 / height with the reference's compile-time height;
                                           // 0 = the committed 200.0 / 768
};
struct PpmStats {
    uint64_t hitpoints = 0, photon_events = 0;
    double ms_eye = 0, ms_table = 0, ms_photons = 0, ms_gather = 0;
};

// render(objs) as the reference runs it: eye pass, photon pass (serial semantics: photon i on the keyed stream
// (seed, i); the reference's racing OpenMP threads over time-seeded rand() have no reproducible meaning), final
// gather into image[h][w] (doubles, row 0 = bottom, main.cpp:252-258) and the tone-mapped, flipped bytes that
// main.cpp:403-411 hands to stbi_write_png.
inline void render_ppm(const std::vector<Object *> &objs, const RenderParams &rp, const PhotonParams &pp,
                       std::vector<double> &image, std::vector<unsigned char> &image_data, PpmStats *stats = nullptr) {
    SceneBuilder sb;
    for (const Object *o : objs) o->add_to(sb);
    check(cgrt_scene_commit(sb.scene, rp.device));
    cgrt_camera cam;
    rp.camorg.get(cam.cam);
    cam.half_width = 10.0;
    cam.focus_plane = rp.focus_plane;
    cam.lens_radius = rp.depth_of_field ? rp.radius : 0.0;
    cgrt_grid g{};
    g.width = rp.width;
    g.height = rp.height;
    g.rows = rp.height;
    g.stripe_nranks = 1;
    g.spp = rp.num_of_samples;
    g.spp_total = rp.num_of_samples;
    g.max_depth = rp.max_depth;
    g.seed = rp.seed;
    cgrt_photons ph{};
    pp.lightorg.get(ph.light);
    ph.jitter = pp.jitter;
    ph.power = pp.power;
    ph.alpha = pp.alpha;
    ph.nphotons = pp.num_photon * pp.num_threads;
    ph.hashsize = pp.hashsize;
    ph.seed = pp.seed;
    ph.initial_radius = pp.initial_radius;  // 0: the reference's committed 200.0 / 768 (main.cpp:84,183)
    image.assign((size_t)rp.width * rp.height * 3, 0.0);
    image_data.assign((size_t)rp.width * rp.height * 3, 0);
    cgrt_ppm_result out{};
    out.image = image.data();
    out.rgb8 = image_data.data();
    check(cgrt_ppm_render(sb.scene, &cam, &g, &ph, &out));
    if (stats) {
        stats->hitpoints = out.hp_count;
        stats->photon_events = out.n_events;
        stats->ms_eye = out.ms_eye; stats->ms_table = out.ms_table;
        stats->ms_photons = out.ms_photons; stats->ms_gather = out.ms_gather;
    }
}

// stbi_write_png("test.png", width, height, 3, image_data, width * 3), main.cpp:412
inline void write_png(const char *path, int width, int height, const std::vector<unsigned char> &image_data) {
    check(cgrt_write_png(path, width, height, image_data.data()));
}

}  // namespace cgrt_host
#endif
