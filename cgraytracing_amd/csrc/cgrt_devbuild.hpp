// Row f3 of SURVEY.md section 8: acceleration structures built ON THE DEVICE (cgrt_scene_set_build(CGRT_BUILD_DEVICE) or
// CGRT_BUILD=device).  Included by cgrt_hip.hip.  Two parts, both for OPAQUE owners (a transparent owner's improvement
// counter makes the reference's std::sort leaf order observable, quirk Q5: those keep the host build, cgrt_build.cpp):
//
//   (i)  bump floors -- Texture's height field (texture.h:28-35) and Plane's displacement mesh (objects.h:485-497) are
//        generated from the texture BYTES by two kernels: vertex heights, then one HCellRec (both triangles of a quad) per
//        cell in grid order plus the same TriRecs in construction order.  The device walks such a floor as a grid
//        (hfield_intersect), so no tree is needed at all.
//   (ii) opaque meshes -- the triangle-level hierarchy of tree_intersect_wide is built from the triangle soup: Morton
//        codes of the triangles' box centres, one radix sort (rocprim), a binary radix tree over the sorted codes
//        (Karras 2012: every inner node finds its own range and split, no recursion), boxes fitted bottom-up, groups of
//        <= 4 triangles as leaves, then the 4-wide form level by level with the host build's rule (replace the
//        larger-surfaced inner child by its children until there are four).  The walk's stack bound (kWideStack) is checked
//        on the device; a tree that misses it is rebuilt with balanced splits of the sorted order (depth <= log2 n).
//
// Parity class: TOLERANCE, not bit-exact.  Boxes are supersets and the triangle test is the same
// code, so (len, triangle) of every ray with a UNIQUE nearest hit is bit-identical to the host build's.  What differs:
// exact ties (a ray through a shared edge or vertex) go to the triangle with the lower CONSTRUCTION index instead of the
// reference's leaf-order rule (objects.h:281,297) -- the two candidates share the hit point but not the normal --, and
// heights come from a correctly rounded exp (cgrt_ddexp.hpp) where the reference's libm is within 0.52 ulp.
#ifndef CGRT_DEVBUILD_HPP
#define CGRT_DEVBUILD_HPP
#include <rocprim/device/device_radix_sort.hpp>

#include "cgrt_ddexp.hpp"

namespace devbuild {

// order-preserving map double -> uint64 (atomicMin / atomicMax on doubles of either sign)
__host__ __device__ inline unsigned long long dkey(double v) {
    union { double d; unsigned long long u; } c;
    c.d = v;
    return (c.u >> 63) ? ~c.u : (c.u | 0x8000000000000000ull);
}
__host__ __device__ inline double dunkey(unsigned long long k) {
    union { double d; unsigned long long u; } c;
    c.u = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
    return c.d;
}
__device__ inline void atomic_min_max(unsigned long long *mm, double lo, double hi) {
    atomicMin(mm, dkey(lo));
    atomicMax(mm + 1, dkey(hi));
}
__device__ inline float f_down(double v) { return __double2float_rd(v); }  // largest float <= v (cgrt_build.cpp round_down)
__device__ inline float f_up(double v) { return __double2float_ru(v); }

// =====================================================================================================
// (i) bump floor
// =====================================================================================================
struct BumpArgs {
    const uint8_t *texels;  // rows x cols x 3
    int R, C;
    double p0, p2, lenx, leny, plane_y;
    int nx, nz;  // cells along x / z
    double *ys;  // (nz + 1) x (nx + 1) vertex heights (plane y included)
    unsigned long long *minmax;  // [0] = key of the lowest, [1] = of the highest vertex
    HCellRec *cells;
    HCellY *celly;
    TriRec *tris;
};

// texture.h:28-35 for the texels the mesh uses (every third row and column), objects.h:489-492: y = height + plane.y
__global__ void bump_vertex_kernel(BumpArgs a) {
    const int nv = (a.nx + 1) * (a.nz + 1);
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= nv) return;
    const int i = v / (a.nx + 1), j = v % (a.nx + 1);
    const uint8_t *q = a.texels + 3 * ((size_t)(i * 3) * a.C + (size_t)j * 3);
    const double luma = (0.299 * ((double)q[0] / 256.0) + 0.587 * ((double)q[1] / 256.0) + 0.114 * ((double)q[2] / 256.0));
    const double h = 1 - cgrt_dd::exp_dd(-3.3 * luma);
    const double y = h * 0.5 + a.plane_y;
    a.ys[v] = y;
    atomic_min_max(a.minmax, y, y);
}

// objects.h:485-497: quad (i, j) -> triangles (a, b, c) and (d, b, c), the host build's expressions term for term
__global__ void bump_cell_kernel(BumpArgs a) {
    const int nc = a.nx * a.nz;
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= nc) return;
    const int i = c / a.nx, j = c % a.nx, step = 3;
    const double x1 = a.p0 + a.lenx * j * step / a.C;
    const double x2 = a.p0 + a.lenx * (j + 1) * step / a.C;
    const double z1 = a.p2 + a.leny * i * step / a.R;
    const double z2 = a.p2 + a.leny * (i + 1) * step / a.R;
    const int nxv = a.nx + 1;
    const double ya = a.ys[i * nxv + j], yb = a.ys[i * nxv + j + 1], yc = a.ys[(i + 1) * nxv + j], yd = a.ys[(i + 1) * nxv + j + 1];
    const double A[3] = {x1, ya, z1}, B[3] = {x2, yb, z1}, Cc[3] = {x1, yc, z2}, D[3] = {x2, yd, z2};
    HCellRec cell;
    for (int k = 0; k < 3; k++) {
        cell.t[0].pa[k] = A[k];
        cell.t[0].e1[k] = A[k] - B[k];
        cell.t[0].e2[k] = A[k] - Cc[k];
        cell.t[1].pa[k] = D[k];
        cell.t[1].e1[k] = D[k] - B[k];
        cell.t[1].e2[k] = D[k] - Cc[k];
    }
    // ties: no reference leaf order here -- the triangle with the lower construction index wins (header comment)
    cell.k[0] = 2 * c;
    cell.k[1] = 2 * c + 1;
    cell.leaf[0] = 0;
    cell.leaf[1] = 0;
    a.cells[c] = cell;
    HCellY cy;
    cy.lo = f_down(fmin(fmin(ya, yb), fmin(yc, yd)));
    cy.hi = f_up(fmax(fmax(ya, yb), fmax(yc, yd)));
    a.celly[c] = cy;
    a.tris[2 * c] = cell.t[0];
    a.tris[2 * c + 1] = cell.t[1];
}

// =====================================================================================================
// (ii) opaque mesh: LBVH -> groups of <= 4 -> 4-wide
// =====================================================================================================
struct MeshArgs {
    const double *tri9;  // n x 9
    int n;
    TriRec *tris;        // n, construction order
    OTriRec *otris;      // n, hierarchy order
    WideNodeRec *wnodes; // capacity n
    unsigned long long *gbox;  // 6 keys: min x,y,z then max x,y,z of all vertices (dkey), [6] = max squared vertex distance key
    unsigned long long *keys_in, *keys_out;
    unsigned int *idx_in, *idx_out;
    // binary tree: inner node i in [0, n-1), leaf j = n - 1 + j
    int *left, *right, *first, *last, *parent, *flag;
    double *box;  // (2n - 1) x 6: lo(3), hi(3)
    int2 *queue;  // wide-node work list: {binary node, stack entries waiting above it}
    int *tail;    // [0] = wide nodes so far, [1] = stack need, [2] = root (binary node id)
};

__device__ inline void tri_box(const double *t, double lo[3], double hi[3]) {
    for (int k = 0; k < 3; k++) {
        lo[k] = fmin(t[k], fmin(t[3 + k], t[6 + k]));
        hi[k] = fmax(t[k], fmax(t[3 + k], t[6 + k]));
    }
}

// TriRec per triangle (objects.h:98-99: e1 = pa - pb, e2 = pa - pc) and the box of all vertices
__global__ void mesh_prep_kernel(MeshArgs a) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n) return;
    const double *t = a.tri9 + 9 * (size_t)i;
    TriRec tr;
    for (int k = 0; k < 3; k++) {
        tr.pa[k] = t[k];
        tr.e1[k] = t[k] - t[3 + k];
        tr.e2[k] = t[k] - t[6 + k];
    }
    a.tris[i] = tr;
    double lo[3], hi[3];
    tri_box(t, lo, hi);
    for (int k = 0; k < 3; k++) {
        atomicMin(a.gbox + k, dkey(lo[k]));
        atomicMax(a.gbox + 3 + k, dkey(hi[k]));
    }
}

__device__ inline unsigned long long spread21(unsigned long long v) {  // 21 bits -> every third bit
    v &= 0x1fffffull;
    v = (v | (v << 32)) & 0x1f00000000ffffull;
    v = (v | (v << 16)) & 0x1f0000ff0000ffull;
    v = (v | (v << 8)) & 0x100f00f00f00f00full;
    v = (v | (v << 4)) & 0x10c30c30c30c30c3ull;
    v = (v | (v << 2)) & 0x1249249249249249ull;
    return v;
}

// Morton code of the triangle's box centre in the box of all vertices; also the largest squared vertex distance from that
// box's centre (the scene walk's bounding sphere, cgrt_build.cpp add_mesh_triangles)
__global__ void mesh_morton_kernel(MeshArgs a) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n) return;
    double glo[3], ghi[3];
    for (int k = 0; k < 3; k++) {
        glo[k] = dunkey(a.gbox[k]);
        ghi[k] = dunkey(a.gbox[3 + k]);
    }
    const double *t = a.tri9 + 9 * (size_t)i;
    double lo[3], hi[3];
    tri_box(t, lo, hi);
    unsigned long long code = 0;
    for (int k = 0; k < 3; k++) {
        const double ext = ghi[k] - glo[k];
        double q = ext > 0 ? (0.5 * (lo[k] + hi[k]) - glo[k]) / ext * 2097152.0 : 0.0;
        q = q < 0 ? 0 : (q > 2097151.0 ? 2097151.0 : q);
        code |= spread21((unsigned long long)q) << k;
    }
    a.keys_in[i] = code;
    a.idx_in[i] = (unsigned int)i;
    const double c[3] = {0.5 * (glo[0] + ghi[0]), 0.5 * (glo[1] + ghi[1]), 0.5 * (glo[2] + ghi[2])};
    double r2 = 0;
    for (int v = 0; v < 3; v++) {
        const double dx = t[3 * v] - c[0], dy = t[3 * v + 1] - c[1], dz = t[3 * v + 2] - c[2];
        r2 = fmax(r2, dx * dx + dy * dy + dz * dz);
    }
    atomicMax(a.gbox + 6, dkey(r2));
}

// length of the common prefix of sorted keys i and j (equal keys: continued on the index bits); -1 outside the array
__device__ inline int lcp(const unsigned long long *keys, int n, int i, int j) {
    if (j < 0 || j >= n) return -1;
    const unsigned long long x = keys[i] ^ keys[j];
    return x ? __clzll((long long)x) : 64 + __clz(i ^ j);
}

// Karras 2012, "Maximizing parallelism in the construction of BVHs, octrees and k-d trees", section 4: inner node i
__global__ void lbvh_nodes_kernel(MeshArgs a) {
    const int n = a.n;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n - 1) return;
    const unsigned long long *keys = a.keys_out;
    const int d = (lcp(keys, n, i, i + 1) - lcp(keys, n, i, i - 1)) >= 0 ? 1 : -1;
    const int dmin = lcp(keys, n, i, i - d);
    int lmax = 2;
    while (lcp(keys, n, i, i + lmax * d) > dmin) lmax *= 2;
    int l = 0;
    for (int t = lmax / 2; t >= 1; t /= 2)
        if (lcp(keys, n, i, i + (l + t) * d) > dmin) l += t;
    const int j = i + l * d;
    const int dnode = lcp(keys, n, i, j);
    int s = 0, t = l;
    do {
        t = (t + 1) / 2;
        if (lcp(keys, n, i, i + (s + t) * d) > dnode) s += t;
    } while (t > 1);
    const int gamma = i + s * d + (d < 0 ? d : 0);
    const int lo = i < j ? i : j, hi = i < j ? j : i;
    const int L = (lo == gamma) ? n - 1 + gamma : gamma;
    const int Rr = (hi == gamma + 1) ? n - 1 + gamma + 1 : gamma + 1;
    a.left[i] = L;
    a.right[i] = Rr;
    a.first[i] = lo;
    a.last[i] = hi;
    a.parent[L] = i;
    a.parent[Rr] = i;
    if (i == 0) {
        a.parent[0] = -1;
        a.tail[2] = 0;
    }
}

// The fallback: balanced splits of the sorted order -- the inner node that splits between sorted triangles s and s + 1 is
// node s; its range is found by descending from the root (<= log2 n steps)
__global__ void balanced_nodes_kernel(MeshArgs a) {
    const int n = a.n;
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n - 1) return;
    int b = 0, e = n - 1, par = -1, mid;
    while (true) {
        mid = (b + e) / 2;
        if (mid == s) break;
        par = mid;
        if (s < mid) e = mid;
        else b = mid + 1;
    }
    const int L = (b == mid) ? n - 1 + b : (b + mid) / 2;
    const int Rr = (mid + 1 == e) ? n - 1 + e : (mid + 1 + e) / 2;
    a.left[s] = L;
    a.right[s] = Rr;
    a.first[s] = b;
    a.last[s] = e;
    a.parent[s] = par;
    if (L >= n - 1) a.parent[L] = s;
    if (Rr >= n - 1) a.parent[Rr] = s;
    if (par < 0) a.tail[2] = s;
}

// boxes bottom-up: one thread per sorted triangle; the second thread to reach an inner node fits it and goes on
__global__ void lbvh_fit_kernel(MeshArgs a) {
    const int n = a.n;
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const unsigned int src = a.idx_out[j];
    const double *t = a.tri9 + 9 * (size_t)src;
    double lo[3], hi[3];
    tri_box(t, lo, hi);
    double *bx = a.box + 6 * (size_t)(n - 1 + j);
    for (int k = 0; k < 3; k++) {
        bx[k] = lo[k];
        bx[3 + k] = hi[k];
    }
    // the hierarchy's own triangle order: record + construction index (ties: lower index wins, see the header comment)
    OTriRec o;
    o.t = a.tris[src];
    o.k = (int32_t)src;
    o.leaf = 0;
    a.otris[j] = o;
    if (n == 1) {
        a.tail[2] = 0;  // the root is leaf 0 (id n - 1 + 0 = 0)
        return;
    }
    int node = a.parent[n - 1 + j];
    while (node >= 0) {
        __threadfence();
        if (atomicAdd(&a.flag[node], 1) == 0) return;  // the sibling subtree is not fitted yet: its thread will come by
        __threadfence();
        const volatile double *lb = a.box + 6 * (size_t)a.left[node], *rb = a.box + 6 * (size_t)a.right[node];
        double *nb = a.box + 6 * (size_t)node;
        for (int k = 0; k < 3; k++) {
            nb[k] = fmin(lb[k], rb[k]);
            nb[3 + k] = fmax(lb[3 + k], rb[3 + k]);
        }
        node = a.parent[node];
    }
}

__device__ inline bool group_leaf(const MeshArgs &a, int node, int &first, int &cnt) {
    const int n = a.n;
    if (node >= n - 1) {
        first = node - (n - 1);
        cnt = 1;
        return true;
    }
    first = a.first[node];
    cnt = a.last[node] - first + 1;
    return cnt <= 4;  // HostTree::build_bvh: max_leaf = 4
}

// One level of the 4-wide form (BvhBuilder::emit_wide's rule): queue entries [begin, end) become wnodes[begin, end); their
// inner children are appended to the queue for the next launch.
__global__ void wide_level_kernel(MeshArgs a, int begin, int end) {
    const int q = begin + blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= end) return;
    const int node = a.queue[q].x, acc = a.queue[q].y;
    int kids[4], nk = 0, f, c;
    if (group_leaf(a, node, f, c)) {
        kids[nk++] = node;  // a tree that is one leaf: a root with that one child
    } else {
        kids[nk++] = a.left[node];
        kids[nk++] = a.right[node];
        while (nk < 4) {
            int best = -1;
            double best_area = -1;
            for (int k = 0; k < nk; k++) {
                if (group_leaf(a, kids[k], f, c)) continue;
                const double *b = a.box + 6 * (size_t)kids[k];
                const double x = b[3] - b[0], y = b[4] - b[1], z = b[5] - b[2];
                const double area = x * y + y * z + z * x;
                if (area > best_area) {
                    best_area = area;
                    best = k;
                }
            }
            if (best < 0) break;
            const int cnode = kids[best];
            kids[best] = a.left[cnode];
            kids[nk++] = a.right[cnode];
        }
    }
    WideNodeRec w;
    for (int k = 0; k < 4; k++) {
        w.pad[k] = 0;
        if (k >= nk) {
            w.lox[k] = w.loy[k] = w.loz[k] = w.hix[k] = w.hiy[k] = w.hiz[k] = 0.f;
            w.ref[k] = kWideNone;
            continue;
        }
        const double *b = a.box + 6 * (size_t)kids[k];
        w.lox[k] = f_down(b[0] - kBoxPad);
        w.loy[k] = f_down(b[1] - kBoxPad);
        w.loz[k] = f_down(b[2] - kBoxPad);
        w.hix[k] = f_up(b[3] + kBoxPad);
        w.hiy[k] = f_up(b[4] + kBoxPad);
        w.hiz[k] = f_up(b[5] + kBoxPad);
        if (group_leaf(a, kids[k], f, c)) {
            w.ref[k] = (int32_t)(((uint32_t)f << 4) | (uint32_t)c);
        } else {
            const int child = atomicAdd(&a.tail[0], 1);
            a.queue[child] = make_int2(kids[k], acc + nk - 1);
            w.ref[k] = ~child;
        }
    }
    atomicMax(&a.tail[1], acc + nk - 1);
    a.wnodes[q] = w;
}

// Cover spheres for classify_kernel (cgrt_build.cpp cover_spheres): group g = sorted triangles [g n / G, (g + 1) n / G) --
// contiguous in Morton order, hence compact --, sphere = (centre of the group's vertex box, largest vertex distance + 1e-3).
// One workgroup per group.
__global__ __launch_bounds__(256) void cover_kernel(MeshArgs a, int G, double *cover) {
    const int g = blockIdx.x, n = a.n;
    const int b = (int)((long long)g * n / G), e = (int)((long long)(g + 1) * n / G);
    __shared__ double red[256][6];
    double lo[3] = {kInf, kInf, kInf}, hi[3] = {-kInf, -kInf, -kInf};
    for (int j = b + threadIdx.x; j < e; j += 256) {
        double l[3], h[3];
        tri_box(a.tri9 + 9 * (size_t)a.idx_out[j], l, h);
        for (int k = 0; k < 3; k++) {
            lo[k] = fmin(lo[k], l[k]);
            hi[k] = fmax(hi[k], h[k]);
        }
    }
    for (int k = 0; k < 3; k++) {
        red[threadIdx.x][k] = lo[k];
        red[threadIdx.x][3 + k] = hi[k];
    }
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off)
            for (int k = 0; k < 3; k++) {
                red[threadIdx.x][k] = fmin(red[threadIdx.x][k], red[threadIdx.x + off][k]);
                red[threadIdx.x][3 + k] = fmax(red[threadIdx.x][3 + k], red[threadIdx.x + off][3 + k]);
            }
        __syncthreads();
    }
    const double c[3] = {0.5 * (red[0][0] + red[0][3]), 0.5 * (red[0][1] + red[0][4]), 0.5 * (red[0][2] + red[0][5])};
    __syncthreads();
    double r2 = 0;
    for (int j = b + threadIdx.x; j < e; j += 256) {
        const double *t = a.tri9 + 9 * (size_t)a.idx_out[j];
        for (int v = 0; v < 3; v++) {
            const double dx = t[3 * v] - c[0], dy = t[3 * v + 1] - c[1], dz = t[3 * v + 2] - c[2];
            r2 = fmax(r2, dx * dx + dy * dy + dz * dz);
        }
    }
    red[threadIdx.x][0] = r2;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) red[threadIdx.x][0] = fmax(red[threadIdx.x][0], red[threadIdx.x + off][0]);
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        cover[4 * g] = c[0];
        cover[4 * g + 1] = c[1];
        cover[4 * g + 2] = c[2];
        cover[4 * g + 3] = sqrt(red[0][0]) + 1e-3;
    }
}

// ---- host drivers --------------------------------------------------------------------------------------------------

struct BumpResult {
    double ylo = 0, yhi = 0;
};
// texels: device pointer to this texture's bytes; cells / tris: where the records go (device)
static int build_bump_floor(const uint8_t *texels, int R, int C, const double p[3], double lenx, double leny, double plane_y,
                            HCellRec *cells, HCellY *celly, TriRec *tris, BumpResult &out) {
    BumpArgs a;
    a.texels = texels;
    a.R = R;
    a.C = C;
    a.p0 = p[0];
    a.p2 = p[2];
    a.lenx = lenx;
    a.leny = leny;
    a.plane_y = plane_y;
    a.nx = C / 3 - 1;
    a.nz = R / 3 - 1;
    a.cells = cells;
    a.celly = celly;
    a.tris = tris;
    const int nv = (a.nx + 1) * (a.nz + 1), nc = a.nx * a.nz;
    DevBuf ys, mm;
    HIP_TRY(ys.alloc((size_t)nv * sizeof(double)));
    HIP_TRY(mm.alloc(2 * sizeof(unsigned long long)));
    const unsigned long long init[2] = {~0ull, 0ull};
    HIP_TRY(hipMemcpy(mm.p, init, sizeof(init), hipMemcpyHostToDevice));
    a.ys = ys.as<double>();
    a.minmax = mm.as<unsigned long long>();
    hipLaunchKernelGGL(bump_vertex_kernel, dim3((nv + 255) / 256), dim3(256), 0, 0, a);
    hipLaunchKernelGGL(bump_cell_kernel, dim3((nc + 255) / 256), dim3(256), 0, 0, a);
    HIP_TRY(hipGetLastError());
    unsigned long long got[2];
    HIP_TRY(hipMemcpy(got, mm.p, sizeof(got), hipMemcpyDeviceToHost));  // synchronises
    out.ylo = dunkey(got[0]);
    out.yhi = dunkey(got[1]);
    return CGRT_OK;
}

struct MeshResult {
    int nwide = 0, stack_need = 0;
    bool balanced = false;  // the fallback build was needed
    double centre[3] = {0, 0, 0}, r2 = -1.0;  // ObjRec.a / s0 (bounding sphere of the vertices, grown like the host's)
    double bmax = 0;  // largest |coordinate| of a vertex (TreeRec::bmax)
    std::vector<double> cover;  // n_groups x (cx, cy, cz, r)
};
// tri9_host: n x 9 doubles; tris / otris / wnodes: device destinations (n, n, n records)
static int build_mesh_hierarchy(const double *tri9_host, int n, TriRec *tris, OTriRec *otris, WideNodeRec *wnodes, MeshResult &out) {
    out = MeshResult();
    if (n <= 0) return CGRT_OK;
    const size_t N = (size_t)n;
    DevBuf tri9, gbox, k0, k1, i0, i1, tree, box, queue, tail, sort_tmp, cover;
    HIP_TRY(tri9.alloc(N * 9 * sizeof(double)));
    HIP_TRY(hipMemcpy(tri9.p, tri9_host, N * 9 * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(gbox.alloc(7 * sizeof(unsigned long long)));
    const unsigned long long ginit[7] = {~0ull, ~0ull, ~0ull, 0ull, 0ull, 0ull, 0ull};
    HIP_TRY(hipMemcpy(gbox.p, ginit, sizeof(ginit), hipMemcpyHostToDevice));
    HIP_TRY(k0.alloc(N * 8));
    HIP_TRY(k1.alloc(N * 8));
    HIP_TRY(i0.alloc(N * 4));
    HIP_TRY(i1.alloc(N * 4));
    // left, right, first, last, flag: n - 1 each; parent: 2n - 1
    HIP_TRY(tree.alloc((5 * N + 2 * N) * sizeof(int)));
    HIP_TRY(box.alloc((2 * N) * 6 * sizeof(double)));
    HIP_TRY(queue.alloc(N * sizeof(int2)));
    HIP_TRY(tail.alloc(4 * sizeof(int)));
    MeshArgs a;
    a.tri9 = tri9.as<double>();
    a.n = n;
    a.tris = tris;
    a.otris = otris;
    a.wnodes = wnodes;
    a.gbox = gbox.as<unsigned long long>();
    a.keys_in = k0.as<unsigned long long>();
    a.keys_out = k1.as<unsigned long long>();
    a.idx_in = i0.as<unsigned int>();
    a.idx_out = i1.as<unsigned int>();
    int *tp = tree.as<int>();
    a.left = tp;
    a.right = tp + N;
    a.first = tp + 2 * N;
    a.last = tp + 3 * N;
    a.flag = tp + 4 * N;
    a.parent = tp + 5 * N;
    a.box = box.as<double>();
    a.queue = queue.as<int2>();
    a.tail = tail.as<int>();
    const dim3 blk(256), grd((unsigned)((N + 255) / 256));
    hipLaunchKernelGGL(mesh_prep_kernel, grd, blk, 0, 0, a);
    hipLaunchKernelGGL(mesh_morton_kernel, grd, blk, 0, 0, a);
    size_t tmp_bytes = 0;
    HIP_TRY(rocprim::radix_sort_pairs(nullptr, tmp_bytes, a.keys_in, a.keys_out, a.idx_in, a.idx_out, N, 0u, 63u, (hipStream_t)0));
    HIP_TRY(sort_tmp.alloc(tmp_bytes));
    HIP_TRY(rocprim::radix_sort_pairs(sort_tmp.p, tmp_bytes, a.keys_in, a.keys_out, a.idx_in, a.idx_out, N, 0u, 63u, (hipStream_t)0));
    // CGRT_DEVBUILD_BALANCED=1 (test aid, read at every build): skip the radix tree and take the fallback at once
    int first_attempt = 0;
    if (const char *e = std::getenv("CGRT_DEVBUILD_BALANCED")) first_attempt = (*e && *e != '0') ? 1 : 0;
    for (int attempt = first_attempt; attempt < 2; attempt++) {
        HIP_TRY(hipMemsetAsync(a.flag, 0, N * sizeof(int), 0));
        if (n > 1) {
            if (attempt == 0) hipLaunchKernelGGL(lbvh_nodes_kernel, grd, blk, 0, 0, a);
            else hipLaunchKernelGGL(balanced_nodes_kernel, grd, blk, 0, 0, a);
        }
        hipLaunchKernelGGL(lbvh_fit_kernel, grd, blk, 0, 0, a);
        // the wide form, level by level: queue[0] = the root
        int root = 0;
        HIP_TRY(hipMemcpy(&root, a.tail + 2, sizeof(int), hipMemcpyDeviceToHost));
        const int2 q0 = make_int2(root, 0);
        const int t0[2] = {1, 0};
        HIP_TRY(hipMemcpy(a.queue, &q0, sizeof(q0), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(a.tail, t0, sizeof(t0), hipMemcpyHostToDevice));
        int begin = 0, end = 1;
        while (begin < end) {
            hipLaunchKernelGGL(wide_level_kernel, dim3((unsigned)((end - begin + 255) / 256)), blk, 0, 0, a, begin, end);
            int t[2];
            HIP_TRY(hipMemcpy(t, a.tail, sizeof(t), hipMemcpyDeviceToHost));
            begin = end;
            end = t[0];
            out.stack_need = t[1];
            if (end > n) return fail(CGRT_ERR_DEVICE, "device hierarchy build: more wide nodes than triangles");
        }
        out.nwide = end;
        out.balanced = attempt == 1;
        if (out.stack_need < kWideStack) break;
        if (attempt == 1) return fail(CGRT_ERR_LIMIT, "device hierarchy build: stack bound missed by the balanced tree");
    }
    // bounding sphere (add_mesh_triangles) and cover spheres (cover_spheres)
    unsigned long long g[7];
    HIP_TRY(hipMemcpy(g, gbox.p, sizeof(g), hipMemcpyDeviceToHost));
    for (int k = 0; k < 3; k++) {
        out.centre[k] = 0.5 * (dunkey(g[k]) + dunkey(g[3 + k]));
        out.bmax = std::max(out.bmax, std::max(std::fabs(dunkey(g[k])), std::fabs(dunkey(g[3 + k]))));
    }
    const double r = std::sqrt(dunkey(g[6])) + 1e-3;
    out.r2 = r * r * (1 + 1e-9);
    int depth = 0;
    while (depth < 6 && (n >> (depth + 1)) >= 64) depth++;
    const int G = 1 << depth;
    HIP_TRY(cover.alloc((size_t)G * 4 * sizeof(double)));
    hipLaunchKernelGGL(cover_kernel, dim3((unsigned)G), dim3(256), 0, 0, a, G, cover.as<double>());
    HIP_TRY(hipGetLastError());
    out.cover.resize((size_t)G * 4);
    HIP_TRY(hipMemcpy(out.cover.data(), cover.p, out.cover.size() * sizeof(double), hipMemcpyDeviceToHost));
    return CGRT_OK;
}

}  // namespace devbuild
#endif
