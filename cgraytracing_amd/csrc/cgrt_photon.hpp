// Photon pass -- included at the end of cgrt_hip.hip.
//
// Reference: render() main.cpp:223-258 + trace(flag=false) main.cpp:101-128,158-165 + Hashtable hash.h:20-70.
// The reference runs eight OpenMP threads that race on the hitpoints and seed rand() from the clock, so its
// output is not reproducible even against itself.  The semantics implemented -- and pinned bit for bit by the
// oracle and by golden vectors from the compiled reference run on ONE thread -- are its serial meaning:
// photons 0..N-1 one after another, photon i drawing from the keyed stream (seed, i, 0, 'phot') in the
// reference's call order, each diffuse hit updating the hitpoints of the 27 surrounding hash cells.
//
// A hitpoint's evolution (r2, n, flux) depends only on the ORDERED list of photon events that reach it, and
// photon paths do not depend on the hitpoints.  So the serial result is computed in parallel as
//   1. photon_trace_kernel   : one lane per photon; every diffuse hit appends an event {P, n, flux} at a slot
//                              derived from (photon, segment), i.e. in serial order;
//   2. photon_pairs_kernel   : one lane per event; walks the reference's own candidate set -- the buckets the 27
//                              cells hash to (hash.h:35-37), collisions included -- and emits (hitpoint, order)
//                              pairs that pass the static tests (normal, distance against the radius the hitpoint
//                              had at the start of the batch, which only shrinks);
//   3. radix sort of the pairs by (hitpoint, order)                                     [rocprim]
//   4. photon_apply_kernel   : one lane per hitpoint; replays its events in order with the reference's update
//                              (main.cpp:116-122), re-checking the distance against the current radius.
// Hitpoints are kept sorted by (bucket, emission order) = the reference's table order, which is also the order of
// its final gather (main.cpp:252-258); the image is summed per pixel in that order.
#include <rocprim/device/device_radix_sort.hpp>

namespace {

constexpr int kSegStride = 8;          // event slots per photon (MAX_DEPTH = 5 segments)
#ifndef CGRT_PHOTON_WAVES
#define CGRT_PHOTON_WAVES 4
#endif
constexpr int kPhotonWaves = CGRT_PHOTON_WAVES;  // waves per SIMD photon_trace_kernel<false> is compiled for
constexpr double kPiRef = 3.14159265358979;  // main.cpp:26

struct PhotonArgs {
    double light[3], jitter, power, alpha;
    long long first;  // index of the first photon of this batch
    int count;        // photons in this batch
    int max_depth;
    uint64_t seed;
};
struct HashArgs {  // hash.h:20-42
    int hashsize;
    double celllength;
};
__device__ __forceinline__ unsigned ref_hash(int ix, int iy, int iz, int hashsize) {
    return (((unsigned)ix * 73856093u) ^ ((unsigned)iy * 19349663u) ^ ((unsigned)iz * 83492791u)) % (unsigned)hashsize;
}
__device__ __forceinline__ void ref_coord(double x, double y, double z, double cl, int &ix, int &iy, int &iz) {
    ix = (int)floor((x - (-35.0)) / cl);
    iy = (int)floor((y - (-35.0)) / cl);
    iz = (int)floor((z - (-15.0)) / cl);
}

// hitpoint records from the eye pass (10 doubles: f pos normal label) -> sort keys (bucket, emission order)
__global__ void hp_keys_kernel(const double *__restrict__ rec, long long n, HashArgs ha, int rows_w, int spp,
                               unsigned long long *__restrict__ keys, unsigned int *__restrict__ vals) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double *q = rec + 10 * i;
    int ix, iy, iz;
    ref_coord(q[3], q[4], q[5], ha.celllength, ix, iy, iz);
    const unsigned long long b = ref_hash(ix, iy, iz, ha.hashsize);
    const unsigned long long lab = (unsigned long long)q[9];
    const unsigned long long seq = lab & 15ull, ps = lab >> 4;
    const unsigned long long smp = ps / (unsigned long long)rows_w, pix = ps % (unsigned long long)rows_w;
    // serial emission order of the reference's eye pass: pixel-major, then sample, then DFS position
    const unsigned long long em = ((pix * (unsigned long long)spp + smp) << 4) | seq;
    keys[i] = (b << 44) | em;  // bucket < 2^20, emission key < 2^44
    vals[i] = (unsigned int)i;
}
// sorted order -> structure of arrays + per-bucket start offsets
__global__ void hp_gather_kernel(const double *__restrict__ rec, const unsigned long long *__restrict__ keys,
                                 const unsigned int *__restrict__ vals, long long n, double r2_init,
                                 double *__restrict__ hp /* n x 16 */, double *__restrict__ hps /* n x 8 */,
                                 int *__restrict__ bucket_of) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double *q = rec + 10 * (long long)vals[i];
    double *o = hp + 16 * i;
    const unsigned long long em = keys[i] & ((1ull << 44) - 1ull);
    o[0] = (double)(em >> 4);  // pixel*spp + sample (decoded by the caller)
    o[1] = (double)(em & 15ull);
    for (int k = 0; k < 9; k++) o[2 + k] = q[k];
    o[11] = 0; o[12] = 0; o[13] = 0;
    o[14] = r2_init;
    o[15] = 0;
    // what the pair search reads, in half a cache line: {pos, batch-start r2} (every candidate), {normal} (those the
    // radius screen lets through)
    double *c = hps + 8 * i;
    for (int k = 0; k < 3; k++) c[k] = q[3 + k];
    c[3] = r2_init;
    for (int k = 0; k < 3; k++) c[4 + k] = q[6 + k];
    c[7] = 0;
    bucket_of[i] = (int)(keys[i] >> 44);
}
__global__ void bucket_start_kernel(const int *__restrict__ bucket_of, long long n, int hashsize, int *__restrict__ bstart) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b > hashsize) return;
    long long lo = 0, hi = n;  // first index with bucket_of >= b
    while (lo < hi) {
        const long long mid = (lo + hi) >> 1;
        if (bucket_of[mid] < b) lo = mid + 1; else hi = mid;
    }
    bstart[b] = (int)lo;
}

// sampling.h:11-29 on the photon's sequential stream
__device__ __forceinline__ V3 sample_sphere(Stream &rs) {
    while (true) {
        const double x = rs.u01() * 2.0 - 1, y = rs.u01() * 2.0 - 1, z = rs.u01() * 2.0 - 1;
        if (x * x + y * y + z * z <= 1) return normalized(mk(x, y, z));
    }
}

// whether photon_trace_kernel<false> keeps the wide walk's first stack entries in LDS: 32 KiB per workgroup, taken only
// while four workgroups (its 4 waves/SIMD) still fit a CU's 160 KiB beside the object list
__host__ __device__ inline bool photon_lds_stack(const DeviceScene &sc) { return sc.has_wide && sc.n_objs <= 56; }
// 1. photon paths.  events: count*kSegStride records of 9 doubles; valid: same count of bytes.
template <bool BEZ, bool SPILL = false>
__global__ __launch_bounds__(kThreads, BEZ ? 2 : kPhotonWaves) void photon_trace_kernel(DeviceScene sc, PhotonArgs pa, double *__restrict__ events,
                                                                   unsigned char *__restrict__ valid) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    ObjRec *lobjs = reinterpret_cast<ObjRec *>(lds_raw);
    {
        const uint4 *src = reinterpret_cast<const uint4 *>(sc.objs);
        uint4 *dst = reinterpret_cast<uint4 *>(lobjs);
        const int n16 = sc.n_lds * (int)(sizeof(ObjRec) / 16);
        for (int k = threadIdx.x; k < n16; k += kThreads) dst[k] = src[k];
    }
    __syncthreads();
    unsigned char *lrest = lds_raw + obj_list_lds(sc, kThreads / 64);
    // BEZ: one BezLds per wave behind the object list; the Newton starts continue the photon's own stream
    LdsAux aux{BEZ ? reinterpret_cast<volatile BezLds *>(lrest) + (threadIdx.x >> 6) : nullptr, nullptr};
    if (SPILL) aux.spill = lobjs + sc.n_lds + (threadIdx.x >> 6);
    // without Bezier objects: the first entries of the 4-wide walk's stack live in LDS behind the object list, as in the eye pass
    if (!BEZ && !SPILL && photon_lds_stack(sc)) aux.wstack = reinterpret_cast<uint2 *>(lrest);  // (the SPILL launch reserves no room for it)
    const int p = blockIdx.x * kThreads + threadIdx.x;
    bool alive = p < pa.count;
    Stream rs(stream_key(pa.seed, (uint64_t)(pa.first + (alive ? p : 0)), 0, 0x70686f74ull));
    V3 o = mk(0, 0, 0), d = mk(0, 0, 1), flux = mk(0, 0, 0);
    if (alive) {  // main.cpp:240-246
        const double a = rs.u01() * (2 * pa.jitter) - pa.jitter;
        const double b = rs.u01() * (2 * pa.jitter) - pa.jitter;
        o = mk(pa.light[0], pa.light[1], pa.light[2]) + mk(a, 0, b);
        d = sample_sphere(rs);
        flux = mk(pa.power, pa.power, pa.power) * (kPiRef * 4.0);
    }
    uint32_t dn = 0, dt = 0;
    for (int seg = 0; seg < pa.max_depth; seg++) {
        if (__ballot(alive) == 0ull) break;
        RayKey rk{rs.key, 1, true, rs.n};
        const SceneHit hit = intersect_scene<true, BEZ, false, false, SPILL>(lobjs, sc.n_lds, sc.n_objs, sc, o, d, rk, alive, aux, dn, dt);
        if (BEZ) rs.n = rk.n0;
        if (!alive) continue;
        if (hit.id < 0) {
            alive = false;
            continue;
        }
        const ObjMat ob = load_mat<SPILL>(lobjs, sc.n_lds, sc.objs, hit.id);
        const V3 P = o + d * hit.t;
        V3 n = hit.n;
        const V3 n_old = n;
        bool into = true;
        if (dot(n, d) > 0) {
            n = -n;
            into = false;
        }
        V3 f = ob.col;
        if (ob.kind == KIND_PLANE && ob.tex >= 0) {
            V3 c;
            if (texture_color(sc.texs[ob.tex], sc.texels, P, c)) f = c;
        }
        const double pmax = (f.x > f.y && f.x > f.z) ? f.x : (f.y > f.z ? f.y : f.z);  // util.h:16-27, main.cpp:79
        const double refl = ob.refl, transp = ob.transp;
        if (refl < kEps && transp < kEps) {
            // diffuse: record the event the serial loop of main.cpp:103-125 would process now
            const size_t slot = (size_t)p * kSegStride + seg;
            double *e = events + 9 * slot;
            e[0] = P.x; e[1] = P.y; e[2] = P.z;
            e[3] = n.x; e[4] = n.y; e[5] = n.z;
            e[6] = flux.x; e[7] = flux.y; e[8] = flux.z;
            valid[slot] = 1;
            V3 nd;
            while (true) {  // uniform_sampling_halfsphere, sampling.h:22-29
                nd = sample_sphere(rs);
                if (dot(nd, n) > 0) break;
            }
            o = P;  // main.cpp:127: no epsilon offset here
            d = nd;
            flux = mulv(f, flux) * (1.0 / pmax);
        } else if (transp < kEps) {
            const V3 nd = d - n * 2.0 * dot(n, d);  // main.cpp:131-134
            o = P + n * kEps;
            d = nd;
            flux = mulv(f, flux) * refl;
        } else {
            const double nc = 1.0, nt = 1.33;  // main.cpp:140-164
            const double nnt = into ? nc / nt : nt / nc;
            const double ddn = dot(d, n);
            const V3 refl_dir = d - n_old * 2.0 * dot(n_old, d);
            const double cos2t = 1 - nnt * nnt * (1 - ddn * ddn);
            if (cos2t < 0) {
                o = P + n * kEps;
                d = refl_dir;
            } else {
                const V3 refr_dir = normalized(d * nnt - n_old * ((into ? 1 : -1) * (ddn * nnt + sqrt(cos2t))));
                if (rs.u01() < 0.5) {  // Russian roulette; the photon's flux is untouched by glass
                    o = P + n * kEps;
                    d = refl_dir;
                } else {
                    o = P - n * kEps;
                    d = refr_dir;
                }
            }
        }
    }
}

// Spatial order for the pair search: events keyed by their hash-grid cell (invalid slots last).  Lanes of a wave then
// probe the same few buckets, so their loads coalesce and hit in L1 instead of being 64 unrelated L2 round trips.
// Only the ORDER OF THE SEARCH changes; every pair still carries its slot number = serial position.
constexpr unsigned kNoEvent = 0xffffffffu;
__global__ void event_keys_kernel(const double *__restrict__ events, const unsigned char *__restrict__ valid, int nslots,
                                  HashArgs ha, unsigned int *__restrict__ keys, unsigned int *__restrict__ vals) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= nslots) return;
    unsigned key = kNoEvent;
    if (valid[s]) {
        const double *e = events + 9 * (size_t)s;
        int ix, iy, iz;
        ref_coord(e[0], e[1], e[2], ha.celllength, ix, iy, iz);
        ix = ix < 0 ? 0 : (ix > 1023 ? 1023 : ix);
        iy = iy < 0 ? 0 : (iy > 1023 ? 1023 : iy);
        iz = iz < 0 ? 0 : (iz > 1022 ? 1022 : iz);
        key = ((unsigned)iz << 20) | ((unsigned)iy << 10) | (unsigned)ix;
    }
    keys[s] = key;
    vals[s] = (unsigned)s;
}

// 2. candidate (hitpoint, event) pairs.  One lane per event walks the reference's candidate set: the buckets its 27
// neighbour cells hash to (hash.h:35-37, main.cpp:107-113) -- cells that collide in the table are deliberately NOT
// deduplicated, the reference walks such a bucket once per cell.
// Hits are staged in a per-wave LDS buffer (the walk is wave-uniform: every lane steps through its bucket together, a
// ballot hands out buffer slots) and leave for global memory in blocks: space for everything a workgroup still holds at
// the end is reserved with ONE atomic per workgroup, a wave whose buffer fills up earlier reserves for itself.  A global
// atomic per hit serialises on a single address in L2 -- measured 16 ns each, 13.5 ms per 1.3 M events -- and dominated
// the whole photon pass; counting first and writing in a second walk (the previous form) paid for every probe twice.
constexpr int kWalk = 4;      // bucket entries a lane tests per step of the pair search
constexpr int kPairBuf = 768;  // staged pairs per wave (6 KiB); flushed before an iteration that could overflow it

// `base` is a position in the 64-bit count of ALL pairs the batch produces; only positions below `cap` exist in memory.  The
// count itself is never clamped: the host compares the 64-bit total with cap and redoes an overflowing batch in halves (a
// 32-bit count would wrap at settings within reach -- 5 M events x 1000 hitpoints inside the initial radius -- and pass).
__device__ __forceinline__ void pairs_flush(const unsigned long long *buf, unsigned cnt, unsigned long long base,
                                            unsigned long long *__restrict__ keys, unsigned int *__restrict__ vals,
                                            unsigned long long cap) {
    const int lane = threadIdx.x & 63;
    for (unsigned k = lane; k < cnt; k += 64) {
        const unsigned long long key = buf[k];
        if (base + k < cap) {
            keys[base + k] = key;
            vals[base + k] = (unsigned int)(key & 0xffffffull);  // the slot number is the key's low 24 bits
        }
    }
}

__global__ __launch_bounds__(256) void photon_pairs_kernel(const double *__restrict__ events,
                                                           const unsigned int *__restrict__ order_keys,
                                                           const unsigned int *__restrict__ order, int nslots, HashArgs ha,
                                                           const double *__restrict__ hps, const int *__restrict__ bstart,
                                                           unsigned long long *__restrict__ keys,
                                                           unsigned int *__restrict__ vals,
                                                           unsigned long long *__restrict__ npairs /* [0] pairs, [1] events */,
                                                           unsigned long long cap) {
    __shared__ unsigned long long stage[4][kPairBuf];
    __shared__ unsigned wave_cnt[4], wave_ev[4];
    __shared__ unsigned long long block_base;
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool on = t < nslots && order_keys[t] != kNoEvent;
    const int s = on ? (int)order[t] : 0;
    unsigned long long *buf = stage[wave];
    unsigned cnt = 0;  // wave-uniform: pairs staged in buf
    V3 P = mk(0, 0, 0), n = mk(0, 0, 0);
    int ix = 0, iy = 0, iz = 0;
    if (on) {
        const double *e = events + 9 * (size_t)s;
        P = mk(e[0], e[1], e[2]);
        n = mk(e[3], e[4], e[5]);
        ref_coord(P.x, P.y, P.z, ha.celllength, ix, iy, iz);
        ix -= 1; iy -= 1; iz -= 1;
    }
    const unsigned long long lt = (1ull << lane) - 1ull;
    if (__ballot(on) != 0ull) {
        int i = 0, i1 = 0, ni = 0, ni1 = 0;
        if (on) {
            const unsigned b = ref_hash(ix, iy, iz, ha.hashsize);
            ni = bstart[b];
            ni1 = bstart[b + 1];
        }
        for (int c = 0; c < 27; c++) {
            i = ni; i1 = ni1;
            ni = ni1 = 0;
            if (on && c + 1 < 27) {  // the next cell's bucket bounds are fetched under this cell's walk
                const int c1 = c + 1;
                const unsigned b = ref_hash(ix + c1 / 9, iy + (c1 / 3) % 3, iz + c1 % 3, ha.hashsize);
                ni = bstart[b];
                ni1 = bstart[b + 1];
            }
            while (__ballot(i < i1) != 0ull) {  // all lanes step through their buckets together
                if (cnt > (unsigned)(kPairBuf - 64 * kWalk)) {  // the next step could overflow: this wave reserves for itself
                    unsigned long long base = 0;
                    if (lane == 0) base = atomicAdd(npairs, (unsigned long long)cnt);
                    base = __shfl(base, 0);
                    pairs_flush(buf, cnt, base, keys, vals, cap);
                    cnt = 0;
                }
                // kWalk bucket entries per step (their loads in flight together); pair order in the buffer is free, the
                // pairs are sorted by (hitpoint, slot) afterwards
                bool hit[kWalk];
                V3 dd[kWalk];
                double r2s[kWalk];
                bool cand[kWalk];
                const double2 *h = reinterpret_cast<const double2 *>(hps) + 4 * (size_t)i;  // 64-byte records
#pragma unroll
                for (int u = 0; u < kWalk; u++) {
                    hit[u] = false;
                    cand[u] = i + u < i1;
                    const double2 *g = cand[u] ? h + 4 * u : reinterpret_cast<const double2 *>(hps);
                    const double2 g0 = g[0], g1 = g[1];  // {x, y} {z, r2}
                    dd[u] = mk(g0.x, g0.y, g1.x) - P;    // the reference's differences
                    r2s[u] = g1.y;
                }
                // Single-precision screen of the radius test (fewer than 1 in 300 candidates pass it): the fp64
                // differences rounded to fp32, their squares summed in fp32 -- all terms >= 0, so the result is within
                // 2^-21 relative of the fp64 sum (plus at most 3 * 2^-150 where a square is subnormal); an overflow means
                // a distance no radius reaches, a NaN passes the screen.  The bound is r2 * (1 + 2^-18) converted to
                // nearest (>= r2 * (1 + 2^-19)) plus 1e-37, so nothing the exact test accepts is screened out; survivors
                // take the exact test.
#pragma unroll
                for (int u = 0; u < kWalk; u++) {
                    const float ax = (float)dd[u].x, ay = (float)dd[u].y, az = (float)dd[u].z;
                    const float sq = ax * ax + ay * ay + az * az;
                    const float lim = (float)(r2s[u] * (1.0 + 0x1p-18)) + 1e-37f;
                    if (cand[u] && !(sq > lim)) {
                        const double2 g2 = h[4 * u + 2], g3 = h[4 * u + 3];  // {nx, ny} {nz, -}
                        hit[u] = (dot(mk(g2.x, g2.y, g3.x), n) > kEps) && (dot(dd[u], dd[u]) <= r2s[u]);  // main.cpp:116, batch-start r2
                    }
                }
#pragma unroll
                for (int u = 0; u < kWalk; u++) {
                    const unsigned long long m = __ballot(hit[u]);
                    if (hit[u]) buf[cnt + (unsigned)__popcll(m & lt)] = ((unsigned long long)(i + u) << 24) | (unsigned long long)s;  // s < 2^24
                    cnt += (unsigned)__popcll(m);
                }
                i += kWalk;
            }
        }
    }
    const unsigned nev_wave = (unsigned)__popcll(__ballot(on));
    if (lane == 0) {
        wave_cnt[wave] = cnt;
        wave_ev[wave] = nev_wave;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned tot = wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
        const unsigned nev = wave_ev[0] + wave_ev[1] + wave_ev[2] + wave_ev[3];
        block_base = tot ? atomicAdd(npairs, (unsigned long long)tot) : 0ull;
        if (nev) atomicAdd(npairs + 1, (unsigned long long)nev);  // events processed (statistics)
    }
    __syncthreads();
    unsigned long long base = block_base;
    for (int w = 0; w < wave; w++) base += wave_cnt[w];
    pairs_flush(buf, cnt, base, keys, vals, cap);
}

// 4. ordered replay per hitpoint
__global__ void photon_apply_kernel(const unsigned long long *__restrict__ keys, const unsigned int *__restrict__ vals,
                                    unsigned int npairs, const double *__restrict__ events, double alpha,
                                    double *__restrict__ hp, double *__restrict__ hps, long long nhp) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nhp) return;
    const unsigned long long klo = (unsigned long long)i << 24;
    unsigned int lo = 0, hi = npairs;  // first pair of hitpoint i
    while (lo < hi) {
        const unsigned int mid = (lo + hi) >> 1;
        if (keys[mid] < klo) lo = mid + 1; else hi = mid;
    }
    if (lo >= npairs || (keys[lo] >> 24) != (unsigned long long)i) return;
    double *h = hp + 16 * i;
    const V3 f = mk(h[2], h[3], h[4]), pos = mk(h[5], h[6], h[7]);
    V3 flux = mk(h[11], h[12], h[13]);
    double r2 = h[14];
    int n = (int)h[15];
    // The replay is a serial chain per hitpoint and the kernel ends with its longest chains, so the loads of kChunk pairs
    // (key, slot, event: three dependent levels) are issued together and only the update itself runs in sequence.
    constexpr int kChunk = 4;
    for (unsigned int k = lo; k < npairs;) {
        bool mine[kChunk];
        unsigned int slot[kChunk];
#pragma unroll
        for (int c = 0; c < kChunk; c++) {
            const unsigned int kc = k + (unsigned)c < npairs ? k + (unsigned)c : npairs - 1u;
            mine[c] = k + (unsigned)c < npairs && (keys[kc] >> 24) == (unsigned long long)i;
            slot[c] = vals[kc];
        }
        double e[kChunk][6];
#pragma unroll
        for (int c = 0; c < kChunk; c++) {
            const double *q = events + 9 * (size_t)slot[c];
            e[c][0] = q[0]; e[c][1] = q[1]; e[c][2] = q[2];
            e[c][3] = q[6]; e[c][4] = q[7]; e[c][5] = q[8];
        }
        bool more = true;
#pragma unroll
        for (int c = 0; c < kChunk; c++) {
            more = more && mine[c];  // the hitpoint's pairs are contiguous: the first foreign key ends the replay
            if (more) {
                const V3 dd = pos - mk(e[c][0], e[c][1], e[c][2]);
                if (dot(dd, dd) <= r2) {  // main.cpp:116 against the CURRENT radius (the normal test was static)
                    const double g = (n * alpha + alpha) / (n * alpha + 1.0);  // main.cpp:119
                    r2 *= g;
                    n++;
                    flux = (flux + mulv(f, mk(e[c][3], e[c][4], e[c][5])) * (1.0 / kPiRef)) * g;  // main.cpp:122
                }
            }
        }
        if (!more) break;
        k += kChunk;
    }
    h[11] = flux.x; h[12] = flux.y; h[13] = flux.z;
    h[14] = r2;
    h[15] = (double)n;
    hps[8 * i + 3] = r2;  // the next batch's search radius
}

// final gather, main.cpp:252-258: per pixel, in table order
__global__ void photon_image_kernel(const unsigned long long *__restrict__ keys, const unsigned int *__restrict__ vals,
                                    long long nhp, const double *__restrict__ hp, double norm, long long npix,
                                    double *__restrict__ image) {
    const long long px = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (px >= npix) return;
    const unsigned long long klo = (unsigned long long)px << 32;
    long long lo = 0, hi = nhp;
    while (lo < hi) {
        const long long mid = (lo + hi) >> 1;
        if (keys[mid] < klo) lo = mid + 1; else hi = mid;
    }
    double r = 0, g = 0, b = 0;
    for (long long k = lo; k < nhp && (keys[k] >> 32) == (unsigned long long)px; k++) {
        const double *h = hp + 16 * (size_t)vals[k];
        const double sc = 1.0 / (kPiRef * h[14] * norm);  // 1/(PI*r2*N*spp), main.cpp:256
        r += h[11] * sc;
        g += h[12] * sc;
        b += h[13] * sc;
    }
    image[3 * px] = r; image[3 * px + 1] = g; image[3 * px + 2] = b;
}
__global__ void image_keys_kernel(const double *__restrict__ hp, long long nhp, int spp, unsigned long long *__restrict__ keys,
                                  unsigned int *__restrict__ vals) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nhp) return;
    const unsigned long long ps = (unsigned long long)hp[16 * i];
    keys[i] = ((ps / (unsigned long long)spp) << 32) | (unsigned long long)i;  // pixel, then table order
    vals[i] = (unsigned int)i;
}

// Grow-only scratch for the radix sorts: hipMalloc / hipFree per sort would drain the device (hipFree synchronises)
// twice per photon batch.
struct SortTemp {
    DevBuf buf;
    size_t cap = 0;
    hipError_t need(size_t bytes) {
        if (bytes <= cap) return hipSuccess;
        if (buf.p) {
            (void)hipFree(buf.p);
            buf.p = nullptr;
            cap = 0;
        }
        const hipError_t e = buf.alloc(bytes);
        if (e == hipSuccess) cap = bytes;
        return e;
    }
};

int sort_pairs(SortTemp &tmp, unsigned long long *kin, unsigned long long *kout, unsigned int *vin, unsigned int *vout, size_t n,
               int end_bit = 64, hipStream_t st = 0) {
    size_t tmp_bytes = 0;
    HIP_TRY(rocprim::radix_sort_pairs(nullptr, tmp_bytes, kin, kout, vin, vout, n, 0u, (unsigned)end_bit, st));
    HIP_TRY(tmp.need(tmp_bytes));
    HIP_TRY(rocprim::radix_sort_pairs(tmp.buf.p, tmp_bytes, kin, kout, vin, vout, n, 0u, (unsigned)end_bit, st));
    return CGRT_OK;
}

void launch_photon_trace(const cgrt_scene *s, const PhotonArgs &pa, double *events, unsigned char *valid, hipStream_t st = 0) {
    const dim3 grid((pa.count + kThreads - 1) / kThreads), block(kThreads);
    const size_t lds = obj_list_lds(s->dev, kThreads / 64);
    if (s->dev.n_objs > s->dev.n_lds) {  // more objects than the LDS list holds: the variants that read the rest from the uploaded array
        const size_t l2 = lds + (s->dev.has_bezier ? (kThreads / 64) * sizeof(BezLds) : 0);
        if (s->dev.has_bezier) {
            BIG_LDS((photon_trace_kernel<true, true>), l2);
            hipLaunchKernelGGL((photon_trace_kernel<true, true>), grid, block, l2, st, s->dev, pa, events, valid);
        } else {
            BIG_LDS((photon_trace_kernel<false, true>), l2);
            hipLaunchKernelGGL((photon_trace_kernel<false, true>), grid, block, l2, st, s->dev, pa, events, valid);
        }
        return;
    }
    if (s->dev.has_bezier)
        hipLaunchKernelGGL(photon_trace_kernel<true>, grid, block, lds + (kThreads / 64) * sizeof(BezLds), st, s->dev, pa, events,
                           valid);
    else
        hipLaunchKernelGGL(photon_trace_kernel<false>, grid, block,
                           lds + (photon_lds_stack(s->dev) ? (size_t)kThreads * kWideLdsDepth * sizeof(uint2) : 0), st, s->dev, pa, events, valid);
}

int sort_pairs32(SortTemp &tmp, unsigned int *kin, unsigned int *kout, unsigned int *vin, unsigned int *vout, size_t n,
                 hipStream_t st = 0) {
    size_t tmp_bytes = 0;
    HIP_TRY(rocprim::radix_sort_pairs(nullptr, tmp_bytes, kin, kout, vin, vout, n, 0u, 32u, st));
    HIP_TRY(tmp.need(tmp_bytes));
    HIP_TRY(rocprim::radix_sort_pairs(tmp.buf.p, tmp_bytes, kin, kout, vin, vout, n, 0u, 32u, st));
    return CGRT_OK;
}

// The producer side of a photon batch (trace -> event keys -> events in hash-cell order) on its own stream, so that the
// batch after the one being replayed is traced meanwhile: photon paths do not depend on hitpoints.
struct PhotonProducer {
    hipStream_t st = nullptr;
    hipEvent_t produced[2] = {nullptr, nullptr}, consumed[2] = {nullptr, nullptr};
    bool used[2] = {false, false};
    DevBuf ev[2], valid[2], ek0[2], ek1[2], eo0[2], eo1[2];
    SortTemp tmp;
    ~PhotonProducer() {
        if (st) {
            (void)hipStreamSynchronize(st);
            (void)hipStreamDestroy(st);
        }
        for (int k = 0; k < 2; k++) {
            if (produced[k]) (void)hipEventDestroy(produced[k]);
            if (consumed[k]) (void)hipEventDestroy(consumed[k]);
        }
    }
    int init(int nbuf, size_t nslots_max, bool own_stream) {
        if (own_stream) HIP_TRY(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        for (int k = 0; k < nbuf; k++) {
            HIP_TRY(hipEventCreateWithFlags(&produced[k], hipEventDisableTiming));
            HIP_TRY(hipEventCreateWithFlags(&consumed[k], hipEventDisableTiming));
            HIP_TRY(ev[k].alloc(nslots_max * 9 * sizeof(double)));
            HIP_TRY(valid[k].alloc(nslots_max));
            HIP_TRY(ek0[k].alloc(nslots_max * 4)); HIP_TRY(ek1[k].alloc(nslots_max * 4));
            HIP_TRY(eo0[k].alloc(nslots_max * 4)); HIP_TRY(eo1[k].alloc(nslots_max * 4));
        }
        return CGRT_OK;
    }
    // enqueue batch `pa` into buffer b (after the replay that last read b has finished)
    int produce(const cgrt_scene *s, const PhotonArgs &pa, const HashArgs &ha, int b);
    // null stream: the replay of buffer b is enqueued; b may be overwritten once it has run
    int release(int b) {
        HIP_TRY(hipEventRecord(consumed[b], 0));
        used[b] = true;
        return CGRT_OK;
    }
};

int PhotonProducer::produce(const cgrt_scene *s, const PhotonArgs &pa, const HashArgs &ha, int b) {
    const int T = 256;
    const int nslots = pa.count * kSegStride;
    if (used[b]) HIP_TRY(hipStreamWaitEvent(st, consumed[b], 0));
    HIP_TRY(hipMemsetAsync(valid[b].p, 0, (size_t)nslots, st));
    launch_photon_trace(s, pa, ev[b].as<double>(), valid[b].as<unsigned char>(), st);
    hipLaunchKernelGGL(event_keys_kernel, dim3((nslots + T - 1) / T), dim3(T), 0, st, ev[b].as<double>(), valid[b].as<unsigned char>(),
                       nslots, ha, ek0[b].as<unsigned int>(), eo0[b].as<unsigned int>());
    const int rc = sort_pairs32(tmp, ek0[b].as<unsigned int>(), ek1[b].as<unsigned int>(), eo0[b].as<unsigned int>(),
                                eo1[b].as<unsigned int>(), (size_t)nslots, st);
    if (rc) return rc;
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(produced[b], st));
    return CGRT_OK;
}

}  // namespace

// verification probe: the diffuse hits of photons [first, first+count) -- count*8 slots of 9 doubles {P, n, flux}
// (slot = (photon-first)*8 + path segment) and one validity byte per slot, HOST buffers
extern "C" int cgrt_photon_events(const cgrt_scene *s, const cgrt_photons *ph, int max_depth, int64_t first, int32_t count,
                                  double *events9, uint8_t *valid) {
    if (!s || !s->committed || !ph || !events9 || !valid || count <= 0 || count > (1 << 20) || max_depth < 1 ||
        max_depth > kMaxDepth)
        return fail(CGRT_ERR_INVALID, "bad argument");
    ON_DEVICE(s->device);
    DevBuf ev, va;
    const size_t nslots = (size_t)count * kSegStride;
    HIP_TRY(ev.alloc(nslots * 9 * sizeof(double)));
    HIP_TRY(va.alloc(nslots));
    HIP_TRY(hipMemset(ev.p, 0, nslots * 9 * sizeof(double)));
    HIP_TRY(hipMemset(va.p, 0, nslots));
    PhotonArgs pa;
    for (int k = 0; k < 3; k++) pa.light[k] = ph->light[k];
    pa.jitter = ph->jitter; pa.power = ph->power; pa.alpha = ph->alpha;
    pa.first = first; pa.count = count; pa.max_depth = max_depth; pa.seed = ph->seed;
    launch_photon_trace(s, pa, ev.as<double>(), va.as<unsigned char>());
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(events9, ev.p, nslots * 9 * sizeof(double), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(valid, va.p, nslots, hipMemcpyDeviceToHost));
    return CGRT_OK;
}

// gammaCorr (util.h:45-47) and the vertical flip of main.cpp:403-412
__global__ void tonemap_kernel(const double *__restrict__ image, int W, int H, unsigned char *__restrict__ rgb8) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;  // output byte index, top row first
    const long long n = (long long)W * H * 3;
    if (i >= n) return;
    const long long row = i / ((long long)W * 3), rest = i % ((long long)W * 3);
    const double x = image[(long long)(H - 1 - row) * W * 3 + rest];
    const double v = pow(1 - exp(-x), 1 / 2.2) * 255 + .5;
    rgb8[i] = (v >= 0) ? (unsigned char)(int)(v < 255.0 ? v : 255.0) : 0;  // NaN -> 0
}

namespace {
struct Timer {  // device time between two points of the null stream
    hipEvent_t a = nullptr, b = nullptr;
    Timer() { (void)hipEventCreate(&a); (void)hipEventCreate(&b); }
    ~Timer() { if (a) (void)hipEventDestroy(a); if (b) (void)hipEventDestroy(b); }
    void start() { (void)hipEventRecord(a, 0); }
    double stop() {
        (void)hipEventRecord(b, 0);
        (void)hipEventSynchronize(b);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, a, b);
        return (double)ms;
    }
};
}  // namespace

extern "C" int cgrt_tonemap_rgb8(int device, const double *image, int width, int height, uint8_t *rgb8) {
    if (!image || !rgb8 || width < 1 || height < 1) return fail(CGRT_ERR_INVALID, "bad argument");
    ON_DEVICE(device);
    const size_t n = (size_t)width * height * 3;
    DevBuf img, out;
    HIP_TRY(img.alloc(n * sizeof(double)));
    HIP_TRY(out.alloc(n));
    HIP_TRY(hipMemcpy(img.p, image, n * sizeof(double), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(tonemap_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, img.as<double>(), width, height,
                       out.as<unsigned char>());
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(rgb8, out.p, n, hipMemcpyDeviceToHost));
    return CGRT_OK;
}

extern "C" int cgrt_ppm_render(const cgrt_scene *s, const cgrt_camera *cam, const cgrt_grid *grid,
                               const cgrt_photons *ph, cgrt_ppm_result *out) {
    int rc = check_grid(s, cam, grid);
    if (rc) return rc;
    if (!ph || !out) return fail(CGRT_ERR_INVALID, "null argument");
    if (ph->nphotons < 0 || ph->hashsize < 1 || ph->hashsize > (1 << 20) || ph->batch < 0 || !(ph->initial_radius >= 0) ||
        ph->pair_cap < 0)
        return fail(CGRT_ERR_INVALID, "bad photon parameters");
    if (grid->stripe_nranks > 1 && out->rgb8)
        return fail(CGRT_ERR_UNSUPPORTED, "photon pass: rgb8 needs contiguous rows; tone-map the assembled frame (cgrt_tonemap_rgb8)");
    ON_DEVICE(s->device);
    Timer tm;
    // ---- eye pass: hitpoint records, device resident (count first, then capture) ----
    tm.start();
    uint64_t nhp = 0;
    rc = hitpoints_device(s, cam, grid, 0, nullptr, &nhp);
    if (rc) return rc;
    const long long npix = (long long)grid->rows * grid->width;
    const size_t n = (size_t)nhp;
    if (n >= (1ull << 31)) return fail(CGRT_ERR_LIMIT, "photon pass: more than 2^31 hitpoints");
    DevBuf rec, hp, hps, bucket_of, bstart, k0, k1, v0, v1, img;
    SortTemp main_tmp;
    if (n) {
        double *d_rec = nullptr;
        rc = hitpoints_device(s, cam, grid, nhp, &d_rec, &nhp);
        rec.p = d_rec;
        if (rc) return rc;
    }
    out->ms_eye = tm.stop();
    // ---- the reference's table order: (bucket, insertion order) ----
    tm.start();
    HIP_TRY(hp.alloc(n * 16 * sizeof(double)));
    HIP_TRY(hps.alloc(n * 8 * sizeof(double)));
    HIP_TRY(bucket_of.alloc(n * sizeof(int)));
    HIP_TRY(bstart.alloc(((size_t)ph->hashsize + 2) * sizeof(int)));
    HIP_TRY(k0.alloc(n * 8)); HIP_TRY(k1.alloc(n * 8)); HIP_TRY(v0.alloc(n * 4)); HIP_TRY(v1.alloc(n * 4));
    HashArgs ha;
    ha.hashsize = ph->hashsize;
    // main.cpp:84,183: r = 200.0 / height with the reference's COMPILE-TIME height (768) whatever frame is rendered;
    // a host that mirrors a reference built for another height passes that build's 200/height here
    const double r0 = ph->initial_radius > 0 ? ph->initial_radius : 200.0 / 768;
    ha.celllength = 70.0 / std::ceil(70.0 / r0);  // hash.h:25-26
    const int T = 256;
    const unsigned nb = (unsigned)((n + T - 1) / T);
    if (n) {
        hipLaunchKernelGGL(hp_keys_kernel, dim3(nb), dim3(T), 0, 0, rec.as<double>(), (long long)n, ha, (int)npix, grid->spp,
                           k0.as<unsigned long long>(), v0.as<unsigned int>());
        rc = sort_pairs(main_tmp, k0.as<unsigned long long>(), k1.as<unsigned long long>(), v0.as<unsigned int>(), v1.as<unsigned int>(), n);
        if (rc) return rc;
        hipLaunchKernelGGL(hp_gather_kernel, dim3(nb), dim3(T), 0, 0, rec.as<double>(), k1.as<unsigned long long>(),
                           v1.as<unsigned int>(), (long long)n, r0 * r0, hp.as<double>(), hps.as<double>(), bucket_of.as<int>());
    }
    hipLaunchKernelGGL(bucket_start_kernel, dim3((ph->hashsize + 1 + T - 1) / T), dim3(T), 0, 0, bucket_of.as<int>(),
                       (long long)n, ph->hashsize, bstart.as<int>());
    HIP_TRY(hipGetLastError());
    out->ms_table = tm.stop();
    // ---- photons, in batches ----
    tm.start();
    int batch = ph->batch > 0 ? (ph->batch < (1 << 20) ? ph->batch : (1 << 20)) : (1 << 20);
    const int batch_max = batch;
    const int nslots_max = batch * kSegStride;
    DevBuf pk0, pk1, pv0, pv1, npairs;
    // pairs per batch: room for 128 per hitpoint, between 4 M and 128 M (3 GiB of keys and values); a batch that overflows is halved
    const unsigned long long want_cap = (unsigned long long)n * 128ull;
    unsigned long long pair_cap = want_cap < (1ull << 22) ? (1ull << 22) : (want_cap > (1ull << 27) ? (1ull << 27) : want_cap);
    if (ph->pair_cap > 0) pair_cap = (unsigned long long)ph->pair_cap < (1ull << 27) ? (unsigned long long)ph->pair_cap : (1ull << 27);
    // Two event buffers: while batch k's pairs are sorted and replayed (null stream), batch k+1 is traced and its events are
    // put in hash-cell order on the producer's stream.  CGRT_PHOTON_OVERLAP=0: one buffer, everything on the null stream.
    const char *ov = std::getenv("CGRT_PHOTON_OVERLAP");
    const bool overlap = !(ov && ov[0] == '0') && ph->nphotons > batch;
    PhotonProducer pp;
    if (n > 0 && ph->nphotons > 0) {
        rc = pp.init(overlap ? 2 : 1, (size_t)nslots_max, overlap);
        if (rc) return rc;
    }
    HIP_TRY(pk0.alloc((size_t)pair_cap * 8)); HIP_TRY(pk1.alloc((size_t)pair_cap * 8));
    HIP_TRY(pv0.alloc((size_t)pair_cap * 4)); HIP_TRY(pv1.alloc((size_t)pair_cap * 4));
    HIP_TRY(npairs.alloc(16));
    out->n_events = 0;
    out->n_pairs = 0;
    out->n_batch_halvings = 0;
    int pair_key_bits = 25;  // key = hitpoint << 24 | slot
    while (pair_key_bits < 64 && (n >> (pair_key_bits - 24)) != 0) pair_key_bits++;
    auto batch_args = [&](long long first, int batch_now) {
        PhotonArgs pa;
        for (int k = 0; k < 3; k++) pa.light[k] = ph->light[k];
        pa.jitter = ph->jitter; pa.power = ph->power; pa.alpha = ph->alpha;
        pa.first = first;
        pa.count = (int)((ph->nphotons - first < batch_now) ? (ph->nphotons - first) : batch_now);
        pa.max_depth = grid->max_depth;
        pa.seed = ph->seed;
        return pa;
    };
    long long ahead_first = -1;  // the batch already enqueued on the producer: its range and buffer
    int ahead_count = 0, ahead_buf = 0, cur = 0;
    for (long long first = 0; first < ph->nphotons && n > 0;) {
        const PhotonArgs pa = batch_args(first, batch);
        const int nslots = pa.count * kSegStride;
        if (ahead_first == pa.first && ahead_count == pa.count) {
            cur = ahead_buf;
        } else {  // first batch, or the plan changed (a halving): produce it now
            rc = pp.produce(s, pa, ha, cur);
            if (rc) return rc;
        }
        ahead_first = -1;
        if (overlap && first + pa.count < ph->nphotons) {
            // Enqueue the NEXT batch now, so that it is traced under this batch's search, sort and replay.  Its range assumes
            // this batch neither overflows the pair buffer nor changes the batch size; if it does, the range will not match
            // at the top of the loop and the batch is produced again (results do not depend on the batching).
            const PhotonArgs nx = batch_args(first + pa.count, batch);
            rc = pp.produce(s, nx, ha, 1 - cur);
            if (rc) return rc;
            ahead_first = nx.first; ahead_count = nx.count; ahead_buf = 1 - cur;
        }
        if (pp.st) HIP_TRY(hipStreamWaitEvent(0, pp.produced[cur], 0));
        HIP_TRY(hipMemsetAsync(npairs.p, 0, 16, 0));
        hipLaunchKernelGGL(photon_pairs_kernel, dim3((nslots + T - 1) / T), dim3(T), 0, 0, pp.ev[cur].as<double>(),
                           pp.ek1[cur].as<unsigned int>(), pp.eo1[cur].as<unsigned int>(), nslots, ha, hps.as<double>(), bstart.as<int>(),
                           pk0.as<unsigned long long>(), pv0.as<unsigned int>(), npairs.as<unsigned long long>(), pair_cap);
        HIP_TRY(hipGetLastError());
        unsigned long long np2[2] = {0, 0};  // pairs (the full 64-bit count, stored or not), events
        HIP_TRY(hipMemcpy(np2, npairs.p, 16, hipMemcpyDeviceToHost));
        if (np2[0] > pair_cap) {  // nothing has been applied yet: redo this range in smaller batches (same result)
            if (pa.count <= 1) return fail(CGRT_ERR_LIMIT, "photon pass: one photon's pairs exceed the pair buffer");
            batch = (pa.count < batch ? pa.count : batch) / 2;
            out->n_batch_halvings++;
            rc = pp.release(cur);
            if (rc) return rc;
            continue;
        }
        const unsigned int np = (unsigned int)np2[0];  // <= pair_cap <= 2^27
        first += pa.count;
        out->n_events += np2[1];
        if (batch < batch_max && np < pair_cap / 4) batch *= 2;  // radii shrink as photons arrive: later batches hold fewer pairs
        if (np != 0) {
            out->n_pairs += np;
            rc = sort_pairs(main_tmp, pk0.as<unsigned long long>(), pk1.as<unsigned long long>(), pv0.as<unsigned int>(),
                            pv1.as<unsigned int>(), np, pair_key_bits);
            if (rc) return rc;
            hipLaunchKernelGGL(photon_apply_kernel, dim3(nb), dim3(T), 0, 0, pk1.as<unsigned long long>(), pv1.as<unsigned int>(), np,
                               pp.ev[cur].as<double>(), ph->alpha, hp.as<double>(), hps.as<double>(), (long long)n);
            HIP_TRY(hipGetLastError());
        }
        rc = pp.release(cur);
        if (rc) return rc;
    }
    if (pp.st) HIP_TRY(hipStreamSynchronize(pp.st));
    out->ms_photons = tm.stop();
    // ---- final gather + tone map ----
    tm.start();
    HIP_TRY(img.alloc((size_t)npix * 3 * sizeof(double)));
    HIP_TRY(hipMemset(img.p, 0, (size_t)npix * 3 * sizeof(double)));
    if (n) {
        hipLaunchKernelGGL(image_keys_kernel, dim3(nb), dim3(T), 0, 0, hp.as<double>(), (long long)n, grid->spp,
                           k0.as<unsigned long long>(), v0.as<unsigned int>());
        rc = sort_pairs(main_tmp, k0.as<unsigned long long>(), k1.as<unsigned long long>(), v0.as<unsigned int>(), v1.as<unsigned int>(), n);
        if (rc) return rc;
        hipLaunchKernelGGL(photon_image_kernel, dim3((unsigned)((npix + T - 1) / T)), dim3(T), 0, 0, k1.as<unsigned long long>(),
                           v1.as<unsigned int>(), (long long)n, hp.as<double>(), (double)ph->nphotons * grid->spp, npix,
                           img.as<double>());
    }
    DevBuf rgb8;
    if (out->rgb8) {
        HIP_TRY(rgb8.alloc((size_t)npix * 3));
        hipLaunchKernelGGL(tonemap_kernel, dim3((unsigned)((npix * 3 + T - 1) / T)), dim3(T), 0, 0, img.as<double>(), grid->width,
                           grid->rows, rgb8.as<unsigned char>());
    }
    HIP_TRY(hipGetLastError());
    out->ms_gather = tm.stop();
    if (out->image) HIP_TRY(hipMemcpy(out->image, img.p, (size_t)npix * 3 * sizeof(double), hipMemcpyDeviceToHost));
    if (out->rgb8) HIP_TRY(hipMemcpy(out->rgb8, rgb8.p, (size_t)npix * 3, hipMemcpyDeviceToHost));
    out->hp_count = nhp;
    if (out->hp16 && out->hp_cap) {
        const size_t m = n < out->hp_cap ? n : (size_t)out->hp_cap;
        HIP_TRY(hipMemcpy(out->hp16, hp.p, m * 16 * sizeof(double), hipMemcpyDeviceToHost));
    }
    return CGRT_OK;
}

// ---- PNG (host): signature, IHDR, one IDAT of stored deflate blocks, IEND -------------------------------------------
namespace {
struct Crc32 {
    uint32_t table[256];
    Crc32() {
        for (uint32_t i = 0; i < 256; i++) {
            uint32_t c = i;
            for (int k = 0; k < 8; k++) c = (c & 1u) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
            table[i] = c;
        }
    }
    uint32_t run(uint32_t crc, const unsigned char *p, size_t n) const {
        for (size_t i = 0; i < n; i++) crc = table[(crc ^ p[i]) & 0xffu] ^ (crc >> 8);
        return crc;
    }
};
void put32(std::vector<unsigned char> &v, uint32_t x) {
    v.push_back((unsigned char)(x >> 24)); v.push_back((unsigned char)(x >> 16));
    v.push_back((unsigned char)(x >> 8)); v.push_back((unsigned char)x);
}
bool write_chunk(FILE *f, const Crc32 &crc, const char type[4], const std::vector<unsigned char> &data) {
    std::vector<unsigned char> head;
    put32(head, (uint32_t)data.size());
    head.insert(head.end(), type, type + 4);
    uint32_t c = crc.run(0xffffffffu, head.data() + 4, 4);
    c = crc.run(c, data.data(), data.size()) ^ 0xffffffffu;
    std::vector<unsigned char> tail;
    put32(tail, c);
    return std::fwrite(head.data(), 1, head.size(), f) == head.size() &&
           (data.empty() || std::fwrite(data.data(), 1, data.size(), f) == data.size()) &&
           std::fwrite(tail.data(), 1, 4, f) == 4;
}
}  // namespace

extern "C" int cgrt_write_png(const char *path, int width, int height, const uint8_t *rgb8) {
    if (!path || !rgb8 || width < 1 || height < 1 || (uint64_t)width * height > (1ull << 28))
        return fail(CGRT_ERR_INVALID, "bad argument");
    // raw scanlines: filter byte 0 + width*3 bytes
    const size_t stride = (size_t)width * 3, raw_n = (stride + 1) * height;
    std::vector<unsigned char> raw(raw_n);
    for (int y = 0; y < height; y++) {
        raw[(stride + 1) * y] = 0;
        std::memcpy(&raw[(stride + 1) * y + 1], rgb8 + stride * y, stride);
    }
    std::vector<unsigned char> z;
    z.reserve(raw_n + raw_n / 65535 * 5 + 16);
    z.push_back(0x78); z.push_back(0x01);  // zlib header: deflate, 32 K window, no preset dictionary
    uint32_t a = 1, b = 0;                 // adler32
    for (size_t off = 0; off < raw_n; off += 65535) {
        const size_t len = raw_n - off < 65535 ? raw_n - off : 65535;
        z.push_back(off + len == raw_n ? 1 : 0);  // BFINAL, BTYPE = 00 (stored)
        z.push_back((unsigned char)(len & 0xff)); z.push_back((unsigned char)(len >> 8));
        z.push_back((unsigned char)(~len & 0xff)); z.push_back((unsigned char)((~len >> 8) & 0xff));
        z.insert(z.end(), raw.begin() + off, raw.begin() + off + len);
        for (size_t i = 0; i < len; i += 5552) {  // adler32 with deferred modulo
            const size_t m = len - i < 5552 ? len - i : 5552;
            for (size_t k = 0; k < m; k++) { a += raw[off + i + k]; b += a; }
            a %= 65521u; b %= 65521u;
        }
    }
    put32(z, (b << 16) | a);
    FILE *f = std::fopen(path, "wb");
    if (!f) return fail(CGRT_ERR_IO, std::string("cannot open ") + path);
    static const Crc32 crc;
    static const unsigned char sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    std::vector<unsigned char> ihdr;
    put32(ihdr, (uint32_t)width); put32(ihdr, (uint32_t)height);
    ihdr.push_back(8); ihdr.push_back(2); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0);  // 8-bit RGB
    const bool ok = std::fwrite(sig, 1, 8, f) == 8 && write_chunk(f, crc, "IHDR", ihdr) && write_chunk(f, crc, "IDAT", z) &&
                    write_chunk(f, crc, "IEND", std::vector<unsigned char>());
    if (std::fclose(f) != 0 || !ok) return fail(CGRT_ERR_IO, std::string("write failed: ") + path);
    return CGRT_OK;
}
