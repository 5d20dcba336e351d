/* TEST INFRASTRUCTURE. Plain-C structs shared by the CPU oracle (liborc.so) and
 * the compiled-reference harness (oracle/_ref/libcgrt_ref.so).  The product's
 * C ABI lives in include/cgrt.h and does not include this file. */
#ifndef CGRT_TESTAPI_H
#define CGRT_TESTAPI_H
#include <stdint.h>

typedef struct {
    double cam[3];       /* main.cpp:181  camorg = (0,0,-10)                 */
    double half_width;   /* main.cpp:188  the literal 10.0                    */
    double focus_plane;  /* main.cpp:178  20.0                                */
    double lens_radius;  /* main.cpp:179  1.5; 0 => pinhole call main.cpp:209,
                            >0 => thin-lens call main.cpp:207                 */
} orc_camera;

typedef struct {
    int32_t W, H;          /* global image size (camera formulas use these)   */
    int32_t row0, nrows;   /* rows [row0,row0+nrows) rendered; row 0 = bottom */
    int32_t spp, sample0;  /* samples [sample0, sample0+spp)                  */
    int32_t depth;         /* recursion budget (reference MAX_DEPTH = 5)      */
    int32_t pad_;
    uint64_t seed;
} orc_grid;

/* outputs of *_trace_grid:
 *   acc   [nrows*W*3] double : sum over samples and hitpoints of f (=f*adj, main.cpp:88)
 *   nhit  [nrows*W]   uint32 : number of hitpoints stored for the pixel
 *   nrays             uint64 : trace() invocations that passed the depth test
 *   hp    [cap*9]     double : optional hitpoint stream f(3) pos(3) normal(3), emission order
 *   hp_pix[cap]       int64  : (sample << 32) | local pixel index
 */

/* Photon pass (main.cpp:223-258).  Deterministic SERIAL semantics: photons are
 * traced one after another in index order on one thread, photon i drawing from the keyed stream
 * cgrt_key(seed, i, 0, CGRT_PURPOSE_PHOTON) in the reference's own call order. */
#define CGRT_PURPOSE_PHOTON 0x70686f74ULL
typedef struct {
    double light[3];   /* main.cpp:180  lightorg = (0,19.999,20)                                   */
    double jitter;     /* main.cpp:240-241  a,b = u*4-2: half extent 2.0 of the square emitter      */
    double power;      /* main.cpp:246  Vec3(700,700,700) * (PI*4.0): the 700                       */
    double alpha;      /* main.cpp:36   0.7                                                         */
    int64_t nphotons;  /* photons traced in total (reference: num_photon * num_threads, main.cpp:223-224) */
    int32_t hashsize;  /* main.cpp:184  1000001                                                     */
    int32_t pad_;
    uint64_t seed;
} orc_photons;
/* per-hitpoint output record of *_ppm: 16 doubles
 *   [0] local pixel index  [1] sample index  [2..4] f  [5..7] pos  [8..10] normal  [11..13] flux  [14] r2  [15] n */
#endif
