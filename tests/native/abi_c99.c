/* include/cgrt.h must be a plain-C header: compiled with gcc -std=c99 -pedantic -Wall -Werror by
 * tests/test_capi_host.py.  Also pins the struct layouts the ctypes binding assumes. */
#include "cgrt.h"

typedef char check_camera_size[(sizeof(cgrt_camera) == 48) ? 1 : -1];
typedef char check_grid_size[(sizeof(cgrt_grid) == 56) ? 1 : -1];
typedef char check_stats_size[(sizeof(cgrt_scene_stats) == 64) ? 1 : -1];
typedef char check_photons_size[(sizeof(cgrt_photons) == 88) ? 1 : -1];
typedef char check_ppm_result_size[(sizeof(cgrt_ppm_result) == 96) ? 1 : -1];

int cgrt_abi_smoke(void) {
    cgrt_scene *s = 0;
    cgrt_camera cam = {{0.0, 0.0, -10.0}, 10.0, 20.0, 0.0};
    cgrt_grid g;
    double c[3] = {0.0, 0.0, 30.0}, col[3] = {1.0, 1.0, 1.0};
    int rc;
    g.width = 8; g.height = 8; g.rows = 8; g.row_offset = 0; g.stripe_rows = 0; g.stripe_rank = 0; g.stripe_nranks = 1;
    g.spp = 1; g.sample_offset = 0; g.spp_total = 1; g.max_depth = 5; g.flags = CGRT_GRID_STATS; g.seed = 1;
    if (cgrt_version() != CGRT_VERSION) return 1;
    if (cgrt_scene_create(&s) != CGRT_OK) return 2;
    if (cgrt_scene_add_sphere(s, c, 5.0, col, 0.0, 0.0) != 0) return 3;
    {
        cgrt_photons ph = {{0.0, 19.999, 20.0}, 2.0, 700.0, 0.7, 1000, 1000001, 0, 777, 0.0, 0};
        cgrt_ppm_result out = {0, 0, 0, 0, 0, 0, 0, 0, 0.0, 0.0, 0.0, 0.0};
        if (cgrt_ppm_render(s, &cam, &g, &ph, &out) != CGRT_ERR_INVALID) return 5; /* uncommitted scene */
        if (cgrt_tonemap_rgb8(0, 0, 4, 4, 0) != CGRT_ERR_INVALID) return 6;          /* null buffers */
        if (cgrt_write_png(0, 4, 4, 0) != CGRT_ERR_INVALID) return 7;
    }
    rc = cgrt_trace_grid_host(s, &cam, &g, 0, 0, 0); /* uncommitted scene: must be refused, not crash */
    cgrt_scene_destroy(s);
    return rc == CGRT_ERR_INVALID ? 0 : 4;
}
