// gsvc_amd/csrc/knn.hip — mean squared distance to the 3 nearest neighbours of every point, gfx950.
//
// Replaces simple_knn._C.distCUDA2 (reference submodules/simple-knn.zip, simple_knn.cu:63-218; used once at model creation,
// scene/gaussian_model.py:762,784: the initial anchor scales).  The reference Morton-sorts the points and prunes boxes
// of 1024; here the caller bins the points into a uniform grid (a few points per cell, points sorted by cell) and one
// lane per point walks growing cubes of cells around its own cell until the third-best distance found cannot be beaten
// from outside the cube: exact 3-NN, expected work O(1) per point for the roughly uniform anchor clouds GSVC starts from.
#include "common.h"

namespace gsvc {

struct KnnGrid {
    float ox, oy, oz, edge;      // origin of cell (0,0,0), cell edge
    int gx, gy, gz;
};

__global__ void __launch_bounds__(256) k_knn3(const float *__restrict__ pts, const int32_t *__restrict__ cell_start, KnnGrid g,
                                              int64_t n, float *__restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float px = pts[3 * i], py = pts[3 * i + 1], pz = pts[3 * i + 2];
    const int cx = min(max((int)floorf((px - g.ox) / g.edge), 0), g.gx - 1);
    const int cy = min(max((int)floorf((py - g.oy) / g.edge), 0), g.gy - 1);
    const int cz = min(max((int)floorf((pz - g.oz) / g.edge), 0), g.gz - 1);
    float b0 = 3.0e38f, b1 = 3.0e38f, b2 = 3.0e38f;      // three smallest squared distances, ascending
    const int rmax = max(g.gx, max(g.gy, g.gz));
    for (int r = 0; r <= rmax; r++) {
        // the shell of cells at Chebyshev distance exactly r
        const int z0 = max(cz - r, 0), z1 = min(cz + r, g.gz - 1);
        const int y0 = max(cy - r, 0), y1 = min(cy + r, g.gy - 1);
        const int x0 = max(cx - r, 0), x1 = min(cx + r, g.gx - 1);
        for (int z = z0; z <= z1; z++)
            for (int y = y0; y <= y1; y++) {
                const bool face = (z == cz - r) || (z == cz + r) || (y == cy - r) || (y == cy + r);
                const int step = (face || r == 0) ? 1 : max(x1 - x0, 1);   // interior rows of the shell: only the two end cells
                for (int x = x0; x <= x1; x += step) {
                    if (!face && r > 0 && x != cx - r && x != cx + r) continue;
                    const int64_t c = ((int64_t)z * g.gy + y) * g.gx + x;
                    const int s = cell_start[c], e = cell_start[c + 1];
                    for (int j = s; j < e; j++) {
                        if (j == i) continue;
                        const float dx = pts[3 * (int64_t)j] - px, dy = pts[3 * (int64_t)j + 1] - py, dz = pts[3 * (int64_t)j + 2] - pz;
                        const float d = dx * dx + dy * dy + dz * dz;
                        if (d < b2) {
                            if (d < b1) {
                                b2 = b1;
                                if (d < b0) { b1 = b0; b0 = d; } else b1 = d;
                            } else b2 = d;
                        }
                    }
                }
            }
        // every point outside the cube of radius r is at least r * edge away (the point lies inside the centre cell)
        const float reach = (float)r * g.edge;
        if (b2 <= reach * reach) break;
    }
    const int have = (b0 < 3.0e38f) + (b1 < 3.0e38f) + (b2 < 3.0e38f);
    float sum = 0.f;
    if (have > 0) sum += b0;
    if (have > 1) sum += b1;
    if (have > 2) sum += b2;
    out[i] = have ? sum / (float)have : 0.f;
}

}  // namespace gsvc

using namespace gsvc;

extern "C" int gsvc_knn3_mean_dist2(const float *points_by_cell, const int32_t *cell_start, const float *origin3_host, float cell_edge,
                                    int32_t gx, int32_t gy, int32_t gz, int64_t n, float *out, void *stream)
{
    GSVC_REQUIRE(n >= 0 && gx > 0 && gy > 0 && gz > 0 && cell_edge > 0.f && origin3_host, "knn3: bad arguments");
    GSVC_REQUIRE(n < ((int64_t)1 << 31), "knn3: too many points");
    if (n == 0) return GSVC_OK;
    GSVC_REQUIRE(points_by_cell && cell_start && out, "knn3: NULL pointer");
    KnnGrid g{origin3_host[0], origin3_host[1], origin3_host[2], cell_edge, gx, gy, gz};
    hipStream_t s = (hipStream_t)stream;
    ProfScope _p("k_knn3", s);
    hipLaunchKernelGGL(k_knn3, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, points_by_cell, cell_start, g, n, out);
    return check_launch("knn3");
}
