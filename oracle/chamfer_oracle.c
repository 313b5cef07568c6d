/* CPU oracle for the brute-force Chamfer nearest-neighbour search.
 * TEST INFRASTRUCTURE ONLY: loaded by tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg; the product path never links or calls it.
 *
 * PARITY UNPINNED.  The reference calls the un-vendored third-party package
 * `chamferdist` (krrish94/chamferdist, cloned at HEAD per README.md:32-38, no
 * pinned version; pre-1.0 API returning (dist1, dist2, idx1, idx2)) at
 * nnutils/loss_utils.py:85-86.  Its source is absent from /root/reference and
 * the reference holds no golden vector for it.  This file restates the
 * published algorithm of that package's NmDistanceKernel: for every point of
 * the first cloud the minimum SQUARED euclidean distance to the second cloud
 * and the index where it is attained, scanning candidates in increasing index
 * with a strict `<` so the lowest index wins ties.
 *
 * Arithmetic contract (what "bit-exact indices" is measured against):
 *   dx = x2 - x1, dy = y2 - y1, dz = z2 - z1          (each one fp32 rounding)
 *   d  = fmaf(dz, dz, fmaf(dy, dy, dx * dx))           (fused, as nvcc contracts
 *                                                        the upstream expression)
 * Build: gcc -O2 -ffp-contract=off -shared -fPIC (see oracle/Makefile).
 */
#include <math.h>
#include <stdint.h>

static void nn_one_way(const float *a, const float *b, int n, int m, float *dist, int32_t *idx)
{
    for (int i = 0; i < n; ++i) {
        const float x1 = a[3 * i], y1 = a[3 * i + 1], z1 = a[3 * i + 2];
        float best = INFINITY;
        int32_t besti = 0;
        for (int j = 0; j < m; ++j) {
            const float dx = b[3 * j] - x1;
            const float dy = b[3 * j + 1] - y1;
            const float dz = b[3 * j + 2] - z1;
            const float d = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
            if (d < best) {
                best = d;
                besti = j;
            }
        }
        dist[i] = best;
        idx[i] = besti;
    }
}

/* xyz1 [B,N,3], xyz2 [B,M,3] -> dist1 [B,N], dist2 [B,M], idx1 [B,N], idx2 [B,M] */
void chamfer_oracle_forward(const float *xyz1, const float *xyz2, int B, int N, int M,
                            float *dist1, float *dist2, int32_t *idx1, int32_t *idx2)
{
    for (int b = 0; b < B; ++b) {
        const float *p1 = xyz1 + (long)b * N * 3;
        const float *p2 = xyz2 + (long)b * M * 3;
        nn_one_way(p1, p2, N, M, dist1 + (long)b * N, idx1 + (long)b * N);
        nn_one_way(p2, p1, M, N, dist2 + (long)b * M, idx2 + (long)b * M);
    }
}

/* Backward with the indices frozen, in double accumulation:
 *   g1[i]      += 2 gd1[i] (x1[i] - x2[idx1[i]]),  g2[idx1[i]] -= the same
 *   g2[j]      += 2 gd2[j] (x2[j] - x1[idx2[j]]),  g1[idx2[j]] -= the same */
void chamfer_oracle_backward(const float *xyz1, const float *xyz2, int B, int N, int M,
                             const float *gd1, const float *gd2,
                             const int32_t *idx1, const int32_t *idx2,
                             double *g1, double *g2)
{
    for (long k = 0; k < (long)B * N * 3; ++k) g1[k] = 0.0;
    for (long k = 0; k < (long)B * M * 3; ++k) g2[k] = 0.0;
    for (int b = 0; b < B; ++b) {
        const float *p1 = xyz1 + (long)b * N * 3;
        const float *p2 = xyz2 + (long)b * M * 3;
        double *o1 = g1 + (long)b * N * 3;
        double *o2 = g2 + (long)b * M * 3;
        for (int i = 0; i < N; ++i) {
            const int j = idx1[(long)b * N + i];
            const double g = 2.0 * gd1[(long)b * N + i];
            for (int c = 0; c < 3; ++c) {
                const double u = (double)p1[3 * i + c] - (double)p2[3 * j + c];
                o1[3 * i + c] += g * u;
                o2[3 * j + c] -= g * u;
            }
        }
        for (int j = 0; j < M; ++j) {
            const int i = idx2[(long)b * M + j];
            const double g = 2.0 * gd2[(long)b * M + j];
            for (int c = 0; c < 3; ++c) {
                const double v = (double)p2[3 * j + c] - (double)p1[3 * i + c];
                o2[3 * j + c] += g * v;
                o1[3 * i + c] -= g * v;
            }
        }
    }
}
