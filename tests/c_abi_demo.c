/* A caller that knows nothing but include/pccm.h: plain C, no Python, no torch.  Reads two clouds (doubles, [n][3])
 * from a binary file written by tests/test_gpu_c_abi.py, runs both directional searches and a D2 reduction on the GPU
 * and writes idx / d2 / the reduction back for the test to compare with the oracle. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "pccm.h"

#define CHECK(call)                                                              \
    do {                                                                         \
        int rc_ = (call);                                                        \
        if (rc_ != PCCM_OK) {                                                    \
            fprintf(stderr, "%s -> %d: %s\n", #call, rc_, pccm_last_error());    \
            return 1;                                                            \
        }                                                                        \
    } while (0)

int main(int argc, char **argv)
{
    if (argc != 3) return 2;
    FILE *in = fopen(argv[1], "rb");
    if (!in) return 2;
    int64_t n[2];
    if (fread(n, sizeof(int64_t), 2, in) != 2) return 2;
    double *pts[2], *nrm[2];
    for (int k = 0; k < 2; ++k) {
        pts[k] = malloc((size_t)n[k] * 3 * sizeof(double));
        nrm[k] = malloc((size_t)n[k] * 3 * sizeof(double));
        if (fread(pts[k], sizeof(double), (size_t)n[k] * 3, in) != (size_t)n[k] * 3) return 2;
        if (fread(nrm[k], sizeof(double), (size_t)n[k] * 3, in) != (size_t)n[k] * 3) return 2;
    }
    fclose(in);

    pccm_ctx *ctx = NULL;
    CHECK(pccm_ctx_create(0, NULL, &ctx));
    for (int k = 0; k < 2; ++k) {
        CHECK(pccm_set_cloud(ctx, k, pts[k], n[k], PCCM_F64, 0));
        CHECK(pccm_set_normals(ctx, k, nrm[k], n[k], PCCM_F64, 0));
    }
    CHECK(pccm_nn_pair(ctx, PCCM_ENGINE_AUTO));

    FILE *out = fopen(argv[2], "wb");
    if (!out) return 2;
    for (int dir = 0; dir < 2; ++dir) {
        const int64_t m = n[dir];
        int32_t *idx = malloc((size_t)m * sizeof(int32_t));
        double *d2 = malloc((size_t)m * sizeof(double));
        CHECK(pccm_nn_fetch(ctx, dir, idx, d2));
        fwrite(idx, sizeof(int32_t), (size_t)m, out);
        fwrite(d2, sizeof(double), (size_t)m, out);
        double total[3];                                   /* np.sum, np.min, np.max of the D2 column */
        CHECK(pccm_reduce_total(ctx, dir, PCCM_METRIC_D2, PCCM_NORMAL_NEIGHBOUR, total));
        fwrite(total, sizeof(double), 3, out);
        free(idx);
        free(d2);
    }
    fclose(out);
    CHECK(pccm_ctx_destroy(ctx));
    printf("pccm %d ok\n", pccm_version());
    return 0;
}
