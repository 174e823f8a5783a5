// exercises the oracle under ASan/UBSan on CPU (GPU sanitizers are not available on the pool)
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../../oracle/mvs_oracle.h"
static unsigned long long s = 88172645463325252ULL;
static double rnd(void) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (double)(s >> 11) / 9007199254740992.0; }
int main(void)
{
    enum { N = 300 };
    unsigned char *d1 = malloc(N * 32), *d2 = malloc(N * 32);
    float *k1 = malloc(N * 2 * sizeof(float)), *k2 = malloc(N * 2 * sizeof(float));
    for (int i = 0; i < N * 32; ++i) d1[i] = (unsigned char)(rnd() * 256);
    memcpy(d2, d1, N * 32);
    for (int i = 0; i < N; ++i) {
        double X = rnd() * 4 - 2, Y = rnd() * 3 - 1.5, Z = 4 + rnd() * 5;
        k1[2 * i] = (float)(525 * X / Z + 320); k1[2 * i + 1] = (float)(525 * Y / Z + 240);
        k2[2 * i] = (float)(525 * (X + 0.3) / Z + 320); k2[2 * i + 1] = (float)(525 * Y / Z + 240);
        d2[i * 32 + (i % 32)] ^= 1;
    }
    double K[9] = {525, 0, 320, 0, 525, 240, 0, 0, 1};
    orc_params prm = {1e-2, 200, ORC_SAMPLER_PHILOX, 5, 8};
    orc_match *m = malloc(N * sizeof(orc_match));
    orc_two_view_result res;
    unsigned char *mask = malloc(N);
    double *pts = malloc(N * 3 * sizeof(double));
    int64_t *idx = malloc(N * sizeof(int64_t));
    int ok = orc_image_pair(d1, k1, N, d2, k2, N, 32, 0.7, 10.0, K, &prm, m, &res, mask, pts, idx);
    printf("image_pair ok=%d M=%d inl=%d pts=%d\n", ok, res.n_matches, res.n_inliers, res.n_points);
    // degenerate / edge inputs
    orc_image_pair(d1, k1, 1, d2, k2, 5, 32, 0.7, 10.0, K, &prm, m, &res, mask, pts, idx);
    orc_image_pair(d1, k1, 7, d2, k2, 7, 32, 0.99, -1.0, K, &prm, m, &res, mask, pts, idx);
    double A[6] = {1, 1, 0, 0, 1, 1}, w[3], u[9], vt[9];
    orc_svd(A, 2, 3, w, u, vt);
    double E[9] = {0, 0, 0, 0, 0, -1, 0, 1, 0}, Ra[9], Rb[9], t[3];
    orc_decompose_essential(E, Ra, Rb, t);
    // pnp
    double Xw[3 * 40], uv[2 * 40], R[9], tt[3];
    for (int i = 0; i < 40; ++i) {
        Xw[3 * i] = rnd() * 2 - 1; Xw[3 * i + 1] = rnd() * 2 - 1; Xw[3 * i + 2] = 4 + rnd();
        uv[2 * i] = 525 * Xw[3 * i] / Xw[3 * i + 2] + 320; uv[2 * i + 1] = 525 * Xw[3 * i + 1] / Xw[3 * i + 2] + 240;
    }
    orc_pnp_params pp = {64, ORC_SAMPLER_PHILOX, 3, 0.05, 4};
    int ni = 0, bh = -1;
    int okp = orc_pnp_solve(Xw, uv, 40, K, &pp, R, tt, idx, &ni, NULL, NULL, &bh);
    printf("pnp ok=%d inliers=%d\n", okp, ni);
    // refinement (row f4): two-view on 300 points (more than one point per partial sum), then motion-only
    {
        enum { M = 300 };
        static double q1[2 * M], q2[2 * M], Xg[3 * M], Xo[3 * M], pc[9 * M], wc[9 * M];
        const double Rg[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, tg[3] = {0.3, 0.0, 0.0};
        for (int i = 0; i < M; ++i) {
            double x = rnd() * 2 - 1, y = rnd() * 2 - 1, z = 4 + rnd();
            q1[2 * i] = 525 * x / z + 320 + (rnd() - 0.5); q1[2 * i + 1] = 525 * y / z + 240 + (rnd() - 0.5);
            q2[2 * i] = 525 * (x - 0.3) / z + 320 + (rnd() - 0.5); q2[2 * i + 1] = 525 * y / z + 240 + (rnd() - 0.5);
            Xg[3 * i] = x + 0.01 * (rnd() - 0.5); Xg[3 * i + 1] = y; Xg[3 * i + 2] = z;
            for (int k = 0; k < 9; ++k) wc[9 * i + k] = (k % 4 == 0) ? 1e-4 : 0.0;
        }
        orc_refine_params rp;
        orc_refine_params_default(&rp);
        double Ro[9], to[3], cov[36], err = 0;
        int it = 0;
        int okr = orc_sfm_refine(q1, NULL, q2, NULL, M, K, Rg, tg, Xg, &rp, Ro, to, cov, Xo, pc, &err, &it);
        int okq = orc_pnp_refine(Xg, wc, q1, NULL, M, K, Rg, tg, &rp, Ro, to, cov, &err, &it);
        okq &= orc_sfm_refine(q1, NULL, q2, NULL, 1, K, Rg, tg, Xg, &rp, Ro, to, cov, Xo, pc, &err, &it);
        printf("refine ok=%d %d\n", okr, okq);
    }
    // extraction (row f3) on a blocky 200 x 160 image, all levels
    {
        enum { W = 200, H = 160, NF = 300 };
        static uint8_t im[W * H], dd[NF * 32];
        static orc_keypoint kk[NF];
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x)
                im[y * W + x] = (uint8_t)((((x / 6) * 37 + (y / 6) * 101) * 2654435761u) >> 24);
        orc_orb_params op;
        orc_orb_params_default(&op);
        op.nfeatures = NF;
        int nk = 0;
        int oke = orc_orb_extract(im, W, H, &op, kk, dd, &nk);
        printf("orb ok=%d n=%d\n", oke, nk > 50);
        // small feature counts: the rounded per-level shares add up to more than nfeatures (7 -> 2+1+1+1+1+1+1 = 8);
        // outputs sized for exactly nfeatures must not be overrun
        for (int nf = 1; nf <= 24; ++nf) {
            orc_keypoint *k2 = (orc_keypoint *)malloc(sizeof(orc_keypoint) * nf);
            uint8_t *d2 = (uint8_t *)malloc(32 * (size_t)nf);
            op.nfeatures = nf;
            op.fast_threshold = 5;
            int n2 = 0;
            oke &= orc_orb_extract(im, W, H, &op, k2, d2, &n2) && n2 <= nf;
            free(k2);
            free(d2);
        }
        printf("orb small ok=%d\n", oke);
    }
    free(d1); free(d2); free(k1); free(k2); free(m); free(mask); free(pts); free(idx);
    return 0;
}
