/*
 * mvs_orb_oracle.c -- CPU ORACLE for row f3 of SURVEY.md section 8: keypoint + descriptor extraction
 * (VisualFeature::extract, vision/visual-feature.cpp:12-17,40-49 = cv::ORB::create(500) detect + compute).
 * TEST INFRASTRUCTURE ONLY (see mvs_oracle.h).
 *
 * PARITY UNPINNED: cv::ORB lives in OpenCV (absent, version unpinned) and its learned 256-pair sampling pattern
 * (bit_pattern_31_) exists nowhere in the reference tree, so keypoints and descriptors cannot be made identical to
 * the reference's.  What is restated here is ORB's PUBLISHED pipeline with the reference's parameters (500 features,
 * scale 1.2, 8 levels, edge 31, patch 31, FAST threshold 20, Harris score -- the cv::ORB::create defaults):
 *   pyramid -> FAST-9/16 + 3x3 non-maximum suppression -> keep 2 n_l best by FAST score -> Harris response
 *   (7x7, k = 0.04) -> keep n_l best -> intensity-centroid orientation (radius 15) -> 7x7 sigma-2 blur ->
 *   steered BRIEF, 256 tests, LSB-first bytes.
 * Own, fully specified choices (shared with the HIP kernels by specification, so that GPU and oracle agree BIT FOR BIT):
 *   resize    bilinear from the previous level, pixel-centre aligned, integer arithmetic: 11-bit weights from exact
 *             rationals, (sum + 2^21) >> 22
 *   blur      Q8 kernel {18, 34, 49, 54, 49, 34, 18}, BORDER_REFLECT_101, rows then columns, (sum + 2^15) >> 16
 *   order     candidates are ranked by (score desc, y asc, x asc): no dependence on detection order
 *   steering  cos / sin straight from the moments (m10, m01) / hypot in float -- no trigonometric library call;
 *             KeyPoint::angle is OpenCV's fastAtan2 polynomial (degrees)
 *   pattern   256 point pairs from Philox4x32-10 (key 0x0B5EED00, 0x31): each coordinate is a centred sum of four
 *             16-bit uniforms scaled by 11 / 65536 (sigma ~ 6.3 = patch / 5, BRIEF's G II) clipped to [-13, 13]
 * The oracle is pinned by independent numpy restatements of the definitions (tests/test_orb.py: FAST by brute force,
 * Harris, moments, blur, resize) and by behaviour (rotation covariance, repeatability, the tsukuba pair).
 */
#include "mvs_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define ORB_MAX_LEVELS 16

void orc_orb_params_default(orc_orb_params *p)
{
    p->nfeatures = 500;     /* visual-feature.cpp:9 MAX_FEATURE_COUNT */
    p->nlevels = 8;         /* cv::ORB::create defaults */
    p->edge_threshold = 31;
    p->fast_threshold = 20;
}

void orc_orb_pattern(int8_t P[256 * 4])
{
    const uint32_t key[2] = {0x0B5EED00u, 0x31u};
    for (int i = 0; i < 256; ++i) {
        uint32_t w[8];
        const uint32_t c0[4] = {(uint32_t)i, 0, 0, 0}, c1[4] = {(uint32_t)i, 1, 0, 0};
        orc_philox4x32_10(c0, key, w);
        orc_philox4x32_10(c1, key, w + 4);
        int c[4];
        for (int k = 0; k < 4; ++k) {
            const uint32_t a = w[2 * k], b = w[2 * k + 1];
            const int u = (int)((a & 0xffffu) + (a >> 16) + (b & 0xffffu) + (b >> 16)) - 131072;
            int v = (u * 11) / 65536; /* truncation toward zero */
            if (v > 13) v = 13;
            if (v < -13) v = -13;
            c[k] = v;
        }
        if (c[0] == c[2] && c[1] == c[3])
            c[2] = c[0] + (c[0] < 0 ? 3 : -3);
        for (int k = 0; k < 4; ++k)
            P[4 * i + k] = (int8_t)c[k];
    }
}

int orc_orb_layout(int w, int h, const orc_orb_params *p, int *lw, int *lh, int *nl, double *scale)
{
    if (p->nlevels < 1 || p->nlevels > ORB_MAX_LEVELS || p->nfeatures < 1)
        return 0;
    double s = 1.0;
    for (int l = 0; l < p->nlevels; ++l) {
        scale[l] = s;
        lw[l] = (int)lrint((double)w / s);
        lh[l] = (int)lrint((double)h / s);
        s = s * 1.2;
    }
    const double factor = 1.0 / 1.2;
    double fn = 1.0;
    for (int l = 0; l < p->nlevels; ++l)
        fn = fn * factor;
    double nd = (double)p->nfeatures * (1.0 - factor) / (1.0 - fn);
    int sum = 0;
    for (int l = 0; l < p->nlevels - 1; ++l) {
        /* cv::ORB rounds every level's share; for small nfeatures (< ~60 at 8 levels) the rounded shares can add up to
         * MORE than nfeatures, and cv::ORB then returns more keypoints than it was asked for.  The outputs here have
         * room for nfeatures: a level takes at most what is left (the kernel's host side clamps the same way). */
        nl[l] = (int)lrint(nd);
        if (nl[l] > p->nfeatures - sum)
            nl[l] = p->nfeatures - sum;
        sum += nl[l];
        nd = nd * factor;
    }
    nl[p->nlevels - 1] = p->nfeatures - sum > 0 ? p->nfeatures - sum : 0;
    return 1;
}

void orc_orb_resize(const uint8_t *src, int sw, int sh, uint8_t *dst, int dw, int dh)
{
    for (int dy = 0; dy < dh; ++dy) {
        const long ny = (long)(2 * dy + 1) * sh - dh, dny = 2L * dh;
        long sy = ny >= 0 ? ny / dny : -1;
        long fy = ny - sy * dny;
        int wy = (int)((fy * 4096 + dny) / (2 * dny));
        if (sy < 0) sy = 0, wy = 0;
        long sy1 = sy + 1;
        if (sy >= sh - 1) sy = sh - 1, sy1 = sh - 1;
        for (int dx = 0; dx < dw; ++dx) {
            const long nx = (long)(2 * dx + 1) * sw - dw, dnx = 2L * dw;
            long sx = nx >= 0 ? nx / dnx : -1;
            long fx = nx - sx * dnx;
            int wx = (int)((fx * 4096 + dnx) / (2 * dnx));
            if (sx < 0) sx = 0, wx = 0;
            long sx1 = sx + 1;
            if (sx >= sw - 1) sx = sw - 1, sx1 = sw - 1;
            const uint32_t p00 = src[sy * sw + sx], p01 = src[sy * sw + sx1], p10 = src[sy1 * sw + sx],
                           p11 = src[sy1 * sw + sx1];
            const uint32_t v = p00 * (uint32_t)((2048 - wx) * (2048 - wy)) + p01 * (uint32_t)(wx * (2048 - wy)) +
                               p10 * (uint32_t)((2048 - wx) * wy) + p11 * (uint32_t)(wx * wy);
            dst[(long)dy * dw + dx] = (uint8_t)((v + (1u << 21)) >> 22);
        }
    }
}

static const int FAST_DX[16] = {0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1};
static const int FAST_DY[16] = {3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1, 0, 1, 2, 3};

/* largest t for which (x, y) is a FAST-9 corner at threshold t, i.e. s - 1 with
 * s = max over the 16 arcs of 9 contiguous circle pixels of min(+-(circle - centre)); <0: none */
static int fast_score_at(const uint8_t *img, int w, int x, int y)
{
    int d[16];
    const int c = img[y * w + x];
    for (int k = 0; k < 16; ++k)
        d[k] = (int)img[(y + FAST_DY[k]) * w + (x + FAST_DX[k])] - c;
    int best = -256;
    for (int k = 0; k < 16; ++k) {
        int mb = 255, md = 255;
        for (int j = 0; j < 9; ++j) {
            const int v = d[(k + j) & 15];
            if (v < mb) mb = v;
            if (-v < md) md = -v;
        }
        if (mb > best) best = mb;
        if (md > best) best = md;
    }
    return best - 1;
}

void orc_orb_fast_scores(const uint8_t *img, int w, int h, int threshold, uint8_t *score)
{
    memset(score, 0, (size_t)w * h);
    for (int y = 3; y < h - 3; ++y)
        for (int x = 3; x < w - 3; ++x) {
            const int s = fast_score_at(img, w, x, y);
            if (s >= threshold)
                score[y * w + x] = (uint8_t)s;
        }
}

void orc_orb_blur(const uint8_t *img, int w, int h, uint8_t *out)
{
    static const int K[7] = {18, 34, 49, 54, 49, 34, 18};
    uint16_t *tmp = (uint16_t *)malloc(sizeof(uint16_t) * (size_t)w * h);
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            int s = 0;
            for (int k = -3; k <= 3; ++k) {
                int xx = x + k;
                if (xx < 0) xx = -xx;
                if (xx >= w) xx = 2 * (w - 1) - xx;
                s += K[k + 3] * img[y * w + xx];
            }
            tmp[y * w + x] = (uint16_t)s;
        }
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            uint32_t s = 0;
            for (int k = -3; k <= 3; ++k) {
                int yy = y + k;
                if (yy < 0) yy = -yy;
                if (yy >= h) yy = 2 * (h - 1) - yy;
                s += (uint32_t)K[k + 3] * tmp[yy * w + x];
            }
            out[y * w + x] = (uint8_t)((s + 32768u) >> 16);
        }
    free(tmp);
}

float orc_orb_harris(const uint8_t *img, int w, int x0, int y0)
{
    int a = 0, b = 0, c = 0;
    for (int dy = -3; dy <= 3; ++dy)
        for (int dx = -3; dx <= 3; ++dx) {
            const uint8_t *p = img + (y0 + dy) * w + (x0 + dx);
            const int Ix = (p[1] - p[-1]) * 2 + (p[-w + 1] - p[-w - 1]) + (p[w + 1] - p[w - 1]);
            const int Iy = (p[w] - p[-w]) * 2 + (p[w - 1] - p[-w - 1]) + (p[w + 1] - p[-w + 1]);
            a += Ix * Ix;
            b += Iy * Iy;
            c += Ix * Iy;
        }
    const float scale = 1.0f / (4.0f * 7.0f * 255.0f);
    const float scale4 = ((scale * scale) * scale) * scale;
    const float fa = (float)a, fb = (float)b, fc = (float)c;
    const float t1 = fa * fb, t2 = fc * fc, t3 = fa + fb;
    return ((t1 - t2) - (0.04f * t3) * t3) * scale4;
}

static const int UMAX[16] = {15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3};

void orc_orb_moments(const uint8_t *img, int w, int x0, int y0, int *m10, int *m01)
{
    const uint8_t *c = img + y0 * w + x0;
    int s10 = 0, s01 = 0;
    for (int u = -15; u <= 15; ++u)
        s10 += u * c[u];
    for (int v = 1; v <= 15; ++v) {
        int vs = 0;
        const int d = UMAX[v];
        for (int u = -d; u <= d; ++u) {
            const int vp = c[u + v * w], vm = c[u - v * w];
            vs += vp - vm;
            s10 += u * (vp + vm);
        }
        s01 += v * vs;
    }
    *m10 = s10;
    *m01 = s01;
}

float orc_orb_fast_atan2(float y, float x)
{ /* OpenCV's fastAtan2 polynomial, degrees in [0, 360) */
    const float p1 = 0.9997878412794807f * 57.29577951308232f, p3 = -0.3258083974640975f * 57.29577951308232f,
                p5 = 0.1555786518463281f * 57.29577951308232f, p7 = -0.04432655554792128f * 57.29577951308232f;
    const float ax = fabsf(x), ay = fabsf(y);
    float a, c, c2;
    if (ax >= ay) {
        c = ay / (ax + 2.220446049250313e-16f);
        c2 = c * c;
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    } else {
        c = ax / (ay + 2.220446049250313e-16f);
        c2 = c * c;
        a = 90.0f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
    if (x < 0.0f) a = 180.0f - a;
    if (y < 0.0f) a = 360.0f - a;
    return a;
}

typedef struct {
    uint64_t key;
    int x, y, score;
    float harris;
} cand;

static int cand_cmp(const void *a, const void *b)
{
    const uint64_t ka = ((const cand *)a)->key, kb = ((const cand *)b)->key;
    return ka < kb ? -1 : (ka > kb ? 1 : 0);
}

/* ascending key = (descending value, ascending y, ascending x) */
static uint64_t rank_key(uint32_t value_desc, int y, int x)
{
    return ((uint64_t)(0xffffffffu - value_desc) << 32) | ((uint64_t)(uint32_t)y << 16) | (uint32_t)x;
}
static uint32_t float_ordered(float f)
{
    uint32_t u;
    memcpy(&u, &f, 4);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

int orc_orb_extract(const uint8_t *image, int w, int h, const orc_orb_params *prm, orc_keypoint *kp, uint8_t *desc,
                    int *n_out)
{
    *n_out = 0;
    int lw[ORB_MAX_LEVELS], lh[ORB_MAX_LEVELS], nl[ORB_MAX_LEVELS];
    double scale[ORB_MAX_LEVELS];
    if (w < 1 || h < 1 || prm->fast_threshold < 1 || prm->fast_threshold > 254 || prm->edge_threshold < 19 ||
        !orc_orb_layout(w, h, prm, lw, lh, nl, scale))
        return 0;
    int8_t P[256 * 4];
    orc_orb_pattern(P);
    uint8_t *prev = NULL;
    int n = 0;
    for (int l = 0; l < prm->nlevels; ++l) {
        const int W = lw[l], H = lh[l];
        if (W < 1 || H < 1)
            break;
        uint8_t *img = (uint8_t *)malloc((size_t)W * H);
        if (l == 0)
            memcpy(img, image, (size_t)W * H);
        else
            orc_orb_resize(prev, lw[l - 1], lh[l - 1], img, W, H);
        free(prev);
        prev = img;
        const int e = prm->edge_threshold;
        if (W <= 2 * e || H <= 2 * e || nl[l] < 1)
            continue;
        uint8_t *score = (uint8_t *)malloc((size_t)W * H), *blur = (uint8_t *)malloc((size_t)W * H);
        orc_orb_fast_scores(img, W, H, prm->fast_threshold, score);
        int cap = 1024, nc = 0;
        cand *c = (cand *)malloc(sizeof(cand) * cap);
        for (int y = e; y < H - e; ++y)
            for (int x = e; x < W - e; ++x) {
                const int s = score[y * W + x];
                if (!s)
                    continue;
                int is_max = 1;
                for (int dy = -1; dy <= 1 && is_max; ++dy)
                    for (int dx = -1; dx <= 1; ++dx)
                        if ((dx || dy) && score[(y + dy) * W + x + dx] >= s) {
                            is_max = 0;
                            break;
                        }
                if (!is_max)
                    continue;
                if (nc == cap) {
                    cap *= 2;
                    c = (cand *)realloc(c, sizeof(cand) * cap);
                }
                c[nc].x = x, c[nc].y = y, c[nc].score = s;
                c[nc].harris = orc_orb_harris(img, W, x, y);
                c[nc].key = rank_key((uint32_t)s, y, x);
                ++nc;
            }
        qsort(c, nc, sizeof(cand), cand_cmp);             /* retainBest(2 n_l) by FAST score */
        if (nc > 2 * nl[l])
            nc = 2 * nl[l];
        for (int i = 0; i < nc; ++i)
            c[i].key = rank_key(float_ordered(c[i].harris), c[i].y, c[i].x);
        qsort(c, nc, sizeof(cand), cand_cmp);             /* retainBest(n_l) by Harris response */
        if (nc > nl[l])
            nc = nl[l];
        orc_orb_blur(img, W, H, blur);
        const float fs = (float)scale[l];
        for (int i = 0; i < nc; ++i) {
            int m10, m01;
            orc_orb_moments(img, W, c[i].x, c[i].y, &m10, &m01);
            const float f10 = (float)m10, f01 = (float)m01;
            const float h2 = f10 * f10 + f01 * f01;
            float ca = 1.0f, sa = 0.0f;
            if (h2 > 0.0f) {
                const float hh = sqrtf(h2);
                ca = f10 / hh;
                sa = f01 / hh;
            }
            orc_keypoint *k = kp + n;
            k->x = (float)c[i].x * fs;
            k->y = (float)c[i].y * fs;
            k->size = 31.0f * fs;
            k->angle = orc_orb_fast_atan2(f01, f10);
            k->response = c[i].harris;
            k->octave = l;
            k->class_id = -1;
            uint8_t *d = desc + 32 * (size_t)n;
            const uint8_t *ctr = blur + c[i].y * W + c[i].x;
            for (int byte = 0; byte < 32; ++byte) {
                unsigned v = 0;
                for (int bit = 0; bit < 8; ++bit) {
                    const int8_t *q = P + 4 * (8 * byte + bit);
                    const float x1 = (float)q[0], y1 = (float)q[1], x2 = (float)q[2], y2 = (float)q[3];
                    const int ix1 = (int)rintf(x1 * ca - y1 * sa), iy1 = (int)rintf(x1 * sa + y1 * ca);
                    const int ix2 = (int)rintf(x2 * ca - y2 * sa), iy2 = (int)rintf(x2 * sa + y2 * ca);
                    v |= (unsigned)(ctr[iy1 * W + ix1] < ctr[iy2 * W + ix2]) << bit;
                }
                d[byte] = (uint8_t)v;
            }
            ++n;
        }
        free(c), free(score), free(blur);
    }
    free(prev);
    *n_out = n;
    return 1;
}
