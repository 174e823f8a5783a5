// orb.hip -- VisualFeature::extract (vision/visual-feature.cpp:12-17,40-49 = cv::ORB::create(500) detect + compute;
// SURVEY section 8 row f3) for a batch of equally sized grayscale images.
//
// cv::ORB is OpenCV-internal and its learned sampling pattern is not in the reference tree, so this is ORB's PUBLISHED
// pipeline with the reference's parameters and the build's own, fully specified choices where OpenCV's are out of
// reach (the CPU oracle oracle/mvs_orb_oracle.c follows the same specification bit for bit):
//   resize_kernel    level l from level l-1: pixel-centre bilinear in integer arithmetic (11-bit weights from exact
//                    rationals, (sum + 2^21) >> 22)                                           thread per pixel
//   fast_nms_kernel  FAST-9/16 score (the largest threshold at which the pixel is still a corner) and the strict 3x3
//                    maximum of one 64x16 tile through LDS -> rank key (score desc, y, x), one atomic per tile; the
//                    ORDER of the list does not matter, the select kernel sorts it
//   select_kernel    one launch, one workgroup per (image, level): bitonic sort of the keys in LDS, keep 2 n_l, Harris response
//                    (7x7, k = 0.04) of those, sort again by (response desc, y, x), keep n_l     (cv::ORB's retainBest)
//   blur_kernel      7x7 sigma-2 Gaussian as the Q8 kernel {18,34,49,54,49,34,18}, BORDER_REFLECT_101, rows then columns
//                    of one tile through LDS
//   describe_kernel  half a wavefront per keypoint: intensity-centroid moments over the radius-15 disc (32 lanes split
//                    the disc rows, butterfly sum), cos / sin = moments / hypot (no trigonometry), 256 steered BRIEF tests
//                    on the blurred level (lane b < 32 builds byte b), cv::KeyPoint record
// Everything is integer or single IEEE float operations in a stated order, so GPU and oracle agree bit for bit.
#include "kernels.hpp"

#include "device_math.hpp"

namespace mvs {

namespace {

__constant__ int kFastDx[16] = {0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1};
__constant__ int kFastDy[16] = {3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1, 0, 1, 2, 3};
__constant__ int kUmax[16] = {15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3};
__constant__ int kGauss[7] = {18, 34, 49, 54, 49, 34, 18};

// tab: per destination column / row {source index, 11-bit weight of the next sample}, computed once per level on the
// host from the exact rationals ((2 d + 1) s_src - s_dst) / (2 s_dst) -- 64-bit divisions per pixel were 3/4 of this kernel
__global__ void resize_kernel(const uint8_t *src, int sw, int sh, uint8_t *dst, int dw, int dh, const int2 *xtab,
                              const int2 *ytab)
{
    const int dx = blockIdx.x * blockDim.x + threadIdx.x, dy = blockIdx.y * blockDim.y + threadIdx.y;
    if (dx >= dw || dy >= dh)
        return;
    src += (size_t)blockIdx.z * sw * sh;
    dst += (size_t)blockIdx.z * dw * dh;
    const int2 tx = xtab[dx], ty = ytab[dy];
    const int sx = tx.x, wx = tx.y, sy = ty.x, wy = ty.y;
    const int sx1 = min(sx + 1, sw - 1), sy1 = min(sy + 1, sh - 1);
    const uint32_t p00 = src[sy * sw + sx], p01 = src[sy * sw + sx1], p10 = src[sy1 * sw + sx], p11 = src[sy1 * sw + sx1];
    const uint32_t v = p00 * (uint32_t)((2048 - wx) * (2048 - wy)) + p01 * (uint32_t)(wx * (2048 - wy)) +
                       p10 * (uint32_t)((2048 - wx) * wy) + p11 * (uint32_t)(wx * wy);
    dst[(size_t)dy * dw + dx] = (uint8_t)((v + (1u << 21)) >> 22);
}

__device__ __forceinline__ uint64_t rank_key(uint32_t value_desc, int y, int x)
{
    return ((uint64_t)(0xffffffffu - value_desc) << 32) | ((uint64_t)(uint32_t)y << 16) | (uint32_t)x;
}

// FAST-9/16 score of the pixel at LDS position c (row pitch P): the largest threshold at which it is still a corner,
// 0 if below `threshold`
__device__ __forceinline__ int fast_score_lds(const uint8_t *c, int P, int threshold)
{
    const int ctr = c[0];
    // every arc of 9 contiguous circle pixels contains one pixel of each antipodal pair: if both pixels of a pair are
    // within the threshold of the centre there is no corner (most pixels of a real image leave here)
    const int d0 = (int)c[3 * P] - ctr, d8 = (int)c[-3 * P] - ctr, d4 = (int)c[3] - ctr, d12 = (int)c[-3] - ctr;
    if (!((max(abs(d0), abs(d8)) > threshold) && (max(abs(d4), abs(d12)) > threshold)))
        return 0;
    int d[16];
#pragma unroll
    for (int k = 0; k < 16; ++k)
        d[k] = (int)c[kFastDy[k] * P + kFastDx[k]] - ctr;
    // min / max over every window of 9 by doubling: windows of 2, 4, 8, then one more element
    int lo2[16], hi2[16], lo4[16], hi4[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        lo2[k] = min(d[k], d[(k + 1) & 15]);
        hi2[k] = max(d[k], d[(k + 1) & 15]);
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        lo4[k] = min(lo2[k], lo2[(k + 2) & 15]);
        hi4[k] = max(hi2[k], hi2[(k + 2) & 15]);
    }
    int best = -256;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int lo9 = min(min(lo4[k], lo4[(k + 4) & 15]), d[(k + 8) & 15]);   // brighter arc: min of (ring - centre)
        const int hi9 = max(max(hi4[k], hi4[(k + 4) & 15]), d[(k + 8) & 15]);   // darker arc: min of (centre - ring) = -max
        best = max(best, max(lo9, -hi9));
    }
    const int sc = best - 1;
    return sc >= threshold ? sc : 0;
}

// FAST + non-maximum suppression of one 64 x 16 tile through LDS: the image patch (tile + 4) is read from HBM once, the
// scores of (tile + 1) never leave the CU, and the surviving corners are appended with ONE atomic per tile (all corners of
// a level append to the same counter and same-address atomics serialise: one per wavefront was 46 % of the extraction).
// Keys carry (score desc, y, x): the order of the list does not matter, select_kernel sorts it.
constexpr int kTileW = 64, kTileH = 16;
__global__ __launch_bounds__(256) void fast_nms_kernel(const uint8_t *img, int W, int H, int threshold, int edge, uint64_t *keys,
                                                       int32_t *count, int cap, int level, int n_levels)
{
    constexpr int PW = kTileW + 8, PH = kTileH + 8;      // image patch: +-4 (NMS 1 + circle 3)
    constexpr int SW = kTileW + 2, SH = kTileH + 2;      // score patch: +-1
    __shared__ uint8_t s_img[PH * PW];
    __shared__ uint8_t s_sc[SH * SW];
    __shared__ int wave_off[4];
    __shared__ int tile_base;
    const int b = blockIdx.z, tid = threadIdx.x;
    img += (size_t)b * W * H;
    const int x0 = edge + blockIdx.x * kTileW, y0 = edge + blockIdx.y * kTileH;   // first output pixel of the tile
    for (int i = tid; i < PH * PW; i += 256) {
        const int py = i / PW, px = i - py * PW;
        const int gx = min(max(x0 - 4 + px, 0), W - 1), gy = min(max(y0 - 4 + py, 0), H - 1);   // clamped reads are never used
        s_img[i] = img[gy * W + gx];
    }
    __syncthreads();
    for (int i = tid; i < SH * SW; i += 256) {
        const int qy = i / SW, qx = i - qy * SW;
        const int gx = x0 - 1 + qx, gy = y0 - 1 + qy;
        int sc = 0;
        if (gx >= 3 && gy >= 3 && gx < W - 3 && gy < H - 3)
            sc = fast_score_lds(s_img + (qy + 3) * PW + (qx + 3), PW, threshold);
        s_sc[i] = (uint8_t)sc;
    }
    __syncthreads();
    // 4 output pixels per thread: rows ty, ty + 4, ty + 8, ty + 12 of column tx
    const int tx = tid & 63, ty = tid >> 6;
    const int lane = tid & 63, wave = tid >> 6;
    unsigned long long masks[4];
    int scs[4];
    int total_wave = 0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int oy = ty + 4 * r;
        const int gx = x0 + tx, gy = y0 + oy;
        const uint8_t *c = s_sc + (oy + 1) * SW + (tx + 1);
        const int sc = c[0];
        bool is_max = sc != 0 && gx < W - edge && gy < H - edge;
        if (is_max) {
#pragma unroll
            for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
                for (int dx = -1; dx <= 1; ++dx)
                    if ((dx || dy) && c[dy * SW + dx] >= sc)
                        is_max = false;
        }
        scs[r] = is_max ? sc : 0;
        masks[r] = __ballot(is_max);
        total_wave += __popcll(masks[r]);
    }
    if (lane == 0)
        wave_off[wave] = total_wave;
    __syncthreads();
    const size_t slot = (size_t)b * n_levels + level;
    if (tid == 0) {
        int total = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const int c = wave_off[w];
            wave_off[w] = total;
            total += c;
        }
        tile_base = total ? atomicAdd(&count[slot], total) : 0;
    }
    __syncthreads();
    int base = tile_base + wave_off[wave];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        if (scs[r]) {
            const int idx = base + __popcll(masks[r] & ((1ull << lane) - 1ull));
            if (idx < cap)
                keys[slot * cap + idx] = rank_key((uint32_t)scs[r], y0 + ty + 4 * r, x0 + tx);
        }
        base += __popcll(masks[r]);
    }
}

__device__ __forceinline__ float harris_at(const uint8_t *img, int W, int x0, int y0)
{
    int a = 0, b = 0, c = 0;
    for (int dy = -3; dy <= 3; ++dy)
        for (int dx = -3; dx <= 3; ++dx) {
            const uint8_t *p = img + (y0 + dy) * W + (x0 + dx);
            const int Ix = ((int)p[1] - (int)p[-1]) * 2 + ((int)p[-W + 1] - (int)p[-W - 1]) + ((int)p[W + 1] - (int)p[W - 1]);
            const int Iy = ((int)p[W] - (int)p[-W]) * 2 + ((int)p[W - 1] - (int)p[-W - 1]) + ((int)p[W + 1] - (int)p[-W + 1]);
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

__device__ __forceinline__ uint32_t float_ordered(float f)
{
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// ascending bitonic sort of n (power of two) keys in LDS by the whole workgroup
__device__ void bitonic_sort(uint64_t *k, int n)
{
    for (int size = 2; size <= n; size <<= 1)
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            __syncthreads();
            for (int i = threadIdx.x; i < (n >> 1); i += blockDim.x) {
                const int lo = 2 * i - (i & (stride - 1)), hi = lo + stride;
                const bool up = (lo & size) == 0;
                const uint64_t a = k[lo], b = k[hi];
                if ((a > b) == up) {
                    k[lo] = b;
                    k[hi] = a;
                }
            }
        }
    __syncthreads();
}

using Sel = OrbSel;

// grid (n_images, n_levels): all levels of all images in one launch (one launch per level left 3/4 of the CUs idle)
__global__ __launch_bounds__(1024) void select_kernel(OrbDev d)
{
    extern __shared__ uint64_t keys[];
    const int b = blockIdx.x, level = blockIdx.y;
    const OrbLevel &L = d.level[level];
    if (L.w <= 2 * d.edge || L.h <= 2 * d.edge || L.n_keep < 1)
        return;   // sel_count stays 0
    const size_t slot = (size_t)b * d.n_levels + level;
    const int found = d.cand_count[slot];
    const int c = min(found, d.cand_cap);
    int n2 = 1;
    while (n2 < c)
        n2 <<= 1;
    const uint64_t *src = d.cand_keys + slot * d.cand_cap;
    for (int i = threadIdx.x; i < n2; i += blockDim.x)
        keys[i] = i < c ? src[i] : ~0ull;
    bitonic_sort(keys, n2);
    // retainBest(2 n_l) by FAST score, then Harris on the survivors
    const int keep1 = min(c, 2 * L.n_keep);
    const uint8_t *img = d.pyr + L.offset * d.n_images + (size_t)b * L.w * L.h;
    for (int i = threadIdx.x; i < n2; i += blockDim.x) {
        uint64_t k = ~0ull;
        if (i < keep1) {
            const int x = (int)(keys[i] & 0xffffu), y = (int)((keys[i] >> 16) & 0xffffu);
            k = rank_key(float_ordered(harris_at(img, L.w, x, y)), y, x);
        }
        keys[i] = k;   // slot i is read and written by this thread only
    }
    int m2 = 1;
    while (m2 < keep1)
        m2 <<= 1;
    bitonic_sort(keys, max(m2, 1));
    const int keep2 = min(keep1, L.n_keep);
    Sel *sel = d.sel + slot * d.nfeatures;
    for (int i = threadIdx.x; i < keep2; i += blockDim.x) {
        const uint64_t k = keys[i];
        const uint32_t ord = 0xffffffffu - (uint32_t)(k >> 32);
        const uint32_t bits = (ord & 0x80000000u) ? (ord & 0x7fffffffu) : ~ord;
        sel[i].x = (int)(k & 0xffffu);
        sel[i].y = (int)((k >> 16) & 0xffffu);
        sel[i].harris = __uint_as_float(bits);
    }
    if (threadIdx.x == 0) {
        d.sel_count[slot] = keep2;
        if (found > d.cand_cap)
            atomicOr(d.overflow, 1);
    }
}

// 7x7 sigma-2 blur of one 64 x 16 tile: rows then columns through LDS (Q8 kernel, BORDER_REFLECT_101, u16 row sums).
// Every work item produces 4 adjacent pixels from dword LDS reads (3 per row item, 14 per column item); byte-wide LDS
// reads made the first LDS version slower than two global passes.
__global__ __launch_bounds__(256) void blur_kernel(const uint8_t *img, int W, int H, uint8_t *out)
{
    constexpr int PW = kTileW + 8, PH = kTileH + 6;   // 72-byte pitch: output column 4k starts at a dword of the patch
    __shared__ __attribute__((aligned(16))) uint8_t s_img[PH * PW];
    __shared__ __attribute__((aligned(16))) uint16_t s_row[PH * kTileW];
    const int tid = threadIdx.x;
    img += (size_t)blockIdx.z * W * H;
    out += (size_t)blockIdx.z * W * H;
    const int x0 = blockIdx.x * kTileW, y0 = blockIdx.y * kTileH;
    for (int i = tid; i < PH * PW; i += 256) {
        const int py = i / PW, px = i - py * PW;
        int gx = x0 - 3 + px, gy = y0 - 3 + py;
        if (gx < 0) gx = -gx;
        if (gx >= W) gx = 2 * (W - 1) - gx;
        if (gy < 0) gy = -gy;
        if (gy >= H) gy = 2 * (H - 1) - gy;
        gx = min(max(gx, 0), W - 1);   // only for tile pixels beyond the image (never written)
        gy = min(max(gy, 0), H - 1);
        s_img[i] = img[gy * W + gx];
    }
    __syncthreads();
    for (int i = tid; i < PH * (kTileW / 4); i += 256) {   // row pass: 4 outputs from 10 patch bytes
        const int py = i / (kTileW / 4), q = i - py * (kTileW / 4);
        const uint32_t *w = reinterpret_cast<const uint32_t *>(s_img + py * PW + 4 * q);
        const uint32_t w0 = w[0], w1 = w[1], w2 = w[2];
        int p[12];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            p[k] = (w0 >> (8 * k)) & 0xff;
            p[4 + k] = (w1 >> (8 * k)) & 0xff;
            p[8 + k] = (w2 >> (8 * k)) & 0xff;
        }
        uint32_t r[4];
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            int sum = 0;
#pragma unroll
            for (int k = 0; k < 7; ++k)
                sum += kGauss[k] * p[o + k];
            r[o] = (uint32_t)sum;
        }
        uint2 pack;
        pack.x = r[0] | (r[1] << 16);
        pack.y = r[2] | (r[3] << 16);
        *reinterpret_cast<uint2 *>(s_row + py * kTileW + 4 * q) = pack;
    }
    __syncthreads();
    {   // column pass: thread = (row ty, column group q), 4 outputs
        const int q = tid & 15, oy = tid >> 4;
        uint32_t acc[4] = {0, 0, 0, 0};
#pragma unroll
        for (int k = 0; k < 7; ++k) {
            const uint2 v = *reinterpret_cast<const uint2 *>(s_row + (oy + k) * kTileW + 4 * q);
            const uint32_t g = (uint32_t)kGauss[k];
            acc[0] += g * (v.x & 0xffffu);
            acc[1] += g * (v.x >> 16);
            acc[2] += g * (v.y & 0xffffu);
            acc[3] += g * (v.y >> 16);
        }
        const int gx = x0 + 4 * q, gy = y0 + oy;
        if (gy < H) {
#pragma unroll
            for (int o = 0; o < 4; ++o)
                if (gx + o < W)
                    out[gy * W + gx + o] = (uint8_t)((acc[o] + 32768u) >> 16);
        }
    }
}

__device__ __forceinline__ float fast_atan2_deg(float y, float x)
{
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

// grid (n_levels, n_images, kDescSplit), 256 threads = 4 wavefronts, two keypoints per wavefront at a time.  The z
// dimension splits the keypoints of one (level, image) over 16 workgroups (more waves in flight: 0.45 -> 0.39 ms per 64
// frames).  Tried and dropped: lane = column with row-coalesced reads for the moments plus the blurred patch staged in LDS
// -- bit-identical but 0.53 ms: the kernel is bound by the number of byte-load and index instructions, not by the
// scattered addresses.
constexpr int kDescSplit = 16;
__global__ __launch_bounds__(256) void describe_kernel(OrbDev d)
{
    const int level = blockIdx.x, b = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const OrbLevel &L = d.level[level];
    const size_t slot0 = (size_t)b * d.n_levels;
    int offset = 0;
    for (int l = 0; l < level; ++l)
        offset += d.sel_count[slot0 + l];
    const int n = d.sel_count[slot0 + level];
    if (level == d.n_levels - 1 && threadIdx.x == 0 && blockIdx.z == 0)
        d.n_kp[b] = offset + n;
    const int W = L.w;
    const uint8_t *img = d.pyr + L.offset * d.n_images + (size_t)b * L.w * L.h;
    const uint8_t *blr = d.blur + L.offset * d.n_images + (size_t)b * L.w * L.h;
    const Sel *sel = d.sel + (slot0 + level) * d.nfeatures;
    const float fs = L.scale;
    // two keypoints per wavefront: lanes 0-31 take keypoint i0, lanes 32-63 keypoint i0 + 1
    const int half = lane >> 5, hl = lane & 31;
    for (int i0 = 8 * blockIdx.z + 2 * wave; i0 < n; i0 += 8 * kDescSplit) {
        const bool active = i0 + half < n;
        const int i = active ? i0 + half : n - 1;
        const int x0 = sel[i].x, y0 = sel[i].y;
        // intensity-centroid moments over the radius-15 disc: rows +-v, lanes 0-15 take u < 0, lanes 16-31 take u >= 0
        int m10 = 0, m01 = 0;
        {
            const uint8_t *c = img + y0 * W + x0;
            const int v = hl & 15, dmax = kUmax[v];
            const int u0 = hl < 16 ? -dmax : 0, u1 = hl < 16 ? -1 : dmax;
            if (v == 0) {
                for (int u = u0; u <= u1; ++u)
                    m10 += u * c[u];
            } else {
                int vs = 0;
                for (int u = u0; u <= u1; ++u) {
                    const int vp = c[u + v * W], vm = c[u - v * W];
                    vs += vp - vm;
                    m10 += u * (vp + vm);
                }
                m01 = v * vs;
            }
        }
#pragma unroll
        for (int s = 1; s < 32; s <<= 1) {   // integer sums inside each half: any order gives the same result
            m10 += __shfl_xor(m10, s, 64);
            m01 += __shfl_xor(m01, s, 64);
        }
        const float f10 = (float)m10, f01 = (float)m01;
        const float h2 = f10 * f10 + f01 * f01;
        float ca = 1.0f, sa = 0.0f;
        if (h2 > 0.0f) {
            const float hh = sqrtf(h2);
            ca = f10 / hh;
            sa = f01 / hh;
        }
        if (!active)
            continue;
        const size_t o = (size_t)b * d.nfeatures + offset + i;
        {
            const uint8_t *ctr = blr + y0 * W + x0;
            unsigned v = 0;
#pragma unroll
            for (int bit = 0; bit < 8; ++bit) {
                const int8_t *q = d.pattern + 4 * (8 * hl + bit);
                const float x1 = (float)q[0], y1 = (float)q[1], x2 = (float)q[2], y2 = (float)q[3];
                const int ix1 = (int)rintf(x1 * ca - y1 * sa), iy1 = (int)rintf(x1 * sa + y1 * ca);
                const int ix2 = (int)rintf(x2 * ca - y2 * sa), iy2 = (int)rintf(x2 * sa + y2 * ca);
                v |= (unsigned)(ctr[iy1 * W + ix1] < ctr[iy2 * W + ix2]) << bit;
            }
            d.desc[o * 32 + hl] = (uint8_t)v;
        }
        if (hl == 0) {
            mvs_keypoint k;
            k.x = (float)x0 * fs;
            k.y = (float)y0 * fs;
            k.size = 31.0f * fs;
            k.angle = fast_atan2_deg(f01, f10);
            k.response = sel[i].harris;
            k.octave = level;
            k.class_id = -1;
            d.kp[o] = k;
            if (d.kp_xy) {
                d.kp_xy[2 * o] = k.x;
                d.kp_xy[2 * o + 1] = k.y;
            }
            if (d.kp_oct)
                d.kp_oct[o] = (uint8_t)level;
        }
    }
}

__global__ void orb_clear_kernel(int32_t *count, int n, int32_t *overflow)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n)
        count[i] = 0;
    if (i == 0)
        *overflow = 0;
}

}  // namespace

size_t orb_sel_bytes() { return sizeof(Sel); }

hipError_t orb_prepare(int cand_cap)
{
    return hipFuncSetAttribute((const void *)select_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)((size_t)cand_cap * sizeof(uint64_t)));
}

void launch_orb(const OrbDev &d, hipStream_t stream)
{
    const int B = d.n_images;
    if (B <= 0)
        return;
    const int slots = B * d.n_levels;
    hipLaunchKernelGGL(orb_clear_kernel, dim3((slots + 255) / 256), dim3(256), 0, stream, d.cand_count, slots, d.overflow);
    hipLaunchKernelGGL(orb_clear_kernel, dim3((slots + 255) / 256), dim3(256), 0, stream, d.sel_count, slots, d.overflow);
    const dim3 blk(32, 8);
    for (int l = 0; l < d.n_levels; ++l) {
        const OrbLevel &L = d.level[l];
        if (L.w < 1 || L.h < 1)
            break;
        uint8_t *img = d.pyr + L.offset * B;
        if (l > 0) {
            const OrbLevel &Pv = d.level[l - 1];
            const dim3 grid((L.w + 31) / 32, (L.h + 7) / 8, B);
            hipLaunchKernelGGL(resize_kernel, grid, blk, 0, stream, d.pyr + Pv.offset * B, Pv.w, Pv.h, img, L.w, L.h,
                               d.resize_tab + L.tab_offset, d.resize_tab + L.tab_offset + L.w);
        }
        if (L.w <= 2 * d.edge || L.h <= 2 * d.edge || L.n_keep < 1)
            continue;
        const dim3 gin((L.w - 2 * d.edge + kTileW - 1) / kTileW, (L.h - 2 * d.edge + kTileH - 1) / kTileH, B);
        hipLaunchKernelGGL(fast_nms_kernel, gin, dim3(256), 0, stream, img, L.w, L.h, d.fast_threshold, d.edge, d.cand_keys,
                           d.cand_count, d.cand_cap, l, d.n_levels);
        const dim3 gb((L.w + kTileW - 1) / kTileW, (L.h + kTileH - 1) / kTileH, B);
        hipLaunchKernelGGL(blur_kernel, gb, dim3(256), 0, stream, img, L.w, L.h, d.blur + L.offset * B);
    }
    hipLaunchKernelGGL(select_kernel, dim3(B, d.n_levels), dim3(1024), (size_t)d.cand_cap * sizeof(uint64_t), stream,
                       d);   // LDS limit raised in orb_prepare()
    hipLaunchKernelGGL(describe_kernel, dim3(d.n_levels, B, kDescSplit), dim3(256), 0, stream, d);
}

}  // namespace mvs
