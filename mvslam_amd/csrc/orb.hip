// orb.hip -- VisualFeature::extract (vision/visual-feature.cpp:12-17,40-49 = cv::ORB::create(500) detect + compute;
// SURVEY section 8 row f3) for a batch of equally sized grayscale images.
//
// cv::ORB is OpenCV-internal and its learned sampling pattern is not in the reference tree, so this is ORB's PUBLISHED
// pipeline with the reference's parameters and the build's own, fully specified choices where OpenCV's are out of
// reach (the CPU oracle oracle/mvs_orb_oracle.c follows the same specification bit for bit):
//   resize_kernel    level l from level l-1: pixel-centre bilinear in integer arithmetic (11-bit weights from exact
//                    rationals, (sum + 2^21) >> 22)                                  four adjacent pixels per thread
//   fast_nms_kernel  ONE launch over the tiles of every level.  FAST-9/16 score (the largest threshold at which the pixel
//                    is still a corner) and the strict 3x3 maximum of one 64x16 tile through LDS: a cheap pre-test (two
//                    adjacent compass points both brighter or both darker) runs for every pixel, the pixels that pass
//                    it are COMPACTED into an LDS list and only those get the full 16-pixel arc minimum, on dense lanes
//                    (round 5; before, one passing lane made its whole wavefront pay the full score) -> rank key
//                    (score desc, y, x), one atomic per tile; the ORDER of the list does not matter, the select kernel
//                    ranks it
//   select_kernel    one launch, one workgroup per (image, level): the 2 n_l best FAST scores by a RADIX SELECT over the
//                    key bytes (five 256-bin histogram passes give the exact cut-off key; round 5 -- before, a bitonic
//                    sort of all <= 16384 candidates in 128 KB of LDS), Harris response (7x7, k = 0.04) of those from a
//                    9 x 12 register window, bitonic sort of the <= 2 n_l keys (response desc, y, x), keep n_l
//                    (cv::ORB's retainBest)
//   blur_kernel      ONE launch over every level: 7x7 sigma-2 Gaussian as the Q8 kernel {18,34,49,54,49,34,18},
//                    BORDER_REFLECT_101, rows then columns of one tile through LDS
//   describe_kernel  a wavefront per keypoint (round 5): the 31 x 32 image window and the 37 x 40 blurred window arrive as
//                    ten independent unaligned dword loads per lane (the next keypoint's are in flight while this one is
//                    computed) and are parked in LDS; intensity-centroid moments over the radius-15 disc = lane per row,
//                    v_dot4_u32_u8 against the row's mask / (u + 16) weights, butterfly sum; cos / sin = moments / hypot
//                    (no trigonometry); 256 steered BRIEF tests on the blurred window (four per lane, bytes from LDS);
//                    cv::KeyPoint record
// Everything is integer or single IEEE float operations in a stated order, so GPU and oracle agree bit for bit.
#include "kernels.hpp"

#include "device_math.hpp"

namespace mvs {

namespace {

__constant__ int kFastDx[16] = {0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1};
__constant__ int kFastDy[16] = {3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1, 0, 1, 2, 3};
__constant__ int kUmax[16] = {15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3};
__constant__ int kGauss[7] = {18, 34, 49, 54, 49, 34, 18};

// four bytes at ANY address as one load (gfx950 global memory takes unaligned dwords; hipcc emits global_load_dword)
__device__ __forceinline__ uint32_t load_u32_unaligned(const uint8_t *p)
{
    uint32_t v;
    __builtin_memcpy(&v, p, 4);
    return v;
}

// tab: per destination column / row {source index, 11-bit weight of the next sample}, computed once per level on the
// host from the exact rationals ((2 d + 1) s_src - s_dst) / (2 s_dst) -- 64-bit divisions per pixel were 3/4 of this kernel.
// Round 5: FOUR adjacent destination pixels per thread.  Their source columns lie within eight bytes of the first one's
// (the step between levels is 1.2; a thread whose span is wider falls back to byte loads), so the two source rows arrive
// as two unaligned 8-byte windows instead of sixteen byte loads, and the result leaves as one dword.
__global__ void resize_kernel(const uint8_t *src, int sw, int sh, uint8_t *dst, int dw, int dh, const int2 *xtab,
                              const int2 *ytab)
{
    const int dx = 4 * (blockIdx.x * blockDim.x + threadIdx.x), dy = blockIdx.y * blockDim.y + threadIdx.y;
    if (dx >= dw || dy >= dh)
        return;
    src += (size_t)blockIdx.z * sw * sh;
    dst += (size_t)blockIdx.z * dw * dh;
    const int2 ty = ytab[dy];
    const int sy = ty.x, wy = ty.y, sy1 = min(sy + 1, sh - 1);
    const uint8_t *r0 = src + (size_t)sy * sw, *r1 = src + (size_t)sy1 * sw;
    int2 tx[4];
#pragma unroll
    for (int k = 0; k < 4; ++k)
        tx[k] = xtab[min(dx + k, dw - 1)];
    const int s0 = tx[0].x;
    const bool window = tx[3].x + 1 - s0 <= 7;   // (source columns ascend with the destination column)
    uint64_t w0 = 0, w1 = 0;
    if (window) {
        w0 = (uint64_t)load_u32_unaligned(r0 + s0) | ((uint64_t)load_u32_unaligned(r0 + s0 + 4) << 32);
        w1 = (uint64_t)load_u32_unaligned(r1 + s0) | ((uint64_t)load_u32_unaligned(r1 + s0 + 4) << 32);
    }
    uint32_t o8[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int sx = tx[k].x, wx = tx[k].y, sx1 = min(sx + 1, sw - 1);
        uint32_t p00, p01, p10, p11;
        if (window) {
            const int a = 8 * (sx - s0), b = 8 * (sx1 - s0);
            p00 = (uint32_t)(w0 >> a) & 0xffu, p01 = (uint32_t)(w0 >> b) & 0xffu;
            p10 = (uint32_t)(w1 >> a) & 0xffu, p11 = (uint32_t)(w1 >> b) & 0xffu;
        } else {
            p00 = r0[sx], p01 = r0[sx1], p10 = r1[sx], p11 = r1[sx1];
        }
        const uint32_t v = p00 * (uint32_t)((2048 - wx) * (2048 - wy)) + p01 * (uint32_t)(wx * (2048 - wy)) +
                           p10 * (uint32_t)((2048 - wx) * wy) + p11 * (uint32_t)(wx * wy);
        o8[k] = (v + (1u << 21)) >> 22;
    }
    uint8_t *o = dst + (size_t)dy * dw + dx;
    if (dx + 3 < dw) {
        const uint32_t v = o8[0] | (o8[1] << 8) | (o8[2] << 16) | (o8[3] << 24);
        __builtin_memcpy(o, &v, 4);
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (dx + k < dw)
                o[k] = (uint8_t)o8[k];
    }
}

__device__ __forceinline__ uint64_t rank_key(uint32_t value_desc, int y, int x)
{
    return ((uint64_t)(0xffffffffu - value_desc) << 32) | ((uint64_t)(uint32_t)y << 16) | (uint32_t)x;
}

// the tiles of every level flattened into one grid (by-value kernel argument): level l owns blocks [start[l], start[l+1])
// of tx[l] tiles per row
struct OrbGrid {
    int fast_start[kOrbMaxLevels + 1], fast_tx[kOrbMaxLevels];
    int blur_start[kOrbMaxLevels + 1], blur_tx[kOrbMaxLevels];
};

// An arc of 9 contiguous ring pixels contains two ADJACENT compass points (ring positions 0, 4, 8, 12), so a corner at
// threshold t has two adjacent compass points both brighter than centre + t or both darker than centre - t.  (Round 5; the
// unsigned form of rounds 1-4 -- one pixel of each antipodal pair differs by more than t -- let twice as many pixels
// through on a textured image.)
__device__ __forceinline__ bool fast_maybe(const uint8_t *c, int P, int threshold)
{
    const int ctr = c[0];
    const int d0 = (int)c[3 * P] - ctr, d4 = (int)c[3] - ctr, d8 = (int)c[-3 * P] - ctr, d12 = (int)c[-3] - ctr;
    const int brighter = max(max(min(d0, d4), min(d4, d8)), max(min(d8, d12), min(d12, d0)));
    const int darker = min(min(max(d0, d4), max(d4, d8)), min(max(d8, d12), max(d12, d0)));
    return brighter > threshold || darker < -threshold;
}

// FAST-9/16 score of the pixel at LDS position c (row pitch P): the largest threshold at which it is still a corner,
// 0 if below `threshold` (a pixel that fails fast_maybe scores below the threshold: its score is never computed).
// score + 1 = max over the 16 arcs of 9 contiguous ring pixels of max(min(ring) - centre, centre - max(ring)), and
// centre - max(ring) = min(255 - ring) - (255 - centre): both arc minima run through ONE network of packed 16-bit minima
// over (ring, 255 - ring) pairs -- one v_mad_i32_i24 builds a pair (ring * -65535 + (255 << 16)), windows of 2, 4, 8 + 1 by
// doubling (round 5: ~100 vector instructions per pixel instead of ~165 on separate min / max chains).
// Measured and not kept: TWO adjacent pixels per lane in the two halves (ring values and complements through two
// networks, 215 instructions per pair) for EVERY pixel, without pre-test and list -- bit-identical, 0.26 against 0.24 ms per
// 64 frames on the textured bench frames (a third of whose pixels pass the pre-test; real frames pass far fewer).
// Also measured and not kept: the non-maximum suppression over the list's entries with an LDS survivor list instead of
// four output pixels per thread (0.2436 against 0.2439 ms: the suppression was never the cost).
typedef uint16_t u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ int fast_score_lds(const uint8_t *c, int P, int threshold)
{
    const int ctr = c[0];
    u16x2 d[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int ring = c[kFastDy[k] * P + kFastDx[k]];
        d[k] = __builtin_bit_cast(u16x2, ring * -65535 + (255 << 16));   // (ring, 255 - ring)
    }
    u16x2 m2[16], m4[16];
#pragma unroll
    for (int k = 0; k < 16; ++k)
        m2[k] = __builtin_elementwise_min(d[k], d[(k + 1) & 15]);
#pragma unroll
    for (int k = 0; k < 16; ++k)
        m4[k] = __builtin_elementwise_min(m2[k], m2[(k + 2) & 15]);
    u16x2 best = {0, 0};
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const u16x2 m9 = __builtin_elementwise_min(__builtin_elementwise_min(m4[k], m4[(k + 4) & 15]), d[(k + 8) & 15]);
        best = __builtin_elementwise_max(best, m9);
    }
    const int sc = max((int)best.x - ctr, (int)best.y - (255 - ctr)) - 1;
    return sc >= threshold ? sc : 0;
}

// Block order of the tiled kernels (round 5).  Workgroups go round-robin to the eight XCDs, each with its own L2.  The first
// 8 * floor(n_images / 8) images are owned by XCDs: XCD x = block mod 8 takes the images b = x (mod 8) and receives an image's
// blocks one after the other (what they share -- halo rows, 128-byte lines, a keypoint's windows -- meets in one L2).  The
// remaining n_images mod 8 images (all of them when a call brings fewer than eight -- one frame per call is the reference's own
// pattern, vision/visual-feature.cpp:40) are spread over ALL XCDs block by block: owning them would leave most of the chip idle.
// per_image: blocks one image needs.  Returns the image; seq = the block's index among that image's blocks.
__device__ __forceinline__ int orb_block_image(int bid, int n_images, int per_image, int &seq)
{
    const int full = n_images & ~7, owned = full * per_image;
    if (bid < owned) {
        const int s = bid >> 3;
        seq = s % per_image;
        return (s / per_image) * 8 + (bid & 7);
    }
    const int rest = n_images - full, r = bid - owned;   // (the grid has exactly n_images * per_image blocks: rest > 0 here)
    seq = r / rest;
    return full + r % rest;
}

// FAST + non-maximum suppression of one 64 x 16 tile through LDS: the image patch (tile + 4) is read from HBM once (as
// unaligned dwords: columns past the row's end belong to pixels whose score is never taken), the scores of (tile + 1) never
// leave the CU, and the surviving corners are appended with ONE atomic per tile (all corners of a level append to the same
// counter and same-address atomics serialise: one per wavefront was 46 % of the extraction).  Keys carry (score desc, y,
// x): the order of the list does not matter, select_kernel ranks it.
// Round 5: scoring in two passes.  The antipodal test is 4 LDS bytes and a dozen instructions, the full score 16 bytes and
// ~200; on a textured image a tenth to a third of the pixels pass the test, spread so that nearly every wavefront held one
// and paid the full score for all 64 lanes.  Now the passing pixels are compacted (ballot + one LDS atomic per wavefront)
// and the full score runs over the list.
constexpr int kTileW = 64, kTileH = 16, kBlurH = 32;
__global__ __launch_bounds__(256) void fast_nms_kernel(OrbDev d, OrbGrid g, int tile_first, int n_tiles)
{
    constexpr int PW = kTileW + 8, PH = kTileH + 8;      // image patch: +-4 (NMS 1 + circle 3)
    constexpr int SW = kTileW + 2, SH = kTileH + 2;      // score patch: +-1
    __shared__ __attribute__((aligned(16))) uint8_t s_img[PH * PW];
    __shared__ uint8_t s_sc[SH * SW];
    __shared__ uint16_t s_list[SH * SW];
    __shared__ int s_n;
    __shared__ int wave_off[4];
    __shared__ int tile_base;
    // (block order: orb_block_image; a launch covers tiles [tile_first, tile_first + n_tiles) of the flattened list: launch_orb)
    int seq;
    const int b = orb_block_image(blockIdx.x, d.n_images, n_tiles, seq), tile = tile_first + seq, tid = threadIdx.x;
    int level = 0;
    while (level + 1 < d.n_levels && tile >= g.fast_start[level + 1])
        ++level;
    const OrbLevel &L = d.level[level];
    const int W = L.w, H = L.h, edge = d.edge, threshold = d.fast_threshold;
    const int t = tile - g.fast_start[level];
    const int tyb = t / g.fast_tx[level], txb = t - tyb * g.fast_tx[level];
    const uint8_t *img = d.pyr + L.offset * d.n_images + (size_t)b * W * H;
    const int x0 = edge + txb * kTileW, y0 = edge + tyb * kTileH;   // first output pixel of the tile
    if (tid == 0)
        s_n = 0;
    for (int q = tid; q < PH * (PW / 4); q += 256) {
        const int py = q / (PW / 4), c = q - py * (PW / 4);
        const int gy = min(max(y0 - 4 + py, 0), H - 1);           // clamped rows are never used
        reinterpret_cast<uint32_t *>(s_img)[q] = load_u32_unaligned(img + (size_t)gy * W + (x0 - 4 + 4 * c));
    }
    __syncthreads();
    const int lane = tid & 63, wave = tid >> 6;
    for (int i0 = 0; i0 < SH * SW; i0 += 256) {
        const int i = i0 + tid;
        bool f = false;
        if (i < SH * SW) {
            const int qy = i / SW, qx = i - qy * SW;
            const int gx = x0 - 1 + qx, gy = y0 - 1 + qy;
            s_sc[i] = 0;
            if (gx >= 3 && gy >= 3 && gx < W - 3 && gy < H - 3)
                f = fast_maybe(s_img + (qy + 3) * PW + (qx + 3), PW, threshold);
        }
        const unsigned long long m = __ballot(f);
        if (m) {   // wavefront-uniform
            int base = 0;
            if (lane == 0)
                base = atomicAdd(&s_n, __popcll(m));
            base = __builtin_amdgcn_readfirstlane(base);
            if (f)
                s_list[base + __popcll(m & ((1ull << lane) - 1ull))] = (uint16_t)i;
        }
    }
    __syncthreads();
    const int n_list = s_n;
    for (int j = tid; j < n_list; j += 256) {
        const int i = s_list[j];
        const int qy = i / SW, qx = i - qy * SW;
        s_sc[i] = (uint8_t)fast_score_lds(s_img + (qy + 3) * PW + (qx + 3), PW, threshold);
    }
    __syncthreads();
    // 4 output pixels per thread: rows ty, ty + 4, ty + 8, ty + 12 of column tx
    const int tx = tid & 63, ty = tid >> 6;
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
    const size_t slot = (size_t)b * d.n_levels + level;
    if (tid == 0) {
        int total = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const int c = wave_off[w];
            wave_off[w] = total;
            total += c;
        }
        tile_base = total ? atomicAdd(&d.cand_count[slot], total) : 0;
    }
    __syncthreads();
    int base = tile_base + wave_off[wave];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        if (scs[r]) {
            const int idx = base + __popcll(masks[r] & ((1ull << lane) - 1ull));
            if (idx < L.cand_cap)
                d.cand_keys[(size_t)b * d.cand_stride + L.cand_off + idx] = rank_key((uint32_t)scs[r], y0 + ty + 4 * r, x0 + tx);
        }
        base += __popcll(masks[r]);
    }
}

// Harris response (7x7 window, Sobel 3x3 gradients, k = 0.04) of the pixel (x0, y0).  Round 5: rows y0-4 .. y0+4,
// columns x0-4 .. x0+7 arrive as 27 independent unaligned dword loads and everything else happens in registers -- the
// integer sums a, b, c are the same whatever the order (before: 49 x 8 byte loads in a dependent loop, ~1 us each from L2)
__device__ __forceinline__ float harris_at(const uint8_t *img, int W, int x0, int y0)
{
    uint32_t r[9][3];
#pragma unroll
    for (int j = 0; j < 9; ++j) {
        const uint8_t *p = img + (y0 - 4 + j) * W + (x0 - 4);
#pragma unroll
        for (int k = 0; k < 3; ++k)
            r[j][k] = load_u32_unaligned(p + 4 * k);
    }
    // per row j and window column x (0 .. 6 <-> x0-3 .. x0+3; the pixel itself is byte x + 1 of the row):
    //   dxr = p[+1] - p[-1],  sxr = p[-1] + 2 p[0] + p[+1];   Ix = 2 dxr[j] + dxr[j-1] + dxr[j+1],  Iy = sxr[j+1] - sxr[j-1]
    int dxr[9][7], sxr[9][7];
#pragma unroll
    for (int j = 0; j < 9; ++j) {
        int px[9];
#pragma unroll
        for (int k = 0; k < 9; ++k)
            px[k] = (int)((r[j][k >> 2] >> (8 * (k & 3))) & 0xffu);
#pragma unroll
        for (int x = 0; x < 7; ++x) {
            dxr[j][x] = px[x + 2] - px[x];
            sxr[j][x] = px[x] + 2 * px[x + 1] + px[x + 2];
        }
    }
    int a = 0, b = 0, c = 0;
#pragma unroll
    for (int j = 1; j < 8; ++j)
#pragma unroll
        for (int x = 0; x < 7; ++x) {
            const int Ix = 2 * dxr[j][x] + dxr[j - 1][x] + dxr[j + 1][x];
            const int Iy = sxr[j + 1][x] - sxr[j - 1][x];
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

constexpr int kSelTieCap = 1024;   // candidates that share the FAST score at the cut (select_kernel's fast path)
// grid (n_images, n_levels): all levels of all images in one launch (one launch per level left 3/4 of the CUs idle).
// Dynamic LDS: room for the next power of two above 2 max(n_l) keys (launch_orb).
// retainBest(2 n_l) by FAST score needs the SET of the 2 n_l smallest keys, not their order: a radix select over the five
// key bytes that vary (score, y, x; keys are unique) finds the exact cut-off key in five histogram passes over the
// candidates in global memory (L2-resident, coalesced), a sixth pass compacts the keys at or below it.
__global__ __launch_bounds__(1024) void select_kernel(OrbDev d)
{
    extern __shared__ uint64_t keys[];
    __shared__ int hist[256];
    __shared__ int s_bin, s_k, s_hbin, s_cnt, s_tcnt;
    __shared__ uint32_t s_tie[kSelTieCap];
    const int b = blockIdx.x, level = blockIdx.y;
    const OrbLevel &L = d.level[level];
    if (L.w <= 2 * d.edge || L.h <= 2 * d.edge || L.n_keep < 1)
        return;   // sel_count stays 0
    const int tid = threadIdx.x, lane = tid & 63;
    const size_t slot = (size_t)b * d.n_levels + level;
    const int found = d.cand_count[slot];
    const int c = min(found, L.cand_cap);
    const uint64_t *src = d.cand_keys + (size_t)b * d.cand_stride + L.cand_off;
    const int keep1 = min(c, 2 * L.n_keep);
    // one histogram pass over a key byte among the keys that match (prefix, mask): the bin that holds the k-th smallest of
    // them, k reduced to the rank inside that bin, the bin's population
    uint64_t prefix = 0, mask = 0;
    int k = keep1;
    auto radix_pass = [&](int byte) {
        if (tid < 256)
            hist[tid] = 0;
        __syncthreads();
        for (int i = tid; i < c; i += 1024) {
            const uint64_t key = src[i];
            if ((key & mask) == prefix)
                atomicAdd(&hist[(int)(key >> (8 * byte)) & 255], 1);
        }
        __syncthreads();
        if (tid < 64) {   // one wavefront: lane l owns bins 4 l .. 4 l + 3
            const int h0 = hist[4 * tid], h1 = hist[4 * tid + 1], h2 = hist[4 * tid + 2], h3 = hist[4 * tid + 3];
            const int sum = (h0 + h1) + (h2 + h3);
            int incl = sum;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int up = __shfl_up(incl, o, 64);
                if (lane >= o)
                    incl += up;
            }
            const int excl = incl - sum;
            if (excl < k && k <= incl) {   // exactly one lane: the matching keys number at least k
                int r = k - excl, bin = 4 * tid, hb = h0;
                if (r > h0) {
                    r -= h0, ++bin, hb = h1;
                    if (r > h1) {
                        r -= h1, ++bin, hb = h2;
                        if (r > h2)
                            r -= h2, ++bin, hb = h3;
                    }
                }
                s_bin = bin;
                s_k = r;
                s_hbin = hb;
            }
        }
        __syncthreads();
        prefix |= (uint64_t)(uint32_t)s_bin << (8 * byte);
        mask |= 0xffull << (8 * byte);
        k = s_k;
        // (the next pass's barrier behind the histogram reset orders these reads before the next writes of s_bin / s_k)
    };
    uint64_t kstar = ~0ull;   // the keep1-th smallest key (general path)
    bool filled = false;      // keys[0 .. keep1) already holds the selection (tie-class path)
    if (tid == 0) {
        s_cnt = 0;
        s_tcnt = 0;
    }
    if (keep1 < c) {          // (uniform)
        radix_pass(4);        // the score byte: every key in a lower bin is in, the bin s_bin contributes its k smallest (y, x)
        const int n_tie = s_hbin;
        if (n_tie <= kSelTieCap) {
            // The usual case: the score at the cut is shared by a few dozen corners.  ONE more pass sorts the candidates into
            // "better score" (straight into the selection) and "tie class" (an LDS list of their (y, x) words); the k smallest
            // of the tie class are found by counting, each key's rank among its class (keys are unique).  Before: four more
            // histogram passes over all candidates for the (y, x) bytes and a compaction pass.
            const uint32_t cut = (uint32_t)s_bin;
            for (int i0 = 0; i0 < c; i0 += 1024) {
                const int i = i0 + tid;
                const uint64_t key = i < c ? src[i] : ~0ull;
                const uint32_t digit = (uint32_t)(key >> 32) & 255u;
                const bool better = i < c && digit < cut, tie = i < c && digit == cut;
                const unsigned long long mb = __ballot(better), mt = __ballot(tie);
                if (mb) {
                    int base = 0;
                    if (lane == 0)
                        base = atomicAdd(&s_cnt, __popcll(mb));
                    base = __builtin_amdgcn_readfirstlane(base);
                    if (better)
                        keys[base + __popcll(mb & ((1ull << lane) - 1ull))] = key;
                }
                if (mt) {
                    int base = 0;
                    if (lane == 0)
                        base = atomicAdd(&s_tcnt, __popcll(mt));
                    base = __builtin_amdgcn_readfirstlane(base);
                    if (tie)
                        s_tie[base + __popcll(mt & ((1ull << lane) - 1ull))] = (uint32_t)key;
                }
            }
            __syncthreads();   // s_cnt == keep1 - k, s_tcnt == n_tie
            const int n_better = s_cnt;
            const uint64_t hi = (uint64_t)(0xffffff00u | cut) << 32;
            for (int t = tid; t < n_tie; t += 1024) {
                const uint32_t mine = s_tie[t];
                int rank = 0;
                for (int j = 0; j < n_tie; ++j)
                    rank += s_tie[j] < mine ? 1 : 0;
                if (rank < k)
                    keys[n_better + rank] = hi | mine;
            }
            filled = true;
        } else {
            for (int byte = 3; byte >= 0; --byte)
                radix_pass(byte);
            kstar = prefix | 0xffffff0000000000ull;   // the high word of a key is 0xffffffff - score, score <= 255
        }
    }
    __syncthreads();
    if (!filled) {            // (uniform) everything at or below the cut-off key -- or everything, when nothing is cut
        for (int i0 = 0; i0 < c; i0 += 1024) {
            const int i = i0 + tid;
            const uint64_t key = i < c ? src[i] : ~0ull;
            const bool take = i < c && key <= kstar;
            const unsigned long long m = __ballot(take);
            if (m) {
                int base = 0;
                if (lane == 0)
                    base = atomicAdd(&s_cnt, __popcll(m));
                base = __builtin_amdgcn_readfirstlane(base);
                if (take)
                    keys[base + __popcll(m & ((1ull << lane) - 1ull))] = key;
            }
        }
    }
    __syncthreads();   // keys[0 .. keep1) = the selection
    // Harris on the survivors, then the order by (response desc, y, x)
    int n2 = 1;
    while (n2 < keep1)
        n2 <<= 1;
    const uint8_t *img = d.pyr + L.offset * d.n_images + (size_t)b * L.w * L.h;
    for (int i = tid; i < n2; i += blockDim.x) {
        uint64_t k = ~0ull;
        if (i < keep1) {
            const int x = (int)(keys[i] & 0xffffu), y = (int)((keys[i] >> 16) & 0xffffu);
            k = rank_key(float_ordered(harris_at(img, L.w, x, y)), y, x);
        }
        keys[i] = k;   // slot i is read and written by this thread only
    }
    bitonic_sort(keys, n2);
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
        if (found > L.cand_cap)
            atomicOr(d.overflow, 1);
    }
}

// 7x7 sigma-2 blur of one 64 x 32 tile (round 5: 32 rows -- the row pass runs over the tile's rows + 6, at 16 rows that is
// 37 % more than the tile): rows then columns through LDS (Q8 kernel, BORDER_REFLECT_101, u16 row sums).
// Every work item produces 4 adjacent pixels from dword LDS reads (3 per row item, 14 per column item); byte-wide LDS
// reads made the first LDS version slower than two global passes.
__global__ __launch_bounds__(256) void blur_kernel(OrbDev d, OrbGrid g, int tile_first, int n_tiles)
{
    constexpr int PW = kTileW + 8, PH = kBlurH + 6;   // 72-byte pitch: output column 4k starts at a dword of the patch
    __shared__ __attribute__((aligned(16))) uint8_t s_img[PH * PW];
    __shared__ __attribute__((aligned(16))) uint16_t s_row[PH * kTileW];
    const int tid = threadIdx.x;
    int seq;   // (block order and tile range: fast_nms_kernel)
    const int b = orb_block_image(blockIdx.x, d.n_images, n_tiles, seq), tile = tile_first + seq;
    int level = 0;
    while (level + 1 < d.n_levels && tile >= g.blur_start[level + 1])
        ++level;
    const OrbLevel &L = d.level[level];
    const int W = L.w, H = L.h;
    const int t = tile - g.blur_start[level];
    const int tyb = t / g.blur_tx[level], txb = t - tyb * g.blur_tx[level];
    const uint8_t *img = d.pyr + L.offset * d.n_images + (size_t)b * W * H;
    uint8_t *out = d.blur + L.offset * d.n_images + (size_t)b * W * H;
    const int x0 = txb * kTileW, y0 = tyb * kBlurH;
    if (x0 >= 3 && x0 - 3 + PW <= W) {
        // no column of the patch leaves the row: rows reflect, columns are 18 unaligned dwords (round 5; the byte loop
        // below was half of this kernel's instructions)
        for (int q = tid; q < PH * (PW / 4); q += 256) {
            const int py = q / (PW / 4), c = q - py * (PW / 4);
            int gy = y0 - 3 + py;
            if (gy < 0) gy = -gy;
            if (gy >= H) gy = 2 * (H - 1) - gy;
            gy = min(max(gy, 0), H - 1);   // only for tile rows beyond the image (never written)
            reinterpret_cast<uint32_t *>(s_img)[q] = load_u32_unaligned(img + (size_t)gy * W + (x0 - 3 + 4 * c));
        }
    } else {
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
    }
    __syncthreads();
    for (int i = tid; i < PH * (kTileW / 4); i += 256) {   // row pass: 4 outputs from 10 patch bytes
        const int py = i / (kTileW / 4), q = i - py * (kTileW / 4);
        const uint32_t *w = reinterpret_cast<const uint32_t *>(s_img + py * PW + 4 * q);
        const uint32_t w0 = w[0], w1 = w[1], w2 = w[2];
        // output o = taps 0-3 against bytes o .. o+3 and taps 4-6 against bytes o+4 .. o+6: two v_dot4_u32_u8 on byte windows
        // cut out of (w0, w1, w2) by v_alignbyte (round 5: 14 instructions for the four outputs instead of 12 byte
        // extractions + 28 multiply-adds; sums of products of integers: the same numbers)
        constexpr uint32_t g03 = 18u | (34u << 8) | (49u << 16) | (54u << 24), g46 = 49u | (34u << 8) | (18u << 16);
        uint32_t r[4];
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            const uint32_t a = o ? __builtin_amdgcn_alignbyte(w1, w0, o) : w0;
            const uint32_t b = o ? __builtin_amdgcn_alignbyte(w2, w1, o) : w1;
            r[o] = __builtin_amdgcn_udot4(b, g46, __builtin_amdgcn_udot4(a, g03, 0u, false), false);
        }
        uint2 pack;
        pack.x = r[0] | (r[1] << 16);
        pack.y = r[2] | (r[3] << 16);
        *reinterpret_cast<uint2 *>(s_row + py * kTileW + 4 * q) = pack;
    }
    __syncthreads();
    for (int oy = tid >> 4; oy < kBlurH; oy += 16) {   // column pass: thread = (row oy, column group q), 4 outputs
        const int q = tid & 15;
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
            uint32_t o8[4];
#pragma unroll
            for (int o = 0; o < 4; ++o)
                o8[o] = (acc[o] + 32768u) >> 16;
            if (gx + 3 < W) {   // four pixels as one (unaligned) dword store
                const uint32_t v = o8[0] | (o8[1] << 8) | (o8[2] << 16) | (o8[3] << 24);
                __builtin_memcpy(out + (size_t)gy * W + gx, &v, 4);
            } else {
#pragma unroll
                for (int o = 0; o < 4; ++o)
                    if (gx + o < W)
                        out[gy * W + gx + o] = (uint8_t)o8[o];
            }
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

// LDS written by some lanes of a wavefront and read by others: the LDS queue of a wavefront is in order, what is needed is
// that the COMPILER keeps the order
__device__ __forceinline__ void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// grid split x n_levels x n_images (see the block order below), 256 threads = 4 wavefronts, ONE keypoint
// per wavefront at a time (round 5).
// History: half a wavefront per keypoint with byte loads straight from the level (moments: a loop of two dependent byte
// loads per lane and step; tests: 16 byte loads per lane) took 0.39 ms per 64 frames against 0.08 of issue time -- it waited
// for ~20 load round trips per keypoint.  Now a keypoint costs ONE round trip: its two windows are ten independent
// unaligned dword loads per lane, issued for the NEXT keypoint before this one is computed, parked in LDS, and everything
// else reads LDS or registers.
//   image window    rows -15 .. 15, columns -16 .. 15 (pitch 32): lane r < 31 owns row r - 15; its 32 bytes against the row's
//                   byte masks (|u| <= umax(|v|)) and (u + 16) weights through v_dot4_u32_u8 give S1 = sum p and
//                   sum (u + 16) p; m10 = sum over rows of (sum (u + 16) p - 16 S1), m01 = sum of v S1: integers, any order
//   blurred window  rows -18 .. 18, columns -18 .. 21 (pitch 40): a rotated test point has |coordinate| <= rint(13 sqrt 2
//                   (1 + 2^-22)) = 18 (the pattern is clipped to +-13 per axis; edge_threshold >= 19 keeps the window
//                   inside the level); lane l evaluates tests 4 l .. 4 l + 3, the nibbles meet through three shuffles and
//                   the descriptor leaves as eight dwords
constexpr int kDescSplit = 8, kDescSplitFew = 32;   // workgroups per (image, level): 8 in a batch; 32 when a call brings fewer than
                                                    // eight frames (one frame: 64 workgroups would leave three quarters of the CUs idle)
constexpr int kBR = 18, kBRows = 2 * kBR + 1, kBPitch = 40, kBWords = kBRows * kBPitch / 4;   // 370 dwords
constexpr int kIRows = 31, kIPitch = 32, kIWords = kIRows * kIPitch / 4;                      // 248 dwords
constexpr int kBLoads = (kBWords + 63) / 64, kILoads = (kIWords + 63) / 64;                   // 6 + 4 loads per lane
constexpr int kDescImgOff = 1536, kDescWaveLds = 2560;
static_assert(kBWords * 4 <= kDescImgOff && kDescImgOff + kIWords * 4 <= kDescWaveLds, "describe_kernel LDS layout");
__global__ __launch_bounds__(256) void describe_kernel(OrbDev d, int split)
{
    __shared__ __attribute__((aligned(16))) uint8_t s_patch[4][kDescWaveLds];
    // Block order (orb_block_image).  A keypoint's windows touch 68 rows of 128-byte lines for 40 + 32 useful bytes each: with
    // the blocks of one (image, level) spread over the XCDs every XCD pulled every line through the fabric (~1 GB per 64 frames
    // -- what the kernel's time was in rounds 2-4), and with the level as the fastest block index one XCD had all of level
    // 0, a fifth of the keypoints.  Now the blocks of an image walk the split, then the level, on the XCD that owns the image.
    int seq;
    int b = orb_block_image(blockIdx.x, d.n_images, split * d.n_levels, seq);
    int zsplit = seq % split, level = seq / split;
    if (d.flat_order) {   // diagnostics only: the level as the fastest index, then the image, then the split (rounds 2-4)
        const int bid = blockIdx.x;
        level = bid % d.n_levels;
        b = (bid / d.n_levels) % d.n_images;
        zsplit = bid / (d.n_levels * d.n_images);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const OrbLevel &L = d.level[level];
    const size_t slot0 = (size_t)b * d.n_levels;
    int offset = 0;
    for (int l = 0; l < level; ++l)
        offset += d.sel_count[slot0 + l];
    const int n = d.sel_count[slot0 + level];
    if (level == d.n_levels - 1 && threadIdx.x == 0 && zsplit == 0)
        d.n_kp[b] = offset + n;
    const int W = L.w;
    const uint8_t *img = d.pyr + L.offset * d.n_images + (size_t)b * L.w * L.h;
    const uint8_t *blr = d.blur + L.offset * d.n_images + (size_t)b * L.w * L.h;
    const Sel *sel = d.sel + (slot0 + level) * d.nfeatures;
    const float fs = L.scale;
    uint8_t *sb = s_patch[wave], *si = sb + kDescImgOff;
    // this lane's share of the two windows: offsets from the keypoint's pixel (the same for every keypoint of the level)
    int boff[kBLoads], ioff[kILoads];
#pragma unroll
    for (int t = 0; t < kBLoads; ++t) {
        const int q = min(lane + 64 * t, kBWords - 1), row = q / (kBPitch / 4), c = q - row * (kBPitch / 4);
        boff[t] = (row - kBR) * W + 4 * c - kBR;
    }
#pragma unroll
    for (int t = 0; t < kILoads; ++t) {
        const int q = min(lane + 64 * t, kIWords - 1), row = q / (kIPitch / 4), c = q - row * (kIPitch / 4);
        ioff[t] = (row - 15) * W + 4 * c - 16;
    }
    // moment weights of this lane's row (lanes 31 .. 63: all zero)
    const int mrow = min(lane, kIRows - 1), mv = mrow - 15, um = kUmax[abs(mv)];
    uint32_t wm[8], wu[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        wm[j] = 0;
        wu[j] = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int u = 4 * j + k - 16;
            if (abs(u) <= um && lane < kIRows) {
                wm[j] |= 1u << (8 * k);
                wu[j] |= (uint32_t)(u + 16) << (8 * k);
            }
        }
    }
    // this lane's four tests
    float pat[16];
    {
        const uint4 q = *reinterpret_cast<const uint4 *>(d.pattern + 16 * lane);
        const uint32_t w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int k = 0; k < 16; ++k)
            pat[k] = (float)(int8_t)(w[k >> 2] >> (8 * (k & 3)));
    }
    // three keypoints in flight per wavefront: this one (windows in LDS), the next one (its windows travelling into
    // registers) and the one after (its record travelling: the windows' addresses depend on it, and a wait for it at the
    // head of the window loads would expose one load round trip per keypoint)
    const int step = 4 * split;
    int i = 4 * zsplit + wave;
    uint32_t nb[kBLoads], ni[kILoads];
    Sel cur{}, nxt{};
    auto fetch = [&](const Sel &k) {
        const uint8_t *cb = blr + k.y * W + k.x, *ci = img + k.y * W + k.x;
#pragma unroll
        for (int t = 0; t < kBLoads; ++t)
            nb[t] = load_u32_unaligned(cb + boff[t]);
#pragma unroll
        for (int t = 0; t < kILoads; ++t)
            ni[t] = load_u32_unaligned(ci + ioff[t]);
    };
    if (i < n) {
        cur = sel[i];
        fetch(cur);
        if (i + step < n)
            nxt = sel[i + step];
    }
    while (i < n) {
        const int x0 = cur.x, y0 = cur.y;
        const float harris = cur.harris;
#pragma unroll
        for (int t = 0; t < kBLoads; ++t)
            if (lane + 64 * t < kBWords)
                reinterpret_cast<uint32_t *>(sb)[lane + 64 * t] = nb[t];
#pragma unroll
        for (int t = 0; t < kILoads; ++t)
            if (lane + 64 * t < kIWords)
                reinterpret_cast<uint32_t *>(si)[lane + 64 * t] = ni[t];
        wave_lds_sync();
        const int inext = i + step;
        if (inext < n) {
            cur = nxt;
            fetch(cur);   // in flight while this keypoint is computed
            if (inext + step < n)
                nxt = sel[inext + step];
        }
        // intensity-centroid moments over the radius-15 disc
        int m10, m01;
        {
            const uint4 lo = *reinterpret_cast<const uint4 *>(si + mrow * kIPitch);
            const uint4 hi = *reinterpret_cast<const uint4 *>(si + mrow * kIPitch + 16);
            const uint32_t p[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
            uint32_t s1 = 0, su = 0;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                s1 = __builtin_amdgcn_udot4(p[j], wm[j], s1, false);
                su = __builtin_amdgcn_udot4(p[j], wu[j], su, false);
            }
            m10 = (int)su - 16 * (int)s1;
            m01 = mv * (int)s1;
        }
#pragma unroll
        for (int s = 1; s < 64; s <<= 1) {   // integer sums: any order gives the same result
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
        const size_t o = (size_t)b * d.nfeatures + offset + i;
        {
            const uint8_t *ctr = sb + kBR * kBPitch + kBR;
            unsigned v = 0;
#pragma unroll
            for (int bit = 0; bit < 4; ++bit) {
                const float x1 = pat[4 * bit], y1 = pat[4 * bit + 1], x2 = pat[4 * bit + 2], y2 = pat[4 * bit + 3];
                const int ix1 = (int)rintf(x1 * ca - y1 * sa), iy1 = (int)rintf(x1 * sa + y1 * ca);
                const int ix2 = (int)rintf(x2 * ca - y2 * sa), iy2 = (int)rintf(x2 * sa + y2 * ca);
                v |= (unsigned)(ctr[iy1 * kBPitch + ix1] < ctr[iy2 * kBPitch + ix2]) << bit;
            }
            // nibble of lane l = bits 4 (l & 1) .. of byte l >> 1; bytes 4 g .. 4 g + 3 (g = l >> 3) make dword g
            v <<= 4 * (lane & 1);
            v |= __shfl_xor(v, 1, 64);
            v <<= 8 * ((lane >> 1) & 3);
            v |= __shfl_xor(v, 2, 64);
            v |= __shfl_xor(v, 4, 64);
            if ((lane & 7) == 0)
                reinterpret_cast<uint32_t *>(d.desc + o * 32)[lane >> 3] = v;
        }
        if (lane == 0) {
            mvs_keypoint k;
            k.x = (float)x0 * fs;
            k.y = (float)y0 * fs;
            k.size = 31.0f * fs;
            k.angle = fast_atan2_deg(f01, f10);
            k.response = harris;
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
        wave_lds_sync();   // every read of the windows before the next keypoint's are written
        i = inext;
    }
}

__global__ void orb_clear_kernel(int32_t *count_a, int32_t *count_b, int n, int32_t *overflow)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        count_a[i] = 0;
        count_b[i] = 0;
    }
    if (i == 0)
        *overflow = 0;
}

}  // namespace

size_t orb_sel_bytes() { return sizeof(Sel); }

hipError_t orb_prepare()
{
    return hipFuncSetAttribute((const void *)select_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)((size_t)kOrbSelCap * sizeof(uint64_t)));
}

// Measured and not kept: level 0's FAST + blur on a second stream beside the pyramid chain (a fork inside the captured graph):
// 110.6 k against 111.2 k images/s -- the graph's launches already follow each other without gaps.
void launch_orb(const OrbDev &d, hipStream_t stream)
{
    const int B = d.n_images;
    if (B <= 0)
        return;
    const int slots = B * d.n_levels;
    hipLaunchKernelGGL(orb_clear_kernel, dim3((slots + 255) / 256), dim3(256), 0, stream, d.cand_count, d.sel_count, slots,
                       d.overflow);
    // tile lists of every level first (host arithmetic only)
    OrbGrid g{};
    int max_keep = 0, n_live = 0;
    for (int l = 0; l < d.n_levels; ++l) {
        const OrbLevel &L = d.level[l];
        g.fast_start[l + 1] = g.fast_start[l];
        g.blur_start[l + 1] = g.blur_start[l];
        g.fast_tx[l] = g.blur_tx[l] = 1;
        if (n_live == l && L.w >= 1 && L.h >= 1)
            n_live = l + 1;   // (levels behind an empty one are empty too)
        if (n_live <= l || L.w <= 2 * d.edge || L.h <= 2 * d.edge || L.n_keep < 1)
            continue;
        g.fast_tx[l] = (L.w - 2 * d.edge + kTileW - 1) / kTileW;
        g.fast_start[l + 1] += g.fast_tx[l] * ((L.h - 2 * d.edge + kTileH - 1) / kTileH);
        g.blur_tx[l] = (L.w + kTileW - 1) / kTileW;
        g.blur_start[l + 1] += g.blur_tx[l] * ((L.h + kBlurH - 1) / kBlurH);
        max_keep = std::max(max_keep, L.n_keep);
    }
    auto detect = [&](int first_level, int end_level, hipStream_t st) {   // FAST + NMS and blur of levels [first, end)
        const int f0 = g.fast_start[first_level], fn = g.fast_start[end_level] - f0;
        const int b0 = g.blur_start[first_level], bn = g.blur_start[end_level] - b0;
        if (fn > 0)
            hipLaunchKernelGGL(fast_nms_kernel, dim3(fn * B), dim3(256), 0, st, d, g, f0, fn);
        if (bn > 0)
            hipLaunchKernelGGL(blur_kernel, dim3(bn * B), dim3(256), 0, st, d, g, b0, bn);
    };
    // the pyramid (level l from level l - 1), then ONE launch each for FAST + NMS and the blur over the tiles of every level
    const dim3 blk(32, 8);
    for (int l = 1; l < n_live; ++l) {
        const OrbLevel &L = d.level[l], &Pv = d.level[l - 1];
        const dim3 grid((L.w + 127) / 128, (L.h + 7) / 8, B);   // four pixels per thread
        hipLaunchKernelGGL(resize_kernel, grid, blk, 0, stream, d.pyr + Pv.offset * B, Pv.w, Pv.h, d.pyr + L.offset * B, L.w,
                           L.h, d.resize_tab + L.tab_offset, d.resize_tab + L.tab_offset + L.w);
    }
    detect(0, d.n_levels, stream);
    size_t sel_keys = 1;
    while (sel_keys < (size_t)std::min(2 * (long long)max_keep, (long long)kOrbSelCap))   // (orb_run caps the lists when 2 n_l is larger)
        sel_keys <<= 1;
    hipLaunchKernelGGL(select_kernel, dim3(B, d.n_levels), dim3(1024), sel_keys * sizeof(uint64_t), stream,
                       d);   // LDS limit raised in orb_prepare()
    const int split = B >= 8 ? kDescSplit : kDescSplitFew;
    hipLaunchKernelGGL(describe_kernel, dim3(split * d.n_levels * B), dim3(256), 0, stream, d, split);
}

}  // namespace mvs
