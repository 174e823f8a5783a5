// kernels.hip -- the four kernels of the two-view-geometry path, written for gfx950 (wave64).
//
//   match_topk     brute-force Hamming 2-NN + Lowe ratio / max-dist       (visual-feature.cpp:51-70)
//   match_compact  canonical sort (distance, queryIdx) + keypoint gather +
//                  K^-1 normalisation                                       (visual-feature.cpp:72-80,
//                                                                            image-pair.cpp:123-140, camera.cpp:55-79)
//   ransac         one hypothesis per LANE: sample 8 -> normalise -> A^T A -> 9x9 Jacobi SVD ->
//                  rank-2 -> de-normalise -> score against all M matches    (fundamental-matrix.cpp,
//                                                                            estimator-RANSAC.cpp)
//   finalize       arg-best, inlier mask, E projection, decomposition, 4 x M_inl triangulations
//                  (4x4 Jacobi SVD per lane), candidate selection, pose     (sfm-solve.cpp:64-368)
//
// Why one hypothesis per lane (not per wavefront): the 8-point solve is a long serial
// chain on an 81+81-double state with a handful of transcendental-rate scalars per rotation;
// spreading one 9x9 problem over lanes leaves >80 % of the wave idle in the rotation-angle
// computation and needs a cross-lane reduction per dot product, whereas 64 independent
// problems per wave keep every lane busy, need no cross-lane traffic, and give the
// reference's sequential residual order for free.  See DESIGN.md.
#include <algorithm>
#include "kernels.hpp"

#include "device_math.hpp"
#include "prescreen.hpp"

namespace mvs {

// ---------------------------------------------------------------------------------------------
// match_topk: grid (ceil(N/64), P), block 1024 = 16 waves.  Lane = one query descriptor (registers).  The train
// descriptors are staged through LDS in 64 KB tiles with coalesced 16-byte loads; wave w scans sixteenth w of each
// tile reading every train row with broadcast ds_read_b128 (in order -> pipelined), xor + popcount against the
// lane's query, running top-2 per lane.  The 16 partial lists are merged through LDS in (distance, train index)
// order, which reproduces the strict-'<' insertion of a sequential scan (OpenCV brute-force k-NN).
// ---------------------------------------------------------------------------------------------
// The running top-2 of a lane is kept as two KEYS (distance << 16 | train index), k0 <= k1: keys are unique per train
// row and their order is the (distance, train index) order, i.e. the strict-'<' insertion of a scan in index order.
// Inserting k is k0' = min(k0, k), k1' = med3(k0, k1, k) -- two instructions instead of a divergent compare / shift chain
// (the kernel is bound by its vector instruction count: 16 for the xor + popcount of 256 bits, 10 for the old insertion).
constexpr uint32_t kKeyNone = 0xffffffffu;
__device__ __forceinline__ uint32_t bcnt_acc(uint32_t x, uint32_t acc)
{
    uint32_t r;
    asm("v_bcnt_u32_b32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(acc));
    return r;
}
__device__ __forceinline__ void key_insert(uint32_t &k0, uint32_t &k1, uint32_t k)
{
    uint32_t m;   // min(k1, max(k0, k)) = the median of the three, given k0 <= k1
    asm("v_med3_u32 %0, %1, %2, %3" : "=v"(m) : "v"(k0), "v"(k1), "v"(k));
    k1 = m;
    k0 = min(k0, k);
}

constexpr int kTopkWaves = 16;
constexpr int kTopkTileBytes = 65536;

template <int DW>
__global__ __launch_bounds__(1024) void match_topk_kernel(BatchDev b, double ratio, double max_dist)
{
    constexpr int kTileRows = kTopkTileBytes / (DW * 4);
    __shared__ __attribute__((aligned(16))) uint32_t s_tile[kTopkTileBytes / 4];
    __shared__ uint32_t s_k0[kTopkWaves][64], s_k1[kTopkWaves][64];
    static_assert(DW * 32 < 0xffff && kMaxKp <= 0x10000, "distance and train index share a 32-bit key");
    const int pair = blockIdx.y;
    const int n1 = min(b.n1[pair], b.max_kp), n2 = min(b.n2[pair], b.max_kp);
    const int q0 = blockIdx.x * 64;
    if (q0 >= n2)
        return;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int q = q0 + lane;
    const bool live = q < n2;
    const size_t base = (size_t)pair * b.max_kp;

    uint32_t qv[DW];
    {
        const uint32_t *qd = b.desc2 + (base + (live ? q : q0)) * DW;
#pragma unroll
        for (int k = 0; k < DW; k += 4) {
            const uint4 v = *reinterpret_cast<const uint4 *>(qd + k);
            qv[k] = v.x; qv[k + 1] = v.y; qv[k + 2] = v.z; qv[k + 3] = v.w;
        }
    }
    const uint4 *tr4 = reinterpret_cast<const uint4 *>(b.desc1 + base * DW);

    uint32_t k0 = kKeyNone, k1 = kKeyNone;
    for (int tile0 = 0; tile0 < n1; tile0 += kTileRows) {
        const int rows = min(kTileRows, n1 - tile0);
        __syncthreads();  // previous tile fully consumed
        for (int i = tid; i < rows * (DW / 4); i += 1024)
            reinterpret_cast<uint4 *>(s_tile)[i] = tr4[(size_t)tile0 * (DW / 4) + i];
        __syncthreads();
        const int chunk = (rows + kTopkWaves - 1) / kTopkWaves;
        const int r0 = __builtin_amdgcn_readfirstlane(w * chunk);
        const int r1 = min(rows, r0 + chunk);
        auto row = [&](int r) {
            const uint4 *td = reinterpret_cast<const uint4 *>(s_tile + r * DW);  // wave-uniform -> LDS broadcast
            uint32_t d = 0;
#pragma unroll
            for (int k = 0; k < DW; k += 4) {
                const uint4 v = td[k / 4];
                d = bcnt_acc(qv[k] ^ v.x, d);       // v_bcnt_u32_b32 adds its second operand: one instruction per
                d = bcnt_acc(qv[k + 1] ^ v.y, d);   // dword (hipcc otherwise counts into fresh registers and adds
                d = bcnt_acc(qv[k + 2] ^ v.z, d);   // them up with v_add3: 11 instead of 8 per row)
                d = bcnt_acc(qv[k + 3] ^ v.w, d);
            }
            key_insert(k0, k1, (d << 16) | (uint32_t)(tile0 + r));
        };
        int r = r0;   // unrolled by hand: hipcc does not unroll a loop with inline asm in it
        for (; r + 4 <= r1; r += 4) {
            row(r);
            row(r + 1);
            row(r + 2);
            row(r + 3);
        }
        for (; r < r1; ++r)
            row(r);
    }

    s_k0[w][lane] = k0;
    s_k1[w][lane] = k1;
    __syncthreads();
    if (w == 0 && live) {
        uint32_t m0 = kKeyNone, m1 = kKeyNone;
#pragma unroll
        for (int c = 0; c < kTopkWaves; ++c) {
            key_insert(m0, m1, s_k0[c][lane]);
            key_insert(m0, m1, s_k1[c][lane]);
        }
        const int D0 = m0 == kKeyNone ? 0x7fffffff : (int)(m0 >> 16), D1 = m1 == kKeyNone ? 0x7fffffff : (int)(m1 >> 16);
        const int I0 = m0 == kKeyNone ? -1 : (int)(m0 & 0xffffu);
        // Lowe ratio in double on float distances (visual-feature.cpp:67-68)
        const float f0 = (float)D0, f1 = (float)D1;
        const bool check1 = (double)f0 < ratio * (double)f1;
        const bool check2 = (max_dist < 0.0) || ((double)f0 <= max_dist);
        const bool pass = (n1 >= 2) && check1 && check2;
        b.knn_train[base + q] = pass ? I0 : -1;
        b.knn_dist[base + q] = D0;
    }
}

// ---------------------------------------------------------------------------------------------
// match_mfma (256-bit descriptors): the same 2-NN on the matrix cores.  Hamming(q, t) = |q| + |t| - 2 q.t with the bits as
// {0, 1} int8: the all-pairs dot products of a pair are one [train x 256] . [256 x query] GEMM, exact in the i32
// accumulators of v_mfma_i32_32x32x32_i8 (8 k-steps per 32 x 32 tile).  grid (ceil(N / 256), P), block 512 = 8 wavefronts;
// wavefront w owns 32 queries (the MFMA's columns = lanes) whose unpacked bits stay in registers for the whole scan (the B
// operand: 8 x 4 VGPRs); trains go by in tiles of 32 rows (the A operand), unpacked ONCE per workgroup into LDS (bit ->
// byte: a nibble times 0x00204081, masked with 0x01010101, is its four bits as four bytes) and read by every wavefront
// with ds_read_b128.  The accumulator tile has the query on the lane and 16 trains in the registers: key = ((|t| + 256) << 16
// | index) - (dot << 17) is ONE v_mad_i32_i24 per element (the query's own popcount, equal for all its keys, is added at the
// end; unique keys ordered like (distance, index), as in match_topk), inserted into the lane's running top-2 with v_min_u32 +
// v_med3_u32.  Lanes l and l + 32 hold the same query over different rows and merge at the end.  Both operands take the
// same (lane half, element) -> bit mapping, so the dot product does not depend on the k order inside a step.
// ---------------------------------------------------------------------------------------------
constexpr int kMmThreads = 256;
constexpr int kMmQpw = 64;                           // queries per wavefront: two 32-column blocks share every A fragment read
constexpr int kMmQueries = kMmQpw * (kMmThreads / 64);   // 256 queries per workgroup
constexpr int kMmRowBytes = 256 + 16;                // unpacked row of a train tile, padded: ds_read_b128 of 32 rows spread over the banks
constexpr int kMmKeyShift = 12;                      // key = distance field << 12 | train index: one int8 product (64 x -128) is -(2 << 12)
static_assert(kMaxKp <= (1 << kMmKeyShift), "the train index must fit below the distance field of a match_mfma key");
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

// 16 bits -> 16 bytes (element j = bit j), each byte 0 or 1 << SH.  A nibble times 0x00204081 has bit i of the nibble at
// positions i + 7 k (k = 0..3: no two terms meet, no carries); the mask keeps bit 0 of every byte, the shift moves it to bit
// SH.  SH = 6: the train operand (0 / 64); SH = 7: the query operand (0 / 0x80 = -128 as int8), so that one product is
// -8192 = -(2 << 12): the accumulator counts the dot product in units of the key's distance field (below).  The product is a
// 4-bit by 22-bit one: v_mul_u32_u24 (full rate; v_mul_lo_u32 issues at a quarter of it -- eight of them per thread and tile
// were a third of the staging's issue slots).
template <int SH>
__device__ __forceinline__ v4i unpack16(uint32_t bits)
{
    v4i r;
    r.x = (int)((__umul24((bits >> 0) & 0xfu, 0x00204081u) & 0x01010101u) << SH);
    r.y = (int)((__umul24((bits >> 4) & 0xfu, 0x00204081u) & 0x01010101u) << SH);
    r.z = (int)((__umul24((bits >> 8) & 0xfu, 0x00204081u) & 0x01010101u) << SH);
    r.w = (int)((__umul24((bits >> 12) & 0xfu, 0x00204081u) & 0x01010101u) << SH);
    return r;
}

__device__ __forceinline__ uint32_t umin3(uint32_t a, uint32_t b, uint32_t c)
{
    uint32_t r;
    asm("v_min3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// CAPPED (round 5): the caller's max_dist >= 0 makes most keys irrelevant.  A query passes iff D0 <= max_dist and
// D0 < ratio * D1 (visual-feature.cpp:67-68); with C = the smallest integer whose ratio * C exceeds max_dist (host, the same
// float -> double expression as below), any D1 >= C passes the ratio test for every D0 <= max_dist, and any D0 >= C fails the
// distance test.  So only keys with a distance below C can change the result: the accumulators of a tile are reduced to group
// minima (v_min3_u32: 10 instructions per 16 keys instead of 32) and a group is inserted -- in full, so the lane's two smallest
// RELEVANT keys are always exact -- only when some lane of the wavefront sees a relevant key in it (a query has one true
// partner among 2000 random rows: 0.4 of the blocks take the slow path, for one group of four).  Keys at or above the cap that
// ride along are real keys of real rows: wherever they end up in the top two they stand for "some distance >= C", which is
// what the exact list would show there too.  Match lists stay byte-identical (tests/test_gpu_parity.py::test_match_*).
template <bool CAPPED>
__global__ __launch_bounds__(kMmThreads) __attribute__((amdgpu_waves_per_eu(3, 8))) void match_mfma_kernel(BatchDev b, double ratio,
                                                                                                            double max_dist, int cap_dist)
{
    __shared__ __attribute__((aligned(16))) unsigned char s_tile[2][32 * kMmRowBytes];   // unpacked train tiles (double buffer)
    __shared__ __attribute__((aligned(16))) uint32_t s_key[2][32];                        // (|t| + 256) << 12 | train index
    __shared__ uint32_t s_k0[2][kMmThreads], s_k1[2][kMmThreads];
    const int pair = blockIdx.y;
    const int n1 = min(b.n1[pair], b.max_kp), n2 = min(b.n2[pair], b.max_kp);
    const int q0 = blockIdx.x * kMmQueries;
    if (q0 >= n2)
        return;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, col = lane & 31, half = lane >> 5;
    const size_t base = (size_t)pair * b.max_kp;
    // B operands: this lane's 16-bit slices of its two queries (column blocks 0 and 1), one per k-step, unpacked once
    v4i Bf[2][8];
    int qn[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const int q = q0 + w * kMmQpw + c * 32 + col;
        const uint32_t *qd = b.desc2 + (base + (q < n2 ? q : q0)) * 8;
        const uint4 lo = *reinterpret_cast<const uint4 *>(qd), hi = *reinterpret_cast<const uint4 *>(qd + 4);
        const uint32_t d[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
        qn[c] = 0;
#pragma unroll
        for (int s8 = 0; s8 < 8; ++s8) {
            qn[c] += __popc(d[s8]);
            Bf[c][s8] = unpack16<7>(d[s8] >> (16 * half));
        }
    }
    const uint32_t *tr = b.desc1 + base * 8;
    // stage tile `t` (trains [32 t, 32 t + 32)) into buffer `buf`: thread = (row, dword): 32 bits -> 32 bytes
    auto stage = [&](int t, int buf) {
        const int row = tid >> 3, dw = tid & 7;
        const int tr_i = t * 32 + row;
        const uint32_t bits = tr_i < n1 ? tr[(size_t)tr_i * 8 + dw] : 0u;
        unsigned char *dst = &s_tile[buf][row * kMmRowBytes + dw * 32];
        *reinterpret_cast<v4i *>(dst) = unpack16<6>(bits);
        *reinterpret_cast<v4i *>(dst + 16) = unpack16<6>(bits >> 16);
        // the tile's key bases: |t| of the row, summed over its 8 dwords (8 adjacent lanes)
        int tn = __popc(bits);
        tn += __shfl_xor(tn, 1);
        tn += __shfl_xor(tn, 2);
        tn += __shfl_xor(tn, 4);
        if (dw == 0)
            s_key[buf][row] = tr_i < n1 ? (((uint32_t)(tn + 256) << kMmKeyShift) | (uint32_t)tr_i) : 0xffffffffu;
    };
    const int n_tiles = (n1 + 31) / 32;
    uint32_t k0[2] = {kKeyNone, kKeyNone}, k1[2] = {kKeyNone, kKeyNone};
    // relevant keys of this lane's two queries: distance = (key >> 12) - 256 + |q| < cap_dist
    const uint32_t capk[2] = {(uint32_t)(cap_dist + 256 - qn[0]) << kMmKeyShift, (uint32_t)(cap_dist + 256 - qn[1]) << kMmKeyShift};
    stage(0, 0);
    __syncthreads();
    for (int t = 0; t < n_tiles; ++t) {
        const int buf = t & 1;
        if (t + 1 < n_tiles)
            stage(t + 1, buf ^ 1);
        // The accumulators START as the rows' key bases and the products are -(2 << 12) per common bit, so the tile product
        // leaves the finished keys: key = ((|t| + 256 - 2 dot) << 12 | index) -- no vector instruction per element beyond the
        // insertion itself (round 4: one v_mad_i32_i24 per element on top; 96 -> 64 vector instructions per tile pair).
        // accumulator: column = lane & 31 (this lane's query), row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5) (the train):
        // registers 4 g4 .. 4 g4 + 3 are rows 8 g4 + 4 half + (0 .. 3), whose key bases are 16 consecutive bytes.  Rows past
        // the end have all-zero bits and the base 0xffffffff: the key stays kKeyNone, never selected.
        v16i kb;
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            const v4i kb4 = *reinterpret_cast<const v4i *>(&s_key[buf][8 * g4 + 4 * half]);
#pragma unroll
            for (int e = 0; e < 4; ++e)
                kb[4 * g4 + e] = kb4[e];
        }
        v16i acc0, acc1;
        const unsigned char *arow = &s_tile[buf][col * kMmRowBytes + half * 16];   // A: row = lane & 31, k = 16 half + j
#pragma unroll
        for (int s8 = 0; s8 < 8; ++s8) {
            const v4i Af = *reinterpret_cast<const v4i *>(arow + s8 * 32);
            // two independent accumulation chains; the first step of each reads the key bases as its C operand
            acc0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(Af, Bf[0][s8], s8 == 0 ? kb : acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(Af, Bf[1][s8], s8 == 0 ? kb : acc1, 0, 0, 0);
        }
        if (!CAPPED) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                key_insert(k0[0], k1[0], (uint32_t)acc0[r]);
                key_insert(k0[1], k1[1], (uint32_t)acc1[r]);
            }
        } else {
            auto lazy = [&](const v16i &acc, uint32_t &a0, uint32_t &a1, uint32_t cap) {
                uint32_t g[4];
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4)
                    g[q4] = min(umin3((uint32_t)acc[4 * q4], (uint32_t)acc[4 * q4 + 1], (uint32_t)acc[4 * q4 + 2]), (uint32_t)acc[4 * q4 + 3]);
                const uint32_t tm = min(umin3(g[0], g[1], g[2]), g[3]);
                if (__any(tm < cap)) {       // wave-uniform: some lane has a relevant key in this tile
#pragma unroll
                    for (int q4 = 0; q4 < 4; ++q4)
                        if (__any(g[q4] < cap)) {
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                key_insert(a0, a1, (uint32_t)acc[4 * q4 + e]);
                        }
                }
            };
            lazy(acc0, k0[0], k1[0], capk[0]);
            lazy(acc1, k0[1], k1[1], capk[1]);
        }
        __syncthreads();
    }
    // lanes l and l + 32 of a wavefront saw the same query over different rows: merge through LDS
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        s_k0[c][tid] = k0[c];
        s_k1[c][tid] = k1[c];
    }
    __syncthreads();
    if (half == 0) {
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int q = q0 + w * kMmQpw + c * 32 + col;
            if (q >= n2)
                continue;
            uint32_t m0 = k0[c], m1 = k1[c];
            key_insert(m0, m1, s_k0[c][tid + 32]);
            key_insert(m0, m1, s_k1[c][tid + 32]);
            // distance = (key >> 12) - 256 + |q|
            const int D0 = m0 == kKeyNone ? 0x7fffffff : (int)(m0 >> kMmKeyShift) - 256 + qn[c];
            const int D1 = m1 == kKeyNone ? 0x7fffffff : (int)(m1 >> kMmKeyShift) - 256 + qn[c];
            const int I0 = m0 == kKeyNone ? -1 : (int)(m0 & ((1u << kMmKeyShift) - 1u));
            // Lowe ratio in double on float distances (visual-feature.cpp:67-68)
            const float f0 = (float)D0, f1 = (float)D1;
            const bool check1 = (double)f0 < ratio * (double)f1;
            const bool check2 = (max_dist < 0.0) || ((double)f0 <= max_dist);
            const bool pass = (n1 >= 2) && check1 && check2;
            b.knn_train[base + q] = pass ? I0 : -1;
            b.knn_dist[base + q] = D0;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// match_compact: grid P, block 1024.  Bitonic sort of the keys (distance << 16 | queryIdx) in LDS (unique keys,
// rejected queries = 0xffffffff sort to the end) = the canonical (distance, queryIdx) order; then match m gathers its
// two keypoints and applies K^-1.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void match_compact_kernel(BatchDev b)
{
    __shared__ __attribute__((aligned(16))) uint32_t s_key[kMaxKp];
    __shared__ int s_count;
    const int pair = blockIdx.x;
    const int n2 = min(b.n2[pair], b.max_kp);
    const size_t base = (size_t)pair * b.max_kp;
    const int tid = threadIdx.x;
    if (tid == 0)
        s_count = 0;
    int S = 2;
    while (S < n2)
        S <<= 1;
    for (int q = tid; q < S; q += 1024) {
        uint32_t key = 0xffffffffu;
        if (q < n2) {
            const int tr = b.knn_train[base + q];
            if (tr >= 0)
                key = ((uint32_t)b.knn_dist[base + q] << 16) | (uint32_t)q;
        }
        s_key[q] = key;
    }
    __syncthreads();
    for (int k = 2; k <= S; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = tid; t < (S >> 1); t += 1024) {
                const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                const int ixj = i | j;
                const uint32_t x = s_key[i], y = s_key[ixj];
                const bool up = (i & k) == 0;
                if ((x > y) == up) {
                    s_key[i] = y;
                    s_key[ixj] = x;
                }
            }
            __syncthreads();
        }
    }
    int local = 0;
    const double *Ki = b.Kinv + (size_t)pair * 9;
    const double k0 = Ki[0], k1 = Ki[1], k2 = Ki[2], k3 = Ki[3], k4 = Ki[4], k5 = Ki[5];
    // rows [M, max_kp) of the match list are cleared, so that a download of the whole capacity is deterministic
    // (valid keys sort first: row m is a match iff its key is valid)
    for (int m = tid; m < b.max_kp; m += 1024) {
        const uint32_t key = m < n2 ? s_key[m] : 0xffffffffu;
        if (key == 0xffffffffu) {
            mvs_match z;
            z.queryIdx = 0; z.trainIdx = 0; z.imgIdx = 0; z.distance = 0.f;
            b.matches[base + m] = z;
            continue;
        }
        ++local;
        const int q = (int)(key & 0xffffu);
        const int tr = b.knn_train[base + q];
        mvs_match mt;
        mt.queryIdx = q;
        mt.trainIdx = tr;
        mt.imgIdx = 0;
        mt.distance = (float)(key >> 16);
        b.matches[base + m] = mt;
        // base_points[m] = kp1[trainIdx], pair_points[m] = kp2[queryIdx], float -> double, K^-1 (u, v, 1)
        const float2 a = *reinterpret_cast<const float2 *>(b.kp1 + (base + tr) * 2);
        const float2 c = *reinterpret_cast<const float2 *>(b.kp2 + (base + q) * 2);
        const double u1 = (double)a.x, v1 = (double)a.y, u2 = (double)c.x, v2 = (double)c.y;
        double4 p;
        p.x = (k0 * u1 + k1 * v1) + k2;
        p.y = (k3 * u1 + k4 * v1) + k5;
        p.z = (k0 * u2 + k1 * v2) + k2;
        p.w = (k3 * u2 + k4 * v2) + k5;
        *reinterpret_cast<double4 *>(b.pts + (base + m) * 4) = p;
    }
    if (local)
        atomicAdd(&s_count, local);
    __syncthreads();
    if (tid == 0)
        b.M[pair] = s_count;
}

// prep_points: single-shot sfm_solve / sfm_triangulate entry: image points are given directly.
// grid (ceil(N/256), P).  uv: [P][N][2] doubles; M[pair] preset by the host.
__global__ __launch_bounds__(256) void prep_points_kernel(BatchDev b, const double *uv1, const double *uv2)
{
    const int pair = blockIdx.y;
    const int m = blockIdx.x * 256 + threadIdx.x;
    if (m >= b.M[pair])
        return;
    const size_t base = (size_t)pair * b.max_kp;
    const double *Ki = b.Kinv + (size_t)pair * 9;
    const double u1 = uv1[(base + m) * 2], v1 = uv1[(base + m) * 2 + 1];
    const double u2 = uv2[(base + m) * 2], v2 = uv2[(base + m) * 2 + 1];
    double4 p;
    p.x = (Ki[0] * u1 + Ki[1] * v1) + Ki[2];
    p.y = (Ki[3] * u1 + Ki[4] * v1) + Ki[5];
    p.z = (Ki[0] * u2 + Ki[1] * v2) + Ki[2];
    p.w = (Ki[3] * u2 + Ki[4] * v2) + Ki[5];
    *reinterpret_cast<double4 *>(b.pts + (base + m) * 4) = p;
}

// ---------------------------------------------------------------------------------------------
// ransac: grid (G, P), block 256 (one wave per SIMD; the 9x9 state needs the whole register file).
// ---------------------------------------------------------------------------------------------
struct Cand {
    int cnt;
    uint32_t hyp;
    double res;
};

// the reference's sequential replacement rule (estimator-RANSAC.cpp:76-84) as a total order:
// more inliers, then smaller residual, then smaller hypothesis id.
__device__ __forceinline__ bool cand_better(const Cand &a, const Cand &b)
{
    if (a.cnt != b.cnt)
        return a.cnt > b.cnt;
    if (a.res != b.res)
        return a.res < b.res;
    return a.hyp < b.hyp;
}

__device__ __forceinline__ double pair_max_error_sq(const BatchDev &b, const RunParams &rp, int pair)
{
    if (rp.max_error_sq > 0.0)
        return rp.max_error_sq;
    const double *K = b.K + (size_t)pair * 9;
    return 5e-2 / K[0] / K[4];  // sfm-solve.cpp:311
}

// VAR (co-compiled A/B variants): 8 = point stream staged in LDS (broadcast ds_read_b128: in order, so the compiler
// keeps many in flight; scalar loads return out of order and drain at every batch), 16 = in-place rotation,
// 32 = unscaled sqrt/div behind a range guard, 64 = mask-multiply residual accumulation
// sample 8 matches, gather them, fit F (one hypothesis, one lane)
template <int VAR>
__device__ __forceinline__ bool solve_hypothesis(uint64_t seed, uint32_t hyp, int M, int sampler, const double *P,
                                                 double (&F)[9], unsigned &rot, unsigned &pairs, bool &bad)
{
    int idx[8];
    sample8(seed, hyp, M, sampler, idx);
    double x1[8], y1[8], x2[8], y2[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const double4 p = *reinterpret_cast<const double4 *>(P + (size_t)idx[k] * 4);
        x1[k] = p.x; y1[k] = p.y; x2[k] = p.z; y2[k] = p.w;
    }
    return eight_point<VAR>(x1, y1, x2, y2, F, rot, pairs, bad);
}

template <bool STATS, int VAR>
__global__ __launch_bounds__(256, 1) void ransac_kernel(BatchDev b, RunParams rp)
{
    const int pair = blockIdx.y, g = blockIdx.x, tid = threadIdx.x;
    const int M = b.M[pair];
    WgBest *out = b.wgbest + (size_t)pair * b.max_groups + g;
    if (M < 8) {  // estimator-RANSAC.cpp:25-29
        if (tid == 0) {
            out->count = -1;
            out->hyp = 0xffffffffu;
            out->residual = 0.0;
        }
        return;
    }
    const int H = rp.num_hypotheses;
    const uint32_t h = (uint32_t)g * kHypPerBlock + tid;
    const bool live = h < (uint32_t)H;
    const uint32_t hh = live ? h : (uint32_t)(H - 1);
    const uint64_t seed = rp.seed + (uint64_t)b.gidx[pair];
    const double *P = b.pts + (size_t)pair * b.max_kp * 4;

    // rows 1..8 of every lane's Vt (72 doubles x 256 lanes = 144 KB: one workgroup per CU, one wave per SIMD)
    constexpr int kLdsDoubles = (VAR & 8) ? kMaxKp * 4 : 2;
    __shared__ __attribute__((aligned(16))) double s_vt[kLdsDoubles];
    if (VAR & 8) {
        // stage the pair's M point pairs (32 B each) in LDS once per workgroup, BEFORE the solve: all four waves
        // arrive here together, so the barrier is free (after the solve it would add the waves' run-time skew)
        const double2 *src = reinterpret_cast<const double2 *>(P);
        double2 *dst = reinterpret_cast<double2 *>(s_vt);
        for (int i = tid; i < 2 * M; i += kHypPerBlock)
            dst[i] = src[i];
        __syncthreads();
    }
    double F[9];
    unsigned rot = 0, pairs = 0;
    bool bad = false;
    bool ok = solve_hypothesis<VAR>(seed, hh, M, rp.sampler, P, F, rot, pairs, bad);
    if ((VAR & 32) && __builtin_expect(__any(bad), 0)) {
        // an operand left the range in which the unscaled sqrt / div sequences are provably IEEE-exact:
        // recompute this wave's hypotheses with the compiler's fully scaled sequences
        rot = 0;
        pairs = 0;
        ok = solve_hypothesis<(VAR & ~(32 | 128))>(seed, hh, M, rp.sampler, P, F, rot, pairs, bad);
    }

    // score against all M matches.  Every lane reads the SAME point: the address is wave-uniform, but it is made
    // opaque to the compiler so that it emits VECTOR loads (one cache line per wave-instruction, served by L1/L2).
    // Scalar loads return out of order, so every batch would need s_waitcnt lgkmcnt(0) and the loop would stall on
    // the full SMEM latency each iteration (20 % of the kernel in the first profile); vector loads retire in order
    // and the compiler pipelines them with counted vmcnt.
    const double thr = pair_max_error_sq(b, rp, pair);
    int cnt = 0;
    double res = 0.0;
    if (VAR & 8) {
        // every lane reads the SAME LDS address (broadcast).  LDS returns in order, so the compiler keeps many
        // reads in flight (counted lgkmcnt) instead of draining the scalar-load queue every batch.
        const double4 *L4 = reinterpret_cast<const double4 *>(s_vt);
#pragma unroll 8
        for (int i = 0; i < M; ++i) {
            const double4 p = L4[i];
            const double r = epipolar_residual(F, p.x, p.y, p.z, p.w);
            const bool in = r < thr;
            cnt += in ? 1 : 0;
            // (round 1-2 accumulated fma(r, in ? 1.0 : 0.0, res): one select + one fma, exact for FINITE r only -- a
            // degenerate sample can give an F with NaN entries, NaN * 0 poisons the sum where the reference adds nothing;
            // found by the adversarial pairs of tests/prescreen_gpu_check.py)
            res += in ? r : 0.0;   // adding +0.0 is exact: identical to the conditional add
        }
    } else {
        const double4 *P4 = reinterpret_cast<const double4 *>(P);
#pragma unroll 8
        for (int i = 0; i < M; ++i) {
            const double4 p = P4[i];
            const double r = epipolar_residual(F, p.x, p.y, p.z, p.w);
            const bool in = r < thr;  // strict (estimator-RANSAC.cpp:117)
            cnt += in ? 1 : 0;
            res += in ? r : 0.0;  // adding +0.0 is exact: identical to the conditional add
        }
    }
    if (!ok || !live) {
        cnt = -1;
        res = 0.0;
    }
    if (b.hyp_count && live) {
        b.hyp_count[(size_t)pair * H + h] = cnt;
        b.hyp_residual[(size_t)pair * H + h] = res;
    }
    if (STATS) {
        unsigned r = live ? rot : 0u, p = live ? pairs : 0u;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            r += __shfl_xor(r, o);
            p += __shfl_xor(p, o);
        }
        unsigned sw = live ? pairs / 36u : 0u;   // 9x9 sweeps of this lane's hypothesis (36 pair visits per sweep)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1)
            sw = max(sw, (unsigned)__shfl_xor((int)sw, o));
        if ((tid & 63) == 0) {
            atomicAdd(&b.stats[0], (unsigned long long)r);
            atomicAdd(&b.stats[1], (unsigned long long)p);
            atomicMax(&b.stats[6], (unsigned long long)sw);
        }
    }

    // workgroup arg-best
    Cand me{cnt, h, res};
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        Cand other;
        other.cnt = __shfl_xor(me.cnt, o);
        other.hyp = __shfl_xor(me.hyp, o);
        other.res = __shfl_xor(me.res, o);
        if (cand_better(other, me))
            me = other;
    }
    __shared__ Cand s_c[4];
    __shared__ uint32_t s_win;
    if ((tid & 63) == 0)
        s_c[tid >> 6] = me;
    __syncthreads();
    if (tid == 0) {
        Cand best = s_c[0];
#pragma unroll
        for (int w2 = 1; w2 < 4; ++w2)
            if (cand_better(s_c[w2], best))
                best = s_c[w2];
        s_win = best.hyp;
        out->count = best.cnt;
        out->hyp = best.hyp;
        out->residual = best.res;
    }
    __syncthreads();
    if (h == s_win) {
#pragma unroll
        for (int k = 0; k < 9; ++k)
            out->F[k] = F[k];
    }
}

// ---- split variant (VAR & 512): the solve and the scoring as two launches -----------------------------------------
// The fused kernel runs at 1 wave/SIMD (380 registers for the solve), where every non-fp64 instruction of the scoring
// loop costs a full fp64 issue slot (profiles/r01_fp64_issue_microbench.txt).  Scoring needs 18 registers of state: as
// its own kernel it runs at 4 waves/SIMD and the compare / count / mask instructions overlap with other waves' FMAs.
// Price: F of every hypothesis goes through HBM once (72 B x 50 000 x 512 pairs = 1.8 GB written + read per batch).
// one hypothesis per lane: exact solve, record, state byte
template <int VAR>
__device__ __forceinline__ void solve_record(const BatchDev &b, const RunParams &rp, int pair, int M, uint32_t h)
{
    const int H = rp.num_hypotheses;
    const uint32_t hh = h < (uint32_t)H ? h : (uint32_t)(H - 1);
    const uint64_t seed = rp.seed + (uint64_t)b.gidx[pair];
    const double *P = b.pts + (size_t)pair * b.max_kp * 4;
    double F[9];
    unsigned rot = 0, pairs = 0;
    bool bad = false;
    bool ok = solve_hypothesis<VAR>(seed, hh, M, rp.sampler, P, F, rot, pairs, bad);
    if ((VAR & 32) && __builtin_expect(__any(bad), 0)) {
        rot = 0;
        pairs = 0;
        ok = solve_hypothesis<(VAR & ~(32 | 128))>(seed, hh, M, rp.sampler, P, F, rot, pairs, bad);
    }
    const size_t Hp = (size_t)b.max_groups * kHypPerBlock;
    double *Fo = b.hyp_F + ((size_t)pair * Hp + h) * kHypRec;   // one contiguous record per hypothesis (scalar loads in
#pragma unroll                                                  // the counting kernels); the stores cover whole lines
    for (int k = 0; k < 9; ++k)
        Fo[k] = F[k];
    Fo[9] = pair_max_error_sq(b, rp, pair);   // exact F: counted against the threshold itself (band 0)
    b.hyp_okf[(size_t)pair * Hp + h] = ok ? kPsExact : kPsInvalid;
    if (h == 0)
        b.bound[pair] = 0;   // pruning bound of the scoring launch that follows on the stream
}


// The pairs the probe left in mode 0, in pair order: m0list[0] = their number, then the pairs (one workgroup; the pre-screened
// stage's exact-solve launch walks this list instead of sending a workgroup per (pair, 64 hypotheses) that leaves at once --
// 401 k of them, 0.13 ms, when every pair is pre-screened).
__global__ __launch_bounds__(256) void mode0_list_kernel(BatchDev b, int n_active)
{
    __shared__ int s_cnt[4];
    __shared__ int s_total;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    if (tid == 0)
        s_total = 0;
    __syncthreads();
    for (int p0 = 0; p0 < n_active; p0 += 256) {
        const int pair = p0 + tid;
        const bool take = pair < n_active && b.M[pair] >= 8 && b.mode[pair] == 0;
        const unsigned long long m = __ballot(take);
        if (lane == 0)
            s_cnt[w] = __popcll(m);
        __syncthreads();
        int base = s_total;
        for (int k = 0; k < w; ++k)
            base += s_cnt[k];
        if (take)
            b.m0list[1 + base + __popcll(m & ((1ull << lane) - 1ull))] = pair;
        __syncthreads();
        if (tid == 0)
            s_total += s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
        __syncthreads();
    }
    if (tid == 0)
        b.m0list[0] = s_total;
}

// exact solve of every hypothesis of the listed pairs: persistent single-wavefront workgroups striding over
// (listed pair, block of 64 hypotheses); the list length is fixed before the launch
template <int VAR>
__global__ __launch_bounds__(256, 1) void ransac_solve_list_kernel(BatchDev b, RunParams rp, int blocks_per_pair)
{
    const int n_items = b.m0list[0] * blocks_per_pair;
    for (int item = blockIdx.x; item < n_items; item += gridDim.x) {
        const int pair = b.m0list[1 + item / blocks_per_pair];
        solve_record<VAR>(b, rp, pair, b.M[pair], (uint32_t)(item % blocks_per_pair) * 64u + threadIdx.x);
    }
}

constexpr int kCntSlots = 4;   // hypotheses a wavefront of the point-per-lane counting kernels carries at a time
typedef __attribute__((address_space(4))) double CDouble;


// ---- sound pre-screen of the hypotheses (prescreen.hpp) -------------------------------------------------
// pair_prepare   grid P          bounding box of the pair's matches; probe of the first 64 hypotheses: a pair is pre-screened
//                                only if the certified band is useful at its threshold for most of them (otherwise every
//                                hypothesis of the pair is solved exactly by ransac_solve_kernel, as before)
// prescreen      grid (G', P)    approximate F + band per hypothesis -> record; hypotheses without a certificate -> work list
// exact_list     persistent      exact solve of the listed hypotheses, records overwritten in place (band 0)
// count2         grid (wg, P)    ransac_count_kernel with one counting threshold per hypothesis: upper bounds prune, lower
//                                bounds of the hypotheses that finish raise the pair's bound
// survivors      flat            approximate records whose upper bound reaches the final bound -> work list (-> exact_list)
// select         grid P          exact count + residual of everything at or above the bound (ransac_select_kernel)
__device__ __forceinline__ PairBox load_box(const BatchDev &b, int pair)
{
    const double *q = b.box + (size_t)pair * 8;
    PairBox bx;
    bx.x1lo = q[0]; bx.x1hi = q[1]; bx.y1lo = q[2]; bx.y1hi = q[3];
    bx.x2lo = q[4]; bx.x2hi = q[5]; bx.y2lo = q[6]; bx.y2hi = q[7];
    return bx;
}

// sample + gather + pre-screen of one hypothesis (one lane)
template <int VAR>
__device__ __forceinline__ int prescreen_sample(uint64_t seed, uint32_t hyp, int M, int sampler, const double *P, double *park,
                                                const PairBox &bx, double thr, double (&F)[9], double &band, double &e32,
                                                bool &bad3)
{
    int idx[8];
    sample8(seed, hyp, M, sampler, idx);
    return prescreen_hypothesis<VAR>(P, idx, park, bx, thr, F, band, e32, bad3);
}

constexpr int kPsVar = 16 + 32 + 128 + 1024;   // the 3x3 SVD of the pre-screen runs the solve's guarded pair step

__global__ __launch_bounds__(256) void pair_prepare_kernel(BatchDev b, RunParams rp, int force_mode)
{
    __shared__ double s_red[4][8];
    __shared__ double s_park[kPsParked * 64];
    const int pair = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int M = min(b.M[pair], b.max_kp);
    if (pair == 0 && tid < 2)
        b.xcount[tid] = 0u;
    if (tid == 0) {
        b.bound[pair] = 0;
        b.ccount[pair] = 0;
        b.pcount[pair] = 0;
    }
    if (M < 8) {
        if (tid == 0)
            b.mode[pair] = 0;
        return;
    }
    const double4 *P4 = reinterpret_cast<const double4 *>(b.pts + (size_t)pair * b.max_kp * 4);
    const double big = 0x1p1000;
    double lo[4] = {big, big, big, big}, hi[4] = {-big, -big, -big, -big};
    for (int i = tid; i < M; i += 256) {
        const double4 p = P4[i];
        const double v[4] = {p.x, p.y, p.z, p.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            lo[k] = fmin(lo[k], v[k]);
            hi[k] = fmax(hi[k], v[k]);
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            lo[k] = fmin(lo[k], __shfl_xor(lo[k], o));
            hi[k] = fmax(hi[k], __shfl_xor(hi[k], o));
        }
    }
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            s_red[w][2 * k] = lo[k];
            s_red[w][2 * k + 1] = hi[k];
        }
    }
    __syncthreads();
    PairBox bx;
    {
        double q[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const double a0 = s_red[0][k], a1 = s_red[1][k], a2 = s_red[2][k], a3 = s_red[3][k];
            q[k] = (k & 1) ? fmax(fmax(a0, a1), fmax(a2, a3)) : fmin(fmin(a0, a1), fmin(a2, a3));
        }
        bx.x1lo = q[0]; bx.x1hi = q[1]; bx.y1lo = q[2]; bx.y1hi = q[3];
        bx.x2lo = q[4]; bx.x2hi = q[5]; bx.y2lo = q[6]; bx.y2hi = q[7];
        if (tid < 8)
            b.box[(size_t)pair * 8 + tid] = q[tid];
    }
    // probe: the first 64 hypotheses of the pair through the pre-screen
    if (w == 0) {
        const int H = rp.num_hypotheses;
        const uint64_t seed = rp.seed + (uint64_t)b.gidx[pair];
        const double thr = pair_max_error_sq(b, rp, pair);
        const uint32_t hh = (uint32_t)min(lane, H - 1);
        double F[9], band, e32;
        bool bad3 = false;
        int flag = prescreen_sample<kPsVar>(seed, hh, M, rp.sampler, reinterpret_cast<const double *>(P4), s_park + lane, bx, thr,
                                            F, band, e32, bad3);
        if (bad3 && flag == kPsApprox)
            flag = kPsNeedExact;
        const bool live = lane < H;
        const int n_ok = __popcll(__ballot(live && flag != kPsInvalid));
        const int n_s64 = __popcll(__ballot(live && flag == kPsApprox && band <= kPsProbeFrac * thr));
        // e32 = 16 * 2^-24 T: the matrix cores' split-bf16 term 2^-14 T is 64 e32
        const int n_s32 = __popcll(__ballot(live && flag == kPsApprox && band + e32 <= kPsProbeFrac * thr &&
                                            band + 65.0 * e32 <= kPsProbeMfmaFrac * thr));
        // 1: pre-screened, counted in single precision; 2: pre-screened, counted in double precision (the band is useful
        // at this threshold but single-precision evaluation is too coarse for it); 0: every hypothesis solved exactly
        // the pre-screen costs about a tenth of an exact solve, so it pays as soon as a fair share of the hypotheses gets
        // a certificate (the others go to the exact solve either way): one third of the probe is asked for
        const int mode = n_ok == 0 ? 0 : 3 * n_s32 >= n_ok ? 1 : 3 * n_s64 >= n_ok ? 2 : 0;
        if (lane == 0)
            b.mode[pair] = force_mode >= 0 ? force_mode : mode;
    }
}

__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(3, 3))) void ransac_prescreen_kernel(BatchDev b, RunParams rp)
{
    __shared__ double s_park[kPsParked * 64];   // this wavefront's parked R entries: [entry][lane]
    const int pair = blockIdx.y, lane = threadIdx.x;
    const int M = b.M[pair];
    if (M < 8 || b.mode[pair] == 0)
        return;
    const int H = rp.num_hypotheses;
    const uint32_t h = blockIdx.x * 64 + lane;
    const bool live = h < (uint32_t)H;
    const uint32_t hh = live ? h : (uint32_t)(H - 1);
    const uint64_t seed = rp.seed + (uint64_t)b.gidx[pair];
    const double *P = b.pts + (size_t)pair * b.max_kp * 4;
    const PairBox bx = load_box(b, pair);
    const double thr = pair_max_error_sq(b, rp, pair);
    double F[9], band, e32;
    bool bad3 = false;
    int flag = prescreen_sample<kPsVar>(seed, hh, M, rp.sampler, P, s_park + lane, bx, thr, F, band, e32, bad3);
    const int mode = b.mode[pair];
    const double btot = mode == 1 ? band + e32 : band;   // what the counting kernel of this pair has to allow for
    if (flag == kPsApprox && (bad3 || !(btot <= kPsBandFrac * thr)))
        flag = kPsNeedExact;   // band too wide for the threshold, or a range guard of the unscaled 3x3 sequences was violated
    if (!live)
        flag = kPsInvalid;
    const size_t Hp = (size_t)b.max_groups * kHypPerBlock;
    const size_t rec = (size_t)pair * Hp + h;
    double *Fo = b.hyp_F + rec * kHypRec;
    if (mode == 1) {
        // single-precision record (48 bytes, its own array): F~ (9 floats), upper and lower counting thresholds rounded outwards
        float *fo = b.hyp_r32 + rec * kHypRec32;
        float q[12];
#pragma unroll
        for (int k = 0; k < 9; ++k)
            q[k] = (float)F[k];
        q[9] = (float)((thr + btot) * (1.0 + 0x1p-22));
        q[10] = (float)(fmax(thr - btot, 0.0) * (1.0 - 0x1p-22));   // (0: nothing is below it, the lower count bound is 0)
        q[11] = 0.f;
#pragma unroll
        for (int k = 0; k < 12; k += 4)
            *reinterpret_cast<float4 *>(fo + k) = make_float4(q[k], q[k + 1], q[k + 2], q[k + 3]);
    } else {
#pragma unroll
        for (int k = 0; k < 9; ++k)
            Fo[k] = F[k];
        Fo[9] = flag == kPsApprox ? thr + band : thr;
    }
    b.hyp_okf[rec] = (uint8_t)flag;
    // hypotheses without a certificate (flag kPsNeedExact) are put on the work list of the exact solve by
    // ransac_survivors_kernel together with the survivors of the counting: an append here was one returning atomic on a
    // single address per wavefront with a flagged lane (70 k per 512 pairs)
}

// exact solve of the hypotheses on the work list (flat indices pair * Hp + h; list `which`).  Persistent grid: wavefront
// w takes entries [64 w, 64 w + 64), then strides by the grid; the list length is fixed before the launch, so every
// wavefront reaches its exit.
template <int VAR>
__global__ __launch_bounds__(64, 1) void ransac_exact_list_kernel(BatchDev b, RunParams rp, int which)
{
    const unsigned n = b.xcount[which];
    const int lane = threadIdx.x;
    const size_t Hp = (size_t)b.max_groups * kHypPerBlock;
    for (unsigned base = blockIdx.x * 64u; base < n; base += gridDim.x * 64u) {
        const bool live = base + lane < n;
        const uint32_t rec = b.xlist[live ? base + lane : base];
        const int pair = (int)(rec / Hp);
        const uint32_t h = (uint32_t)(rec - (size_t)pair * Hp);
        const int M = b.M[pair];
        const uint64_t seed = rp.seed + (uint64_t)b.gidx[pair];
        const double *P = b.pts + (size_t)pair * b.max_kp * 4;
        double F[9];
        unsigned rot = 0, pairs = 0;
        bool bad = false;
        bool ok = solve_hypothesis<VAR>(seed, h, M, rp.sampler, P, F, rot, pairs, bad);
        if ((VAR & 32) && __builtin_expect(__any(bad), 0)) {
            rot = 0;
            pairs = 0;
            ok = solve_hypothesis<(VAR & ~(32 | 128))>(seed, h, M, rp.sampler, P, F, rot, pairs, bad);
        }
        if (live) {
            double *Fo = b.hyp_F + (size_t)rec * kHypRec;
#pragma unroll
            for (int k = 0; k < 9; ++k)
                Fo[k] = F[k];
            Fo[9] = pair_max_error_sq(b, rp, pair);
            b.hyp_okf[rec] = ok ? kPsExact : kPsInvalid;
        }
    }
}

__device__ __forceinline__ int count_block_thr(const double (&F)[9], const double4 &p, double thr)
{
    const double r = epipolar_residual(F, p.x, p.y, p.z, p.w);
    return __popcll(__ballot(r < thr));   // NaN (padding lanes, degenerate F) compares false
}

// ransac_count_kernel with one counting threshold per hypothesis (record[9] = thr + band).  The running counts are UPPER
// bounds of the exact counts: a slot dies when even its upper bound can no longer reach the pair's bound.  A slot that
// survives all blocks is counted once more against thr - band: a LOWER bound of its exact count, and only lower bounds
// raise the pair's bound.  Exact records have band 0: upper = lower = exact, no second pass -- the kernel then does
// exactly what ransac_count_kernel does.
template <int CNT_THREADS, int PPL, bool STATS = false>
__global__ __launch_bounds__(CNT_THREADS) void ransac_count2_kernel(BatchDev b, RunParams rp, int wg_per_pair)
{
    extern __shared__ __attribute__((aligned(16))) double s_cpts[];
    __shared__ int s_bound;
    const int pair = blockIdx.y, tid = threadIdx.x, lane = tid & 63;
    const int M = min(b.M[pair], b.max_kp);
    if (M < 8 || b.mode[pair] == 1)
        return;   // mode 1: this pair's records are single precision (ransac_count32_kernel)
    const int H = rp.num_hypotheses;
    const size_t Hp = (size_t)b.max_groups * kHypPerBlock;
    constexpr int BW = 64 * PPL;                  // points per block
    const int nblk = (M + BW - 1) / BW;
    double2 *s_p1 = reinterpret_cast<double2 *>(s_cpts);
    double2 *s_p2 = s_p1 + nblk * BW;
    {
        const double4 *src = reinterpret_cast<const double4 *>(b.pts + (size_t)pair * b.max_kp * 4);
        const double qnan = __builtin_nan("");
        for (int i = tid; i < nblk * BW; i += CNT_THREADS) {
            const double4 p = i < M ? src[i] : make_double4(qnan, qnan, qnan, qnan);
            s_p1[i] = make_double2(p.x, p.y);
            s_p2[i] = make_double2(p.z, p.w);
        }
    }
    int *gbound = b.bound + pair;
    if (tid == 0)
        s_bound = __hip_atomic_load(gbound, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const double thr = pair_max_error_sq(b, rp, pair);
    const double *Fp = b.hyp_F + (size_t)pair * Hp * kHypRec;
    const uint32_t *okp = reinterpret_cast<const uint32_t *>(b.hyp_okf + (size_t)pair * Hp);
    int32_t *cntp = b.hyp_cnt + (size_t)pair * Hp;
    const double2 *L1 = s_p1 + lane, *L2 = s_p2 + lane;
    const int n_groups = (H + kCntSlots - 1) / kCntSlots;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n_waves = wg_per_pair * (CNT_THREADS / 64);
    int B = 0;
    unsigned long long visits = 0;   // STATS: (hypothesis, block) evaluations this wavefront executed
    for (int g = blockIdx.x * (CNT_THREADS / 64) + wave; g < n_groups; g += n_waves) {
        const int h0 = g * kCntSlots;
        const int gb = __hip_atomic_load(gbound, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        B = __builtin_amdgcn_readfirstlane(max(B, *(volatile int *)&s_bound));
        double F0[9], F1[9], F2[9], F3[9];
        const CDouble *f = (const CDouble *)(uintptr_t)(Fp + (size_t)h0 * kHypRec);
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            F0[k] = f[k];
            F1[k] = f[kHypRec + k];
            F2[k] = f[2 * kHypRec + k];
            F3[k] = f[3 * kHypRec + k];
        }
        const double t0 = f[9], t1 = f[kHypRec + 9], t2 = f[2 * kHypRec + 9], t3 = f[3 * kHypRec + 9];
#pragma unroll
        for (int k = 6; k < 9; ++k) {
            asm volatile("" : "+v"(F0[k]));
            asm volatile("" : "+v"(F1[k]));
            asm volatile("" : "+v"(F2[k]));
            asm volatile("" : "+v"(F3[k]));
        }
        const uint32_t ok4 = __builtin_amdgcn_readfirstlane(okp[g]);
        // state bytes of the four records: 1 (approximate) and 3 (exact) are counted; 2 waits for the exact solve that
        // follows the counting and goes straight to the selection (count "infinite"); 0 is a rejected sample
        unsigned alive = 0, wait = 0;
#pragma unroll
        for (int k = 0; k < kCntSlots; ++k) {
            const unsigned st = (ok4 >> (8 * k)) & 0xffu;
            alive |= ((st == kPsApprox || st == kPsExact) && h0 + k < H) ? (1u << k) : 0u;
            wait |= (st == kPsNeedExact && h0 + k < H) ? (1u << k) : 0u;
        }
        int c0 = 0, c1 = 0, c2 = 0, c3 = 0;
        double2 pa0[PPL], pb0[PPL], pa1[PPL], pb1[PPL];
        auto load = [&](double2 (&pa)[PPL], double2 (&pb)[PPL], int blk) {
            const int nb = min(blk, nblk - 1) * BW;
#pragma unroll
            for (int u = 0; u < PPL; ++u) {
                pa[u] = L1[nb + u * 64];
                pb[u] = L2[nb + u * 64];
            }
        };
        auto process = [&](const double2 (&pa)[PPL], const double2 (&pb)[PPL], int blk) {
            double4 p[PPL];
#pragma unroll
            for (int u = 0; u < PPL; ++u)
                p[u] = make_double4(pa[u].x, pa[u].y, pb[u].x, pb[u].y);
            const int need = B - max(M - (blk + 1) * BW, 0);   // a slot whose count stays below this cannot reach B
            if (STATS)
                visits += (unsigned)__builtin_popcount(alive);
            if (alive & 1u) {
#pragma unroll
                for (int u = 0; u < PPL; ++u)
                    c0 += count_block_thr(F0, p[u], t0);
                if (c0 < need) alive &= ~1u;
            }
            if (alive & 2u) {
#pragma unroll
                for (int u = 0; u < PPL; ++u)
                    c1 += count_block_thr(F1, p[u], t1);
                if (c1 < need) alive &= ~2u;
            }
            if (alive & 4u) {
#pragma unroll
                for (int u = 0; u < PPL; ++u)
                    c2 += count_block_thr(F2, p[u], t2);
                if (c2 < need) alive &= ~4u;
            }
            if (alive & 8u) {
#pragma unroll
                for (int u = 0; u < PPL; ++u)
                    c3 += count_block_thr(F3, p[u], t3);
                if (c3 < need) alive &= ~8u;
            }
        };
        load(pa0, pb0, 0);
        for (int blk = 0; blk < nblk && alive; blk += 2) {
            load(pa1, pb1, blk + 1);
            process(pa0, pb0, blk);
            if (!(blk + 1 < nblk && alive))
                break;
            load(pa0, pb0, blk + 2);
            process(pa1, pb1, blk + 1);
        }
        // a slot that is still alive has seen every point: its upper-bound count is final.  Its lower bound: the same
        // count for an exact record (threshold == thr), one more pass against thr - band for an approximate one.
        auto lower = [&](const double (&F)[9], double tu, int cu) -> int {
            if (tu == thr)
                return cu;
            // thr - band, rounded down: band' = tu - thr is at most one rounding below the certified band
            const double tl = thr - (tu - thr) * (1.0 + 1e-9) - 1e-15 * thr;
            int cl = 0;
            for (int blk = 0; blk < nblk; ++blk) {
#pragma unroll
                for (int u = 0; u < PPL; ++u) {
                    const double2 a = L1[blk * BW + u * 64], c = L2[blk * BW + u * 64];
                    cl += count_block_thr(F, make_double4(a.x, a.y, c.x, c.y), tl);
                }
            }
            if (STATS)
                visits += (unsigned)nblk;
            return cl;
        };
        const int v0 = (alive & 1u) ? c0 : (wait & 1u) ? 0x7fffffff : -1, v1 = (alive & 2u) ? c1 : (wait & 2u) ? 0x7fffffff : -1;
        const int v2 = (alive & 4u) ? c2 : (wait & 4u) ? 0x7fffffff : -1, v3 = (alive & 8u) ? c3 : (wait & 8u) ? 0x7fffffff : -1;
        int l0 = -1, l1 = -1, l2 = -1, l3 = -1;
        if (alive & 1u) l0 = lower(F0, t0, c0);
        if (alive & 2u) l1 = lower(F1, t1, c1);
        if (alive & 4u) l2 = lower(F2, t2, c2);
        if (alive & 8u) l3 = lower(F3, t3, c3);
        if (lane < kCntSlots)
            cntp[h0 + lane] = lane == 0 ? v0 : lane == 1 ? v1 : lane == 2 ? v2 : v3;
        const int cm = __builtin_amdgcn_readfirstlane(max(max(l0, l1), max(l2, l3)));
        if (cm > B) {
            B = cm;
            if (lane == 0) {
                atomicMax(&s_bound, cm);
                __hip_atomic_fetch_max(gbound, cm, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        B = max(B, __builtin_amdgcn_readfirstlane(gb));
    }
    if (STATS && lane == 0 && b.stats)
        atomicAdd(&b.stats[2], visits * (unsigned long long)BW);
}

// Single-precision form of ransac_count2_kernel for pairs in mode 1: the record holds F~ as 9 floats and the two counting
// thresholds, already widened by the pre-screen's band AND by the bound on the binary32 evaluation error (prescreen.hpp),
// the points are rounded to binary32 in LDS (16 bytes per point: one ds_read_b128).  v_fma_f32 issues at twice the rate of
// v_fma_f64, the residual is the same nine instructions.  Exact records do not occur in these pairs before the selection
// (hypotheses without a certificate wait for the exact solve with an "infinite" count).
typedef __attribute__((address_space(4))) float CFloat;

// Packed single precision.  A plain v_fma_f32 issues at the rate of v_fma_f64 on this part (16 lanes per clock and SIMD);
// the fp32 vector peak is v_pk_fma_f32's: two FMAs per lane and instruction.  Each lane therefore carries TWO points per
// packed register pair -- the LDS block is laid out so that (x2 of point l, x2 of point l + 64) arrive adjacent -- and F comes
// straight out of the scalar registers the record was loaded into: op_sel picks the low or the high float of an aligned SGPR
// pair for both halves, so nothing is duplicated or moved.  Written as inline asm: left to itself hipcc's SLP vectoriser does
// emit v_pk_fma_f32, but with F splatted into vector register pairs first (200 v_mov per 192 packed FMAs; this file is built
// with -fno-slp-vectorize).
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(4))) unsigned long long CU64;

// d = a * s.lo + c  /  d = a * s.hi + c  (both halves of a, c; s = an aligned scalar register pair holding two floats)
__device__ __forceinline__ f32x2 pk_fma_slo(f32x2 a, unsigned long long s, f32x2 c)
{
    f32x2 d;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=v"(d) : "v"(a), "s"(s), "v"(c));
    return d;
}
__device__ __forceinline__ f32x2 pk_fma_shi(f32x2 a, unsigned long long s, f32x2 c)
{
    f32x2 d;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "=v"(d) : "v"(a), "s"(s), "v"(c));
    return d;
}
__device__ __forceinline__ f32x2 pk_fma_vv(f32x2 a, f32x2 b, f32x2 c)
{
    f32x2 d;
    asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}

struct Rec32 {                 // one single-precision record as it sits in scalar registers
    unsigned long long q01, q23, q45;   // (F0, F1), (F2, F3), (F4, F5)
    f32x2 f6, f7, f8;                   // (F6, F6), (F7, F7), (F8, F8): the inner addends, in vector registers
    float tu, tl;
};

// inliers among the 128 points a wavefront's lanes hold as two packed planes: A = (x1a, x1b, y1a, y1b), B = (x2a, x2b, y2a, y2b)
__device__ __forceinline__ int count_pair32(const Rec32 &r, const float4 &A, const float4 &B, float thr)
{
    const f32x2 X1 = {A.x, A.y}, Y1 = {A.z, A.w}, X2 = {B.x, B.y}, Y2 = {B.z, B.w};
    const f32x2 u0 = pk_fma_slo(X2, r.q01, pk_fma_shi(Y2, r.q23, r.f6));   // x2 F0 + (y2 F3 + F6)
    const f32x2 u1 = pk_fma_shi(X2, r.q01, pk_fma_slo(Y2, r.q45, r.f7));   // x2 F1 + (y2 F4 + F7)
    const f32x2 u2 = pk_fma_slo(X2, r.q23, pk_fma_shi(Y2, r.q45, r.f8));   // x2 F2 + (y2 F5 + F8)
    const f32x2 e = pk_fma_vv(u0, X1, pk_fma_vv(u1, Y1, u2));
    // NaN (padding lanes) compares false
    return __popcll(__ballot(__builtin_fabsf(e.x) < thr)) + __popcll(__ballot(__builtin_fabsf(e.y) < thr));
}

constexpr int kPilotHyp = 256;   // hypotheses of a pair the (vector, diagnostics) pilot counts in full
constexpr int kPilotMfmaHyp = 1024;   // ... and the matrix-core pilot (ransac_finish_mfma_kernel<., true>)
// phase: 0 = the PILOT (hypotheses [0, kPilotHyp) counted in full: their best lower bound is the pair's first bound, from
// which the dense phase derives how many points it has to look at); 2 = the FINISH (every hypothesis resumes behind the
// points[0, n1) the dense matrix-core phase has already counted: hyp_cnt holds that partial upper-bound count); 1 = everything
// in one go (no dense phase)
template <int CNT_THREADS, int PPL, int SLOTS, int phase, bool STATS = false>
__global__ __launch_bounds__(CNT_THREADS) void ransac_count32_kernel(BatchDev b, RunParams rp, int wg_per_pair)
{
    static_assert(SLOTS == 4 && PPL % 2 == 0, "four records per group; points come in packed pairs");
    extern __shared__ __attribute__((aligned(16))) double s_cpts[];
    __shared__ int s_bound;
    const int pair = blockIdx.y, tid = threadIdx.x, lane = tid & 63;
    const int M = min(b.M[pair], b.max_kp);
    if (M < 8 || b.mode[pair] != 1)
        return;
    const int Hall = rp.num_hypotheses;
    const int H = phase == 0 ? min(Hall, kPilotHyp) : Hall;
    const size_t Hp = (size_t)b.max_groups * kHypPerBlock;
    constexpr int BW = 64 * PPL;                  // points per block
    constexpr int NP = PPL / 2;                   // packed pairs per lane and block
    const int nblk = (M + BW - 1) / BW;
    // points [0, n1) were counted by the dense phase (n1 = dense_points(M, the pilot's bound), a multiple of 32): the finish
    // starts in the block that holds point n1 and blanks the points before it
    const int n1 = phase == 2 ? b.dense_n1[pair] : 0;
    const int blk0 = n1 / BW;
    // phase 2 works through the pair's list of hypotheses the dense phase left alive, four list entries per group
    const uint32_t *clist = b.clist + (size_t)pair * Hp;
    const int n_list = phase == 2 ? b.ccount[pair] : 0;
    // LDS: [block][pair u][plane A / B][lane] float4; point i = blk * BW + u * 128 + half * 64 + lane sits in half `half`
    float *s_f = reinterpret_cast<float *>(s_cpts);
    {
        const double4 *src = reinterpret_cast<const double4 *>(b.pts + (size_t)pair * b.max_kp * 4);
        const float qnan = __builtin_nanf("");
        for (int i = tid; i < nblk * BW; i += CNT_THREADS) {
            float x1 = qnan, y1 = qnan, x2 = qnan, y2 = qnan;
            if (i < M) {
                const double4 p = src[i];
                x1 = (float)p.x; y1 = (float)p.y; x2 = (float)p.z; y2 = (float)p.w;
            }
            const int blk = i / BW, w = i - blk * BW, u = w >> 7, half = (w >> 6) & 1, l = w & 63;
            float *A = s_f + ((((size_t)blk * NP + u) * 2 + 0) * 64 + l) * 4;
            float *Bp = s_f + ((((size_t)blk * NP + u) * 2 + 1) * 64 + l) * 4;
            A[half] = x1; A[2 + half] = y1;
            Bp[half] = x2; Bp[2 + half] = y2;
        }
    }
    int *gbound = b.bound + pair;
    if (tid == 0)
        s_bound = __hip_atomic_load(gbound, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const float *Fp = b.hyp_r32 + (size_t)pair * Hp * kHypRec32;   // 48-byte single-precision records
    const uint32_t *okp = reinterpret_cast<const uint32_t *>(b.hyp_okf + (size_t)pair * Hp);
    int32_t *cntp = b.hyp_cnt + (size_t)pair * Hp;
    const float4 *L = reinterpret_cast<const float4 *>(s_f) + lane;   // plane stride 64, pair stride 128, block stride 128 NP
    const int n_groups = phase == 2 ? (n_list + SLOTS - 1) / SLOTS : (H + SLOTS - 1) / SLOTS;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n_waves = wg_per_pair * (CNT_THREADS / 64);
    int B = 0;
    unsigned long long visits = 0;
    for (int g = blockIdx.x * (CNT_THREADS / 64) + wave; g < n_groups; g += n_waves) {
        const int h0 = g * SLOTS;
        const int gb = __hip_atomic_load(gbound, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        B = __builtin_amdgcn_readfirstlane(max(B, *(volatile int *)&s_bound));
        // the four hypotheses of the group: consecutive ones, or (finish) four entries of the list
        int hq[SLOTS];
        if (phase == 2) {
            const uint32_t mine = lane < SLOTS && h0 + lane < n_list ? clist[h0 + lane] : 0u;
#pragma unroll
            for (int q = 0; q < SLOTS; ++q)
                hq[q] = __builtin_amdgcn_readlane((int)mine, q);
        } else {
#pragma unroll
            for (int q = 0; q < SLOTS; ++q)
                hq[q] = h0 + q;
        }
        Rec32 R[SLOTS];
#pragma unroll
        for (int q = 0; q < SLOTS; ++q) {
            // (one base + constant offsets when the four records are neighbours: the loads coalesce)
            const CU64 *f = phase == 2 ? (const CU64 *)(uintptr_t)(Fp + (size_t)hq[q] * kHypRec32)
                                       : (const CU64 *)(uintptr_t)(Fp + (size_t)h0 * kHypRec32) + q * (kHypRec32 / 2);
            R[q].q01 = f[0];
            R[q].q23 = f[1];
            R[q].q45 = f[2];
            const unsigned long long q67 = f[3], q8u = f[4], ql = f[5];
            const float f6 = __uint_as_float((unsigned)q67), f7 = __uint_as_float((unsigned)(q67 >> 32));
            const float f8 = __uint_as_float((unsigned)q8u);
            R[q].f6 = f32x2{f6, f6};
            R[q].f7 = f32x2{f7, f7};
            R[q].f8 = f32x2{f8, f8};
            R[q].tu = __uint_as_float((unsigned)(q8u >> 32));
            R[q].tl = __uint_as_float((unsigned)ql);
        }
        unsigned alive = 0, wait = 0;
        int c[SLOTS];
#pragma unroll
        for (int q = 0; q < SLOTS; ++q)
            c[q] = 0;
        if (phase == 2) {
            // resume: every listed hypothesis is an approximate record; the dense phase left its upper-bound count over
            // points [0, n1) in hyp_cnt.  A slot that cannot reach the bound (it may have risen since the list was
            // written) even if every remaining point were an inlier is dead on arrival
            const int have = lane < SLOTS && h0 + lane < n_list ? cntp[clist[h0 + lane]] : 0;
#pragma unroll
            for (int q = 0; q < SLOTS; ++q) {
                c[q] = __builtin_amdgcn_readlane(have, q);
                if (h0 + q < n_list && !(c[q] + (M - n1) < B))
                    alive |= 1u << q;
            }
        } else {
            const uint32_t ok4 = __builtin_amdgcn_readfirstlane(okp[g]);
#pragma unroll
            for (int k = 0; k < SLOTS; ++k) {
                const unsigned st = (ok4 >> (8 * k)) & 0xffu;
                alive |= (st == kPsApprox && h0 + k < H) ? (1u << k) : 0u;
                wait |= (st == kPsNeedExact && h0 + k < H) ? (1u << k) : 0u;
            }
        }
        float4 pa[PPL], pb[PPL];   // [2 u] = plane A, [2 u + 1] = plane B of packed pair u
        auto load = [&](float4 (&p)[PPL], int blk) {
            const int nb = min(blk, nblk - 1) * (128 * NP);
#pragma unroll
            for (int u = 0; u < PPL; ++u)
                p[u] = L[nb + u * 64];
            if (blk == blk0 && n1 > blk0 * BW) {
                // the dense phase has already counted the points of this block below n1: blank them (NaN is no inlier).
                // p[2 u] / p[2 u + 1] hold points blk BW + 128 u + lane (x, z components) and + 64 (y, w components)
                const float qnan = __builtin_nanf("");
#pragma unroll
                for (int u = 0; u < NP; ++u) {
                    const int i0 = blk * BW + u * 128 + lane;
                    if (i0 < n1) { p[2 * u].x = qnan; p[2 * u + 1].x = qnan; }
                    if (i0 + 64 < n1) { p[2 * u].y = qnan; p[2 * u + 1].y = qnan; }
                }
            }
        };
        auto process = [&](const float4 (&p)[PPL], int blk) {
            const int need = B - max(M - (blk + 1) * BW, 0);   // a slot whose count stays below this cannot reach B
            if (STATS)
                visits += (unsigned)__builtin_popcount(alive);
            if (alive == (1u << SLOTS) - 1u) {
                // all four alive (the usual state until they die together): one straight-line stretch, so that the
                // scheduler interleaves the slots' independent FMA chains -- slot by slot behind uniform branches a
                // wavefront has two short dependent chains in flight and the SIMD waits on latencies
                int add[SLOTS];
#pragma unroll
                for (int q = 0; q < SLOTS; ++q) {
                    add[q] = 0;
#pragma unroll
                    for (int u = 0; u < NP; ++u)
                        add[q] += count_pair32(R[q], p[2 * u], p[2 * u + 1], R[q].tu);
                }
#pragma unroll
                for (int q = 0; q < SLOTS; ++q) {
                    c[q] += add[q];
                    alive &= (c[q] < need) ? ~(1u << q) : ~0u;
                }
            } else {
#pragma unroll
                for (int q = 0; q < SLOTS; ++q) {
                    if (alive & (1u << q)) {
#pragma unroll
                        for (int u = 0; u < NP; ++u)
                            c[q] += count_pair32(R[q], p[2 * u], p[2 * u + 1], R[q].tu);
                        if (c[q] < need) alive &= ~(1u << q);
                    }
                }
            }
        };
        if (alive)
            load(pa, blk0);
        for (int blk = blk0; blk < nblk && alive; blk += 2) {
            load(pb, blk + 1);
            process(pa, blk);
            if (!(blk + 1 < nblk && alive))
                break;
            load(pa, blk + 2);
            process(pb, blk + 1);
        }
        // lower bounds of the slots that saw every point: one more pass against the lower threshold
        int v[SLOTS], lo[SLOTS];
        int cm = -1;
#pragma unroll
        for (int q = 0; q < SLOTS; ++q) {
            const bool al = (alive >> q) & 1u;
            v[q] = al ? c[q] : ((wait >> q) & 1u) ? 0x7fffffff : -1;
            lo[q] = -1;
            if (al) {
                int cl = 0;
                for (int blk = 0; blk < nblk; ++blk) {
#pragma unroll
                    for (int u = 0; u < NP; ++u)
                        cl += count_pair32(R[q], L[blk * (128 * NP) + (2 * u) * 64], L[blk * (128 * NP) + (2 * u + 1) * 64], R[q].tl);
                }
                if (STATS)
                    visits += (unsigned)nblk;
                lo[q] = cl;
            }
            cm = max(cm, lo[q]);
        }
        if (lane < SLOTS && (phase != 2 || h0 + lane < n_list)) {
            int mine = v[0], where = hq[0];
#pragma unroll
            for (int q = 1; q < SLOTS; ++q) {
                mine = lane == q ? v[q] : mine;
                where = lane == q ? hq[q] : where;
            }
            cntp[where] = mine;
        }
        cm = __builtin_amdgcn_readfirstlane(cm);
        if (cm > B) {
            B = cm;
            if (lane == 0) {
                atomicMax(&s_bound, cm);
                __hip_atomic_fetch_max(gbound, cm, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        B = max(B, __builtin_amdgcn_readfirstlane(gb));
    }
    if (STATS && lane == 0 && b.stats)
        atomicAdd(&b.stats[3], visits * (unsigned long long)BW);   // executed single-precision evaluations
}

// ---- dense counting on the matrix cores --------------------------------------------------------------------------------
// The residuals of a block of points under a block of hypotheses are one GEMM: r[i][h] = Phi(point i) . F~(h) with the nine
// monomials Phi = (x2 x1, x2 y1, x2, y2 x1, y2 y1, y2, x1, y1, 1).  ransac_count32_kernel spends 84 vector + scalar
// instructions on 1024 evaluations and is bound by instruction issue; the matrix cores do the products of a 32 x 32 tile in
// a handful.  In exact binary32 (v_mfma_f32_32x32x2_f32, five per tile, 320 clocks: the fp32 vector rate) the phase was no
// faster than the vector kernel; here the product runs in SPLIT bf16: every binary32 operand is hi + lo with
// hi = bf16(x), lo = bf16(x - hi) (sixteen significant bits), and Phi_k F_k ~ hi hi + hi lo + lo hi: 27 products of bf16
// numbers (exact in binary32) summed in binary32 by two v_mfma_f32_32x32x16_bf16 (K = 32, five slots zero, 64 clocks).
// Error of the value against the exact sum on the binary32 inputs, per term |Phi_k F_k|: the two representation errors
// 2 x 2^-16 (|x - hi| <= 2^-8 |x|, |x - hi - lo| <= 2^-8 |x - hi|), the dropped lo lo 2^-16, the <= 32 additions of the
// accumulation 2^-23 each (whatever their order and rounding mode: 2^-18), the monomial and the three input roundings
// 4 x 2^-24 -- together < 3.3 x 2^-16 = 0.82 x 2^-14 of T = sum |p2_j| |F_jk| |p1_k|; the phase counts against
// tu' = tu + 2^-14 T(box) (tu already carries the band and the binary32 bound e32), so its count is an UPPER bound of the
// exact count like every other.  A tile cannot drop single hypotheses, so the phase takes NO exit tests: it counts points
// [0, n1) for every approximate record,
//     n1 = dense_points(M, B0) = the multiple of 32 that covers M - B0 + 32 points, clamped to [0, M rounded down],
// B0 = the bound the PILOT (the first kPilotHyp hypotheses counted in full by ransac_count32_kernel) has established: a
// hypothesis cannot be dropped before M - B points have been seen, so nothing is wasted except on the few per cent that go on.
// ransac_count32_kernel then resumes behind n1 for the hypotheses that can still reach the bound.
// Wavefront = 2 x 32 hypotheses (the B operands: 16 VGPRs, built once) x all point tiles (A operands: two ds_read_b128 per
// tile); the accumulator tile has the hypothesis on the lane and 16 points in the registers.
constexpr int kDenseThreads = 512;   // dense phase: 8 wavefronts x 64 hypotheses share one staged chunk of points; with 672-point
                                     // chunks two workgroups fit a CU = 4 wavefronts per SIMD (what the 102 registers allow).
                                     // A/B on the bench batch (tools/dense_ab.py, profiles/r04_dense_ab.json), same box: 256 x 8
                                     // threads x batches with 768-point chunks (2 per SIMD) 1.36 ms, 512 x 4 / 768 (one workgroup
                                     // per CU) 1.44, 512 x 4 / 640 1.22, 512 x 4 / 672 1.19; the three-stage software pipeline
                                     // 1.43 / 1.97 / 1.58 at 256 / 384 / 512 threads
constexpr int kFinishThreads = 256;  // finish: 4 wavefronts x 64 list entries
constexpr int kDenseChunk = 768;     // points staged per pass (multiple of 32): 48 KB of split monomials
constexpr int kDenseBatches = 4;     // batches of kDenseThreads hypotheses a workgroup takes over the points it has staged
constexpr int kDenseChunkD = 672;    // points the DENSE phase stages per pass: 42 KB + 8 x 4 KB of windows + the list = 79.9 KB
constexpr int kDenseWin = 256;       // uint4 words of a wavefront's LDS window: 64 records x 48 B in, 64 operands x 64 B out
typedef float v16f __attribute__((ext_vector_type(16)));
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));

constexpr int kDenseMargin = 64;   // points beyond M - B0 the dense phase covers for every hypothesis (multiple of 32)
__device__ __forceinline__ int dense_points(int M, int B0, int margin)
{
    const int want = ((M - B0 + margin + 31) / 32) * 32;
    return max(0, min(want, (M / 32) * 32));
}

// K slot s = 0 .. 31 of the two MFMAs carries term (k, part) = (s % 9, s / 9) for s < 27: part 0 = hi hi, 1 = hi(Phi) lo(F),
// 2 = lo(Phi) hi(F); slot s sits in MFMA j = s / 16, lane half (s / 8) & 1, element s & 7 -- for BOTH operands, so the sum does
// not depend on how the instruction numbers its k (dense_operands_pk below builds the four 128-bit words [j][lane half]).
// Counting without compares: for a pair of accumulators a, ind = clamp((T2 - a * a) * 2^100) is 1 when a^2 < T2 and 0 when
// a^2 >= T2 (the product with 2^100 is >= 1 as soon as the difference is one ulp of T2 >= 2^-79), so the per-lane count is a
// float sum of indicators: v_pk_mul_f32, v_pk_fma_f32 with the clamp bit, v_pk_add_f32 -- three packed instructions per TWO
// accumulators, no scalar registers, no wait states (v_cmp + v_addc is two per accumulator plus an s_nop each).  T2 =
// tu'^2 (1 + 2^-21): every |a| < tu' is counted whatever the rounding of a * a; an |a| a hair above tu' may be counted too,
// which an UPPER bound of the count can afford.  Counts stay below 2^24: exact in binary32.
__device__ __forceinline__ f32x2 pk_mul(f32x2 a, f32x2 b)
{
    f32x2 d;
    asm("v_pk_mul_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ f32x2 pk_ind(f32x2 sq, f32x2 negH, f32x2 t2H)   // clamp(sq * (-2^100) + T2 * 2^100)
{
    f32x2 d;
    asm("v_pk_fma_f32 %0, %1, %2, %3 clamp" : "=v"(d) : "v"(sq), "v"(negH), "v"(t2H));
    return d;
}
__device__ __forceinline__ f32x2 pk_add(f32x2 a, f32x2 b)
{
    f32x2 d;
    asm("v_pk_add_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ void dense_count(const v16f &acc, f32x2 negH, f32x2 t2H, f32x2 &cnt)
{
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
        const f32x2 a = {acc[r], acc[r + 1]};
        cnt = pk_add(cnt, pk_ind(pk_mul(a, a), negH, t2H));
    }
}

// The thresholds of the matrix-core counting, from a single-precision record (F~, tu, tl) and the pair's box: T = [X2 Y2 1]
// |F~| [X1 Y1 1]^T with every rounding upwards, tu' = tu + 2^-14 T, tl' = tl - 2^-14 T, and the scaled
// squares the compare-free indicator takes.  One definition for the dense phase, the finish and the diagnostics probe.
__device__ __forceinline__ float dense_T(const float (&Ff)[9], float X1, float Y1, float X2, float Y2)
{
    const float t0 = fmaf(X2, fabsf(Ff[0]), fmaf(Y2, fabsf(Ff[3]), fabsf(Ff[6])));
    const float t1 = fmaf(X2, fabsf(Ff[1]), fmaf(Y2, fabsf(Ff[4]), fabsf(Ff[7])));
    const float t2 = fmaf(X2, fabsf(Ff[2]), fmaf(Y2, fabsf(Ff[5]), fabsf(Ff[8])));
    return fmaf(t0, X1, fmaf(t1, Y1, t2)) * (1.f + 0x1p-18f);
}
// upper: every |a| < tu' must be counted whatever the rounding of a * a: T2 = tu'^2 (1 + 2^-21)
__device__ __forceinline__ float dense_t2_upper(float tu_rec, float T, bool on)
{
    const float tu = on ? (tu_rec + 0x1p-14f * T) * (1.f + 0x1p-22f) : 0.f;
    return (tu * tu) * (1.f + 0x1p-21f) * 0x1p100f;
}
// lower: the threshold only shrinks and no |a| >= tl' may be counted: T2 = tl'^2 (1 - 2^-21); a non-positive tl' counts nothing
__device__ __forceinline__ float dense_t2_lower(float tl_rec, float T, bool on, bool &lpos)
{
    const float tl = on ? (tl_rec - 0x1p-14f * T) * (1.f - 0x1p-22f) : 0.f;
    lpos = tl > 0.f;
    return lpos ? (tl * tl) * (1.f - 0x1p-21f) * 0x1p100f : 0.f;
}

// amdgpu_waves_per_eu(4, 8): with the default register budget of a 256-thread kernel hipcc puts the MFMA results into
// accumulation registers and reads every one back with v_accvgpr_read before it can be used (16 more vector
// instructions per tile).  A wavefront carries TWO blocks of 32 hypotheses: their MFMA chains are independent and share
// every A operand read.
// Grid (ceil(G / kDenseBatches), P), G = the pair's batches of 256 hypotheses: a workgroup stages the pair's points ONCE and
// takes kDenseBatches batches over them, and the records of batch g + 1 are in flight while batch g is multiplied -- with
// one batch per workgroup the kernel spent 60 % of its time waiting for its records (1 GB per 256 pairs behind a full
// memory latency per wavefront: 2.04 ms, 0.81 without the loads).
// The 64 records of a wavefront's batch are one contiguous 5 KB span: it is read as five fully coalesced 128-bit loads (lane l
// takes bytes [1024 j + 16 l, + 16)) and transposed through a 5 KB LDS window of the wavefront -- a lane reading "its" record
// directly touches 64 records x 80 bytes with 16-byte pieces, 44 cache-line requests per instruction (1.83 ms -> see DESIGN).
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// bf16 pair of two binary32 numbers, round to nearest even (v_cvt_pk_bf16_f32): first argument in the low half
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t bf16_pk(float lo, float hi)
{
    const bf16x2 v = {(__bf16)lo, (__bf16)hi};
    return __builtin_bit_cast(uint32_t, v);
}
__device__ __forceinline__ float mul_legacy(float a, float b)   // a * b with 0 * anything = 0 (also 0 * inf, 0 * NaN)
{
    float d;
    asm("v_mul_legacy_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
// The four 128-bit MFMA operands [j][lane half] of nine binary32 terms x (split as hi = bf16(x), lo = bf16(x - hi)) in the K slot
// order of dense_slot: lo_part = 1 (a hypothesis: slots 9..17 carry lo) or 2 (a point: slots 18..26 carry lo).  Same bits as
// dense_operands on bf16_split parts; 28-32 instructions instead of ~110 (the hardware conversion packs two values at a time
// in the pair order the slots need).
template <int LO_PART>
__device__ __forceinline__ void dense_operands_pk(const float (&x)[9], uint4 (&op)[2][2])
{
    const uint32_t P01 = bf16_pk(x[0], x[1]), P23 = bf16_pk(x[2], x[3]), P45 = bf16_pk(x[4], x[5]), P67 = bf16_pk(x[6], x[7]);
    const uint32_t P8z = bf16_pk(x[8], 0.f);
    const uint32_t pk[5] = {P01, P23, P45, P67, P8z};
    float L[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        const uint32_t w = pk[k >> 1];
        const float h = __uint_as_float((k & 1) ? (w & 0xffff0000u) : (w << 16));
        L[k] = x[k] - h;   // exact
    }
    if (LO_PART == 1) {
        // slots 0..8 hi | 9..17 lo | 18..26 hi | 27..31 zero
        op[0][0] = make_uint4(P01, P23, P45, P67);
        op[0][1] = make_uint4(bf16_pk(x[8], L[0]), bf16_pk(L[1], L[2]), bf16_pk(L[3], L[4]), bf16_pk(L[5], L[6]));
        op[1][0] = make_uint4(bf16_pk(L[7], L[8]), P01, P23, P45);
        op[1][1] = make_uint4(P67, P8z, 0u, 0u);
    } else {
        // slots 0..8 hi | 9..17 hi | 18..26 lo | 27..31 zero
        op[0][0] = make_uint4(P01, P23, P45, P67);
        op[0][1] = make_uint4(bf16_pk(x[8], x[0]), bf16_pk(x[1], x[2]), bf16_pk(x[3], x[4]), bf16_pk(x[5], x[6]));
        op[1][0] = make_uint4(bf16_pk(x[7], x[8]), bf16_pk(L[0], L[1]), bf16_pk(L[2], L[3]), bf16_pk(L[4], L[5]));
        op[1][1] = make_uint4(bf16_pk(L[6], L[7]), bf16_pk(L[8], 0.f), 0u, 0u);
    }
}

// NORMALISED counting (round 4).  Round 3 counted with three packed vector instructions per two accumulators (v_pk_mul,
// v_pk_fma clamp, v_pk_add: a compare-free indicator against the hypothesis' own threshold) -- 48 packed instructions per tile
// pair at 6.6 clocks each beside 4 MFMAs, and packed binary32 instructions do not overlap with the matrix pipe on this part
// (319 clocks per tile pair against 130 for the MFMAs alone, profiles/r03_pk_mfma_microbench.txt): the kernel was a vector
// kernel with an MFMA inside.  Here the hypothesis' threshold is folded into its B operand: F' = F~ * s with
// s = 2 / tu' rounded down, so that the accumulator is q = s * a and  |a| < tu'  ==>  |q| < 2  -- and |q| < 2 is ONE BIT of the
// binary32 pattern: the biased exponent of q is <= 127 exactly when bit 30 is clear (zero, denormals included; infinities
// and NaNs have it set).  v_alignbit_b32 coll, coll, q, 30 shifts the two top bits (sign, bit 30) of an accumulator into a
// collection word: ONE full-rate integer instruction per accumulator, sixteen accumulators fill the word, and
// popcount(coll & 0x55555555) is the number of accumulators NOT below the threshold: 18 plain vector instructions per 32 x 32
// tile, which issue beside the MFMAs.  Error budget of the scaling: F'_k = fl(F~_k s) adds 2^-24 s T to
// the split-bf16 term 0.82 * 2^-14 s T: still below 2^-14 s T, and s tu' <= 2 (1 - 2^-22)(1 + 2^-24) < 2.
__device__ __forceinline__ float dense_tu(float tu_rec, float T) { return (tu_rec + 0x1p-14f * T) * (1.f + 0x1p-22f); }
// scale of a record's B operand: 2 / tu' rounded down; 0 (every point counted: still an upper bound) for a threshold too
// small to invert or a record that is not counted at all
__device__ __forceinline__ float dense_scale(float tu_rec, float T, bool on)
{
    const float tup = dense_tu(tu_rec, T);
    // v_rcp_f32 is good to one ulp: 2 rcp(tu') (1 - 2^-21) <= (2 / tu')(1 + 2^-23)(1 - 2^-21)(1 + 2^-24) < 2 / tu'
    return (on && tup > 0x1p-60f) ? (2.f * __builtin_amdgcn_rcpf(tup)) * (1.f - 0x1p-21f) : 0.f;
}
__device__ __forceinline__ uint32_t dense_collect(const v16f &acc)
{
    uint32_t c = 0u;
#pragma unroll
    for (int r = 0; r < 16; ++r)
        c = __builtin_amdgcn_alignbit(c, __float_as_uint(acc[r]), 30);   // (c << 2) | (sign, bit 30) of the accumulator
    return c;
}

template <bool STATS, int THREADS, int BATCHES, int CHUNK = kDenseChunkD>
__global__ __launch_bounds__(THREADS) __attribute__((amdgpu_waves_per_eu(2, 4))) void ransac_count_mfma_kernel(BatchDev b, RunParams rp, int margin)
{
    // LDS: the split monomials of a chunk of points as MFMA operands, [tile of 32 points][j][lane half][point] x 16 bytes:
    // the lanes of a wavefront read consecutive 16-byte words (no bank conflicts), two ds_read_b128 per tile; behind them one
    // 4 KB window per wavefront: the 64 records of its batch come in as one coalesced 3 KB span, lane l takes record l out,
    // builds that hypothesis' four operand words ONCE (round 3: every lane built the operands of two hypotheses and kept
    // half of each) and puts them back as [block][j][lane half][hypothesis] for the lanes that feed them to the MFMAs
    extern __shared__ __attribute__((aligned(16))) double s_cpts[];
    __shared__ uint16_t s_list[BATCHES * THREADS];   // the workgroup's survivors, relative to its first hypothesis
    __shared__ int s_nlist, s_base;
    static_assert(BATCHES * THREADS <= 65536, "16-bit list entries");
    const int pair = blockIdx.y, tid = threadIdx.x, lane = tid & 63, w = tid >> 6, col = lane & 31, half = lane >> 5;
    const int M = min(b.M[pair], b.max_kp);
    if (M < 8 || b.mode[pair] != 1)
        return;
    if (tid == 0)
        s_nlist = 0;   // (the first barrier of the chunk loop orders it)
    const int B0 = b.bound[pair];   // final since the pilot launch has completed
    const int n1 = dense_points(M, B0, margin);
    if (blockIdx.x == 0 && tid == 0)
        b.dense_n1[pair] = n1;
    const int H = rp.num_hypotheses;
    const int G = (H + THREADS - 1) / THREADS;
    const int g0 = blockIdx.x * BATCHES, g1 = min(g0 + BATCHES, G);
    if (g0 >= G)
        return;
    const size_t Hp = (size_t)b.max_groups * kHypPerBlock;
    uint4 *s_op = reinterpret_cast<uint4 *>(s_cpts);
    // the pair's largest coordinates, rounded up to binary32
    const PairBox bx = load_box(b, pair);
    const float X1 = (float)fmax(dabs(bx.x1lo), dabs(bx.x1hi)) * (1.f + 0x1p-22f);
    const float Y1 = (float)fmax(dabs(bx.y1lo), dabs(bx.y1hi)) * (1.f + 0x1p-22f);
    const float X2 = (float)fmax(dabs(bx.x2lo), dabs(bx.x2hi)) * (1.f + 0x1p-22f);
    const float Y2 = (float)fmax(dabs(bx.y2lo), dabs(bx.y2hi)) * (1.f + 0x1p-22f);
    const double4 *src = reinterpret_cast<const double4 *>(b.pts + (size_t)pair * b.max_kp * 4);
    uint4 *s_win = s_op + (size_t)CHUNK * 4 + w * kDenseWin;
    struct Raw {
        u32x4 p0, p1, p2;
        int st;
    };
    auto fetch = [&](int g, Raw &r) __attribute__((always_inline)) {
        const int hw = (g * (THREADS / 64) + w) * 64;
        const size_t rec0 = (size_t)pair * Hp + (hw + 64 <= (int)Hp ? hw : 0);
        const u32x4 *span = reinterpret_cast<const u32x4 *>(b.hyp_r32 + rec0 * kHypRec32) + lane;   // 64 x 48 B = 3 x 1 KB
        r.p0 = span[0];
        r.p1 = span[64];
        r.p2 = span[128];
        // (an UNCONDITIONAL load of a clamped index: a select on `hl < H` right here needs the loaded byte at once, and the wait
        // for it is a wait for the three record loads in front of it -- the prefetch then hides nothing: the dense phase spent
        // a memory round trip per batch that way, rounds 3 and 4a)
        r.st = (int)b.hyp_okf[(size_t)pair * Hp + min(hw + lane, (int)Hp - 1)];
    };
    int c0 = 0;
    do {   // the points go through LDS in chunks (one for nearly every pair); n1 = 0 still takes one pass (counts of zero)
        const int nc = max(0, min(CHUNK, n1 - c0));
        const bool first = c0 == 0, last = c0 + CHUNK >= n1;
        __syncthreads();
        for (int i = tid; i < nc; i += THREADS) {
            const double4 pd = src[c0 + i];
            const float x1 = (float)pd.x, y1 = (float)pd.y, x2 = (float)pd.z, y2 = (float)pd.w;
            const float ph[9] = {x2 * x1, x2 * y1, x2, y2 * x1, y2 * y1, y2, x1, y1, 1.f};
            uint4 o[2][2];
            dense_operands_pk<2>(ph, o);
            uint4 *q = s_op + (size_t)(i >> 5) * 128 + (i & 31);
            q[0] = o[0][0];     // j = 0, half 0
            q[32] = o[0][1];    // j = 0, half 1
            q[64] = o[1][0];    // j = 1, half 0
            q[96] = o[1][1];    // j = 1, half 1
        }
        Raw raw;
        fetch(g0, raw);
        __syncthreads();
        for (int g = g0; g < g1; ++g) {
            const int hw = (g * (THREADS / 64) + w) * 64;   // first hypothesis of the wavefront (two blocks of 32)
            // records in: the coalesced span, then lane l reads record l
            u32x4 *wr = reinterpret_cast<u32x4 *>(s_win) + lane;
            wr[0] = raw.p0;
            wr[64] = raw.p1;
            wr[128] = raw.p2;
            const int st_l = hw + lane < H ? raw.st : kPsInvalid;
            const float4 *rq = reinterpret_cast<const float4 *>(s_win + lane * 3);
            const float4 q0 = rq[0], q1 = rq[1], q2 = rq[2];
            if (g + 1 < g1)
                fetch(g + 1, raw);   // in flight during this batch's build and tiles
            {
                const float fr[9] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x};
                const bool on = st_l == kPsApprox;
                // F' = F~ * s,  s = 2 / tu' rounded down,  tu' = tu + 2^-14 T,  T = [X2 Y2 1] |F~| [X1 Y1 1]^T (rounded up)
                float sc = dense_scale(q2.y, dense_T(fr, X1, Y1, X2, Y2), on);
                float big = 0.f;
#pragma unroll
                for (int k = 0; k < 9; ++k)
                    big = fmaxf(big, fabsf(fr[k]));
                // an operand that is not finite would poison the accumulators: F' = 0 counts every point -- still an upper
                // bound (the comparison is false for a NaN, and v_mul_legacy gives 0 * x = 0 for every x)
                sc = (big * sc < 0x1p100f) ? sc : 0.f;
                float Fs[9];
#pragma unroll
                for (int k = 0; k < 9; ++k)
                    Fs[k] = mul_legacy(fr[k], sc);
                uint4 o[2][2];
                dense_operands_pk<1>(Fs, o);
                // operands out: [block][j][lane half][hypothesis of the block]
                uint4 *wo = s_win + (lane >> 5) * 128 + (lane & 31);
                wo[0] = o[0][0];
                wo[32] = o[0][1];
                wo[64] = o[1][0];
                wo[96] = o[1][1];
            }
            int h[2], st[2];
            v8bf Bop[2][2];
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                h[c] = hw + 32 * c + col;    // this lane's hypothesis of block c (both lane halves)
                st[c] = __shfl(st_l, 32 * c + col);
                const uint4 *ro = s_win + c * 128 + half * 32 + col;
                Bop[c][0] = __builtin_bit_cast(v8bf, ro[0]);
                Bop[c][1] = __builtin_bit_cast(v8bf, ro[64]);
            }
            if (hw >= H)
                continue;
            uint32_t nn[2] = {0u, 0u};   // accumulators NOT below the threshold
            // One tile at a time: two LDS reads, four MFMAs, the count (18 plain vector instructions per accumulator set).  A
            // three-stage software pipeline with two accumulator sets and sched_group_barrier was measured in round 4 and was
            // SLOWER (1.31 / 1.83 / 1.47 ms for 256 x 8 / 384 x 6 / 512 x 4 against 1.11: its 136-149 registers cost a wavefront
            // per SIMD, and the hardware already overlaps the pipes across wavefronts); the code is gone.
            const v16f zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
            const uint4 *qt = s_op + half * 32 + col;
            const int nt = nc >> 5;
            for (int t = 0; t < nt; ++t) {
                const uint4 *q = qt + (size_t)t * 128;
                const v8bf A0 = __builtin_bit_cast(v8bf, q[0]), A1 = __builtin_bit_cast(v8bf, q[64]);
                v16f a0, a1;
                a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A0, Bop[0][0], zero, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A0, Bop[1][0], zero, 0, 0, 0);
                a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A1, Bop[0][1], a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A1, Bop[1][1], a1, 0, 0, 0);
                // accumulator: column = lane & 31 (the hypothesis), 16 points in the registers
                nn[0] += (uint32_t)__builtin_popcount(dense_collect(a0) & 0x55555555u);
                nn[1] += (uint32_t)__builtin_popcount(dense_collect(a1) & 0x55555555u);
            }
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                int cnt = (nc >> 1) - (int)nn[c];   // this lane half saw 16 points of every tile
                cnt += __shfl_xor(cnt, 32);   // the two lane halves hold different points of the same hypothesis
                const size_t rec = (size_t)pair * Hp + h[c];
                const bool mine = half == 0 && h[c] < H;
                if (mine && st[c] == kPsApprox && !first)
                    cnt += b.hyp_cnt[rec];    // the earlier chunks' share (written by this lane)
                if (mine)
                    b.hyp_cnt[rec] = st[c] == kPsApprox ? cnt : st[c] == kPsNeedExact ? 0x7fffffff : -1;
                // hypotheses that can still reach the pilot's bound go on the pair's list for the finish
                const bool go = last && mine && st[c] == kPsApprox && !(cnt + (M - n1) < B0);
                // collected in LDS first: one atomic on the pair's counter per workgroup, not one per wavefront and block
                // (1568 returning atomics on the same address per pair cost more than the counting itself)
                const unsigned long long mm = __ballot(go);
                if (mm) {
                    int base = 0;
                    if (lane == 0)
                        base = atomicAdd(&s_nlist, __popcll(mm));
                    base = __builtin_amdgcn_readfirstlane(base);
                    if (go)
                        s_list[base + __popcll(mm & ((1ull << lane) - 1ull))] = (uint16_t)(h[c] - g0 * THREADS);
                }
                if (STATS && b.stats) {
                    const unsigned long long ma = __ballot(st[c] == kPsApprox && mine);
                    if (lane == 0 && ma)
                        atomicAdd(&b.stats[4], (unsigned long long)__popcll(ma) * (unsigned long long)nc);
                }
            }
        }
        c0 += CHUNK;
    } while (c0 < n1);
    __syncthreads();
    const int nl = s_nlist;
    if (nl > 0) {
        if (tid == 0)
            s_base = atomicAdd(&b.ccount[pair], nl);
        __syncthreads();
        uint32_t *dst = b.clist + (size_t)pair * Hp + s_base;
        for (int i = tid; i < nl; i += THREADS)
            dst[i] = (uint32_t)(g0 * THREADS) + s_list[i];
    }
}


// The pair's list in the order of the dense phase's counts, largest first (counting sort on the count, one workgroup per
// pair; the sorted list is the second half of clist).  The finish then meets the likely winners in its first batches.
constexpr int kSortThreads = 1024;   // (256 threads: 0.10 ms per 512 pairs, a chain of entry -> count round trips per thread)
__global__ __launch_bounds__(kSortThreads) void ransac_list_sort_kernel(BatchDev b)
{
    __shared__ int s_hist[kSortBins];
    __shared__ int s_scan[256];
    const int pair = blockIdx.x, tid = threadIdx.x;
    const int M = min(b.M[pair], b.max_kp);
    if (M < 8 || b.mode[pair] != 1)
        return;
    const int n = b.ccount[pair];
    if (n <= 0)
        return;
    const size_t Hp = (size_t)b.max_groups * kHypPerBlock;
    const uint32_t *in = b.clist + (size_t)pair * Hp;
    uint32_t *out = b.clist2 + (size_t)pair * Hp;
    const int32_t *cntp = b.hyp_cnt + (size_t)pair * Hp;
    for (int i = tid; i < kSortBins; i += kSortThreads)
        s_hist[i] = 0;
    __syncthreads();
    // (four entries per thread and trip: the loads of a trip are independent, so the entry -> count round trips overlap; one
    // entry per trip was 49 dependent pairs of round trips per thread and pass, most of this kernel's 0.14 ms)
    for (int e0 = tid; e0 < n; e0 += 4 * kSortThreads) {
        uint32_t hh[4];
        int cc[4];
#pragma unroll
        for (int k = 0; k < 4; ++k)
            hh[k] = in[min(e0 + kSortThreads * k, n - 1)];
#pragma unroll
        for (int k = 0; k < 4; ++k)
            cc[k] = cntp[hh[k]];
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (e0 + kSortThreads * k < n)
                atomicAdd(&s_hist[min(max(cc[k], 0), kSortBins - 1)], 1);
    }
    __syncthreads();
    // exclusive prefix over the bins in DESCENDING order of the count: thread t < 256 owns a run of consecutive bins from the top
    constexpr int per = (kSortBins + 255) / 256;
    int run = 0;
    if (tid < 256) {
        for (int k = 0; k < per; ++k) {
            const int bin = kSortBins - 1 - (tid * per + k);
            run += bin >= 0 ? s_hist[bin] : 0;
        }
        s_scan[tid] = run;
    }
    __syncthreads();
    for (int o = 1; o < 256; o <<= 1) {
        const int v = tid < 256 && tid >= o ? s_scan[tid - o] : 0;
        __syncthreads();
        if (tid < 256)
            s_scan[tid] += v;
        __syncthreads();
    }
    if (tid < 256) {
        int off = s_scan[tid] - run;
        for (int k = 0; k < per; ++k) {
            const int bin = kSortBins - 1 - (tid * per + k);
            if (bin >= 0) {
                const int c = s_hist[bin];
                s_hist[bin] = off;
                off += c;
            }
        }
    }
    __syncthreads();
    for (int e0 = tid; e0 < n; e0 += 4 * kSortThreads) {
        uint32_t hh[4];
        int cc[4];
#pragma unroll
        for (int k = 0; k < 4; ++k)
            hh[k] = in[min(e0 + kSortThreads * k, n - 1)];
#pragma unroll
        for (int k = 0; k < 4; ++k)
            cc[k] = cntp[hh[k]];
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (e0 + kSortThreads * k < n)
                out[atomicAdd(&s_hist[min(max(cc[k], 0), kSortBins - 1)], 1)] = hh[k];
    }
    __syncthreads();
    // every bin's offset has moved to the END of its run: s_hist[c] = entries with a count >= c.  Counts never exceed the
    // points the dense phase saw (<= M): the finish asks for c <= M only
    int32_t *cp = b.cpos + (size_t)pair * kSortBins;
    for (int i = tid; i <= min(M, kSortBins - 1); i += kSortThreads)
        cp[i] = s_hist[i];
}

// The FINISH on the matrix cores.  What the dense phase could not drop against the PILOT's bound (a quarter of the
// hypotheses) is worked off in batches of 256 list entries, best partial counts first, grid (P, batches): the same tile
// product, indicators against tl' = tl - 2^-14 T for every point (the lower-bound count L) and against tu' = tu + 2^-14 T for
// the points from n1 on (added to the dense phase's count in hyp_cnt: the upper bound U).  max L raises the pair's bound (one
// atomic per workgroup); a batch starts by dropping the entries that can no longer reach the bound as it stands, and leaves
// at once if none can -- after the first batches of a pair that is the usual case (the winner is among the largest
// partial counts).  An entry that is dropped can never matter: its upper bound is below a lower bound of another hypothesis.
// ransac_survivors_kernel compares U with the final bound as before.  The points past M in the last tile are staged as zero
// monomials: their residual is exactly 0, so each is counted once by every positive threshold and subtracted again.
// PILOT (round 4): the same kernel over the pair's FIRST kPilotMfmaHyp hypotheses instead of list entries -- both counts over
// every point, nothing assumed about them: the largest lower bound is the pair's first bound B0, from which the dense phase
// derives how many points it has to look at.  Round 3's pilot counted 256 hypotheses in packed binary32 vector code (0.13 ms per
// 512 pairs); four times as many on the matrix cores cost less and find a better B0 (the dense phase's work is M - B0 + 32
// points for every hypothesis).
template <bool STATS, bool PILOT = false>
__global__ __launch_bounds__(kFinishThreads) __attribute__((amdgpu_waves_per_eu(4, 8))) void ransac_finish_mfma_kernel(BatchDev b,
                                                                                                                     RunParams rp,
                                                                                                                     int batch0)
{
    extern __shared__ __attribute__((aligned(16))) double s_cpts[];
    __shared__ int s_lmax;
    const int pair = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6, col = lane & 31, half = lane >> 5;
    const int M = min(b.M[pair], b.max_kp);
    if (M < 8 || b.mode[pair] != 1)
        return;
    const int n_list = PILOT ? min(rp.num_hypotheses, (int)gridDim.y * kFinishThreads) : b.ccount[pair];   // (the pilot's grid says how many)
    const int e0 = (batch0 + blockIdx.y) * kFinishThreads;   // first list entry of the workgroup
    if (e0 >= n_list)
        return;
    if (tid == 0)
        s_lmax = -1;
    // (pilot: only the LOWER bounds matter -- the first bound of the pair --, so every point takes the branch in front of n1:
    // one indicator per accumulator instead of two, no exit tests: there is no bound yet to test against)
    const int n1 = PILOT ? ((M + 31) & ~31) : b.dense_n1[pair];
    const size_t Hp = (size_t)b.max_groups * kHypPerBlock;
    const uint32_t *clist = b.clist2 + (size_t)pair * Hp;   // sorted by ransac_list_sort_kernel
    const int Bnow = __hip_atomic_load(b.bound + pair, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int ew = e0 + w * 64;   // first list entry of the wavefront (two blocks of 32)
    int h[2], ucnt0[2];
    bool on[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const int e = ew + 32 * c + col;
        if (PILOT) {   // hypothesis e itself; only records the pre-screen certified (state: approximate F) take part
            h[c] = e < n_list ? e : 0;
            ucnt0[c] = 0;
            on[c] = e < n_list && b.hyp_okf[(size_t)pair * Hp + h[c]] == kPsApprox;
            continue;
        }
        h[c] = e < n_list ? (int)clist[e] : 0;
        ucnt0[c] = b.hyp_cnt[(size_t)pair * Hp + h[c]];   // the dense phase's count over [0, n1)
        // still able to reach the bound as it stands now?
        on[c] = e < n_list && !(ucnt0[c] + (M - n1) < Bnow);
    }
    if (!__syncthreads_or(on[0] || on[1]))
        return;   // (also orders s_lmax)
    bool wave_live = __ballot(on[0] || on[1]) != 0ull;
    bool wave_dead = false;   // left by the exit test: upper counts stay partial, no lower bound
    int ntile = 0;
    uint4 *s_op = reinterpret_cast<uint4 *>(s_cpts);
    const PairBox bx = load_box(b, pair);
    const float X1 = (float)fmax(dabs(bx.x1lo), dabs(bx.x1hi)) * (1.f + 0x1p-22f);
    const float Y1 = (float)fmax(dabs(bx.y1lo), dabs(bx.y1hi)) * (1.f + 0x1p-22f);
    const float X2 = (float)fmax(dabs(bx.x2lo), dabs(bx.x2hi)) * (1.f + 0x1p-22f);
    const float Y2 = (float)fmax(dabs(bx.y2lo), dabs(bx.y2hi)) * (1.f + 0x1p-22f);
    const f32x2 negH = {-0x1p100f, -0x1p100f};
    bool lpos[2];
    v8bf Bop[2][2];
    f32x2 tuH[2], tlH[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const float4 *fr4 = reinterpret_cast<const float4 *>(b.hyp_r32 + ((size_t)pair * Hp + h[c]) * kHypRec32);
        const float4 q0 = fr4[0], q1 = fr4[1], q2 = fr4[2];
        const float fr[11] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z};
        float Ff[9];
#pragma unroll
        for (int k = 0; k < 9; ++k)
            Ff[k] = on[c] ? fr[k] : 0.f;
        uint4 op[2][2];
        dense_operands_pk<1>(Ff, op);
        Bop[c][0] = __builtin_bit_cast(v8bf, half ? op[0][1] : op[0][0]);
        Bop[c][1] = __builtin_bit_cast(v8bf, half ? op[1][1] : op[1][0]);
        const float T = dense_T(Ff, X1, Y1, X2, Y2);
        const float tuh = dense_t2_upper(fr[9], T, on[c]);
        const float tlh = dense_t2_lower(fr[10], T, on[c], lpos[c]);
        tuH[c] = f32x2{tuh, tuh};
        tlH[c] = f32x2{tlh, tlh};
    }
    f32x2 cu[2] = {{0.f, 0.f}, {0.f, 0.f}}, cl[2] = {{0.f, 0.f}, {0.f, 0.f}};
    const double4 *src = reinterpret_cast<const double4 *>(b.pts + (size_t)pair * b.max_kp * 4);
    const int Mr = (M + 31) & ~31;
    for (int c0 = 0; c0 < Mr; c0 += kDenseChunk) {
        const int nc = min(kDenseChunk, Mr - c0);
        __syncthreads();
        for (int i = tid; i < nc; i += kFinishThreads) {
            uint4 o[2][2];
            if (c0 + i < M) {
                const double4 pd = src[c0 + i];
                const float x1 = (float)pd.x, y1 = (float)pd.y, x2 = (float)pd.z, y2 = (float)pd.w;
                const float ph[9] = {x2 * x1, x2 * y1, x2, y2 * x1, y2 * y1, y2, x1, y1, 1.f};
                dense_operands_pk<2>(ph, o);
            } else {
                o[0][0] = o[0][1] = o[1][0] = o[1][1] = make_uint4(0u, 0u, 0u, 0u);
            }
            uint4 *q = s_op + (size_t)(i >> 5) * 128 + (i & 31);
            q[0] = o[0][0];
            q[32] = o[0][1];
            q[64] = o[1][0];
            q[96] = o[1][1];
        }
        __syncthreads();
        if (!wave_live || wave_dead)
            continue;
        for (int p0 = 0; p0 < nc; p0 += 32) {
            if (c0 + p0 > n1 && ((c0 + p0) & 255) == 0) {
                // exit test of the wavefront, every 256 points behind the dense phase: the list is sorted, so the 64
                // entries of a wavefront are of one quality and die together.  A dead entry keeps an upper-bound count
                // below the bound (what it has + what is left < the bound), its lower bound cannot matter
                const int bn = __hip_atomic_load(b.bound + pair, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                bool any = false;
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    int U = (int)(cu[c].x + cu[c].y);
                    U += __shfl_xor(U, 32);
                    any = any || (on[c] && !(ucnt0[c] + U + (M - (c0 + p0)) < bn));
                }
                if (__ballot(any) == 0ull) {
                    wave_dead = true;
                    break;
                }
            }
            const uint4 *q = s_op + (size_t)(p0 >> 5) * 128 + half * 32 + col;
            const v8bf A0 = __builtin_bit_cast(v8bf, q[0]), A1 = __builtin_bit_cast(v8bf, q[64]);
            v16f a0 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, a1 = a0;
            a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A0, Bop[0][0], a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A0, Bop[1][0], a1, 0, 0, 0);
            a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A1, Bop[0][1], a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A1, Bop[1][1], a1, 0, 0, 0);
            ++ntile;
            if (c0 + p0 >= n1) {   // (wave-uniform) behind the dense phase: both counts
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    const f32x2 x0 = {a0[r], a0[r + 1]}, x1 = {a1[r], a1[r + 1]};
                    const f32x2 s0 = pk_mul(x0, x0), s1 = pk_mul(x1, x1);
                    cu[0] = pk_add(cu[0], pk_ind(s0, negH, tuH[0]));
                    cl[0] = pk_add(cl[0], pk_ind(s0, negH, tlH[0]));
                    cu[1] = pk_add(cu[1], pk_ind(s1, negH, tuH[1]));
                    cl[1] = pk_add(cl[1], pk_ind(s1, negH, tlH[1]));
                }
            } else {
                dense_count(a0, negH, tlH[0], cl[0]);
                dense_count(a1, negH, tlH[1], cl[1]);
            }
        }
    }
    const int npad = Mr - M;   // zero rows of the last tile: counted once by every positive threshold
    int lbest = -1;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        int U = (int)(cu[c].x + cu[c].y), L = (int)(cl[c].x + cl[c].y);
        U += __shfl_xor(U, 32);
        L += __shfl_xor(L, 32);
        U -= wave_dead ? 0 : npad;
        L -= lpos[c] ? npad : 0;
        if (half == 0 && on[c]) {
            const size_t rec = (size_t)pair * Hp + h[c];
            if (!PILOT)
                b.hyp_cnt[rec] = ucnt0[c] + U;   // + the dense phase's count over [0, n1)
            if (!wave_dead)
                lbest = max(lbest, L);
        }
        if (STATS && b.stats) {
            const unsigned long long ma = __ballot(wave_live && half == 0);   // a live wavefront computes all its columns
            if (lane == 0 && ma)
                atomicAdd(&b.stats[PILOT ? 8 : 5], (unsigned long long)__popcll(ma) * (unsigned long long)(32 * ntile));
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
        lbest = max(lbest, __shfl_xor(lbest, o));
    if (lane == 0 && lbest >= 0)
        atomicMax(&s_lmax, lbest);
    __syncthreads();
    if (tid == 0 && s_lmax >= 0)
        __hip_atomic_fetch_max(b.bound + pair, s_lmax, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// The REST of the finish (round 4).  After the first batch of every pair (ransac_finish_mfma_kernel over the 256 largest partial
// counts, upper AND lower bounds) the pair's bound is final for all practical purposes: the other listed hypotheses only need
// their UPPER count completed over the points behind the dense phase, [n1, M) -- a lower bound that does not exceed the
// bound changes nothing, and one that would exceed it belongs to a hypothesis the first batch did not hold: then the bound
// simply stays lower and more hypotheses than necessary survive (sound; the sort puts the winner into the first batch).  So
// this launch is the dense phase's loop again -- thresholds folded into the B operand, one v_alignbit per accumulator --
// over list entries instead of consecutive hypotheses, with a wavefront-level exit test every 256 points.  Round 3 ran
// ransac_finish_mfma_kernel here: every point against both thresholds with five packed instructions per two accumulators,
// in 100 k workgroups of which all but ~4 k left at once.  Grid (P, kFinUpperWg): a workgroup strides over the pair's batches.
// Rows past M in the last tile are staged as NaN monomials: their accumulators are NaN, bit 30 set, never counted.
constexpr int kFinUpperWg = 4;
constexpr int kFinUpperTest = 2;        // tiles of 32 points between two exit tests of a wavefront (a power of two)
constexpr int kFinUpperThreads = 512;   // 8 wavefronts share the staged points: 4 wavefronts per SIMD at two workgroups per CU
constexpr int kFinUpperChunk = 768;   // points staged per pass (512: three workgroups per CU, no faster)
template <bool STATS>
__global__ __launch_bounds__(kFinUpperThreads) __attribute__((amdgpu_waves_per_eu(4, 8))) void ransac_finish_upper_kernel(BatchDev b,
                                                                                                                       RunParams rp)
{
    extern __shared__ __attribute__((aligned(16))) double s_cpts[];
    const int pair = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6, col = lane & 31, half = lane >> 5;
    const int M = min(b.M[pair], b.max_kp);
    if (M < 8 || b.mode[pair] != 1)
        return;
    const int n1 = b.dense_n1[pair];
    const int Mr = (M + 31) & ~31;
    if (n1 >= Mr)
        return;   // the dense phase saw every point: the counts are complete
    const int Bnow = __hip_atomic_load(b.bound + pair, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // Only a PREFIX of the sorted list can still reach the bound: an entry with partial count u over the first n1 points ends
    // at u + (M - n1) at most, the list is sorted by u, and ransac_list_sort_kernel left the number of entries with u >= c in
    // cpos[c].  The rest of the list (most of it: every hypothesis with 32 chance inliers among the first n1 points is on it)
    // is not walked at all -- it used to cost one memory round trip per dead batch and chunk.
    const int cmin = Bnow - (M - n1);
    const int n_all = b.ccount[pair];
    const int n_list = cmin <= 0 ? n_all : min(n_all, b.cpos[(size_t)pair * kSortBins + min(cmin, min(M, kSortBins - 1))]);
    // (the first kFinishThreads entries belong to ransac_finish_mfma_kernel; batches of kFinUpperThreads entries from there)
    const int n_batches = (max(n_list - kFinishThreads, 0) + kFinUpperThreads - 1) / kFinUpperThreads;
    if ((int)blockIdx.y >= n_batches)
        return;
    const size_t Hp = (size_t)b.max_groups * kHypPerBlock;
    const uint32_t *clist = b.clist2 + (size_t)pair * Hp;   // sorted by ransac_list_sort_kernel
    uint4 *s_op = reinterpret_cast<uint4 *>(s_cpts);
    uint4 *s_win = s_op + (size_t)kFinUpperChunk * 4 + w * kDenseWin;
    const PairBox bx = load_box(b, pair);
    const float X1 = (float)fmax(dabs(bx.x1lo), dabs(bx.x1hi)) * (1.f + 0x1p-22f);
    const float Y1 = (float)fmax(dabs(bx.y1lo), dabs(bx.y1hi)) * (1.f + 0x1p-22f);
    const float X2 = (float)fmax(dabs(bx.x2lo), dabs(bx.x2hi)) * (1.f + 0x1p-22f);
    const float Y2 = (float)fmax(dabs(bx.y2lo), dabs(bx.y2hi)) * (1.f + 0x1p-22f);
    const double4 *src = reinterpret_cast<const double4 *>(b.pts + (size_t)pair * b.max_kp * 4);
    const v16f zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long evals = 0;
    // The entries of a batch sit behind dependent global round trips (list entry -> its count and record).  Only the list's live
    // prefix is walked (n_list above), so nearly every batch goes on: its count AND record are requested one batch ahead, the list
    // entry two, and the first requests of a chunk go out before the chunk's points are staged (the kernel waited in s_waitcnt for
    // 76 % of its wavefront cycles: profiles/r04_pmc_summary.json)
    struct Ent {
        int h, u;
        bool have;
        float4 q0, q1, q2;
    };
    auto fetch_h = [&](int batch, Ent &en) __attribute__((always_inline)) {
        const int e = kFinishThreads + batch * kFinUpperThreads + w * 64 + lane;   // this lane's list entry
        en.have = batch < n_batches && e < n_list;
        en.h = (int)clist[en.have ? e : 0];   // (unconditional load of a clamped index: no wait at the select)
    };
    auto fetch_rest = [&](Ent &en) __attribute__((always_inline)) {
        const size_t rec = (size_t)pair * Hp + (en.have ? en.h : 0);
        en.u = b.hyp_cnt[rec];   // upper count over the points seen so far
        const float4 *fr4 = reinterpret_cast<const float4 *>(b.hyp_r32 + rec * kHypRec32);
        en.q0 = fr4[0], en.q1 = fr4[1], en.q2 = fr4[2];
    };
    for (int c0 = n1; c0 < Mr; c0 += kFinUpperChunk) {
        const int nc = min(kFinUpperChunk, Mr - c0);
        Ent nxt, nn;   // next batch: everything; the one after: its list entry
        fetch_h(blockIdx.y, nxt);
        fetch_h(blockIdx.y + gridDim.y, nn);
        fetch_rest(nxt);   // (a later chunk re-reads the count this lane stored in the chunk before: same thread, program order)
        __syncthreads();
        for (int i = tid; i < nc; i += kFinUpperThreads) {
            uint4 o[2][2];
            if (c0 + i < M) {
                const double4 pd = src[c0 + i];
                const float x1 = (float)pd.x, y1 = (float)pd.y, x2 = (float)pd.z, y2 = (float)pd.w;
                const float ph[9] = {x2 * x1, x2 * y1, x2, y2 * x1, y2 * y1, y2, x1, y1, 1.f};
                dense_operands_pk<2>(ph, o);
            } else {
                const float qn = __builtin_nanf("");
                const float ph[9] = {qn, qn, qn, qn, qn, qn, qn, qn, qn};
                dense_operands_pk<2>(ph, o);
            }
            uint4 *q = s_op + (size_t)(i >> 5) * 128 + (i & 31);
            q[0] = o[0][0];
            q[32] = o[0][1];
            q[64] = o[1][0];
            q[96] = o[1][1];
        }
        __syncthreads();
        const int left0 = M - c0;   // points of the pair not yet seen when this chunk starts (>= 1)
        for (int batch = blockIdx.y; batch < n_batches; batch += gridDim.y) {
            const Ent cur = nxt;
            nxt.have = nn.have;
            nxt.h = nn.h;
            fetch_rest(nxt);
            fetch_h(batch + 2 * gridDim.y, nn);
            const bool have = cur.have;
            const int h_l = cur.have ? cur.h : 0;
            const size_t rec_l = (size_t)pair * Hp + h_l;
            const int u_l = cur.u;
            const bool on_l = have && !(u_l + left0 < Bnow);
            if (__ballot(on_l) == 0ull)
                continue;   // (wave-uniform) none of the 64 entries can still reach the bound
            {
                const float4 q0 = cur.q0, q1 = cur.q1, q2 = cur.q2;
                const float fr[9] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x};
                float sc = dense_scale(q2.y, dense_T(fr, X1, Y1, X2, Y2), on_l);
                float big = 0.f;
#pragma unroll
                for (int k = 0; k < 9; ++k)
                    big = fmaxf(big, fabsf(fr[k]));
                sc = (big * sc < 0x1p100f) ? sc : 0.f;
                float Fs[9];
#pragma unroll
                for (int k = 0; k < 9; ++k)
                    Fs[k] = mul_legacy(fr[k], sc);
                uint4 o[2][2];
                dense_operands_pk<1>(Fs, o);
                uint4 *wo = s_win + (lane >> 5) * 128 + (lane & 31);
                wo[0] = o[0][0];
                wo[32] = o[0][1];
                wo[64] = o[1][0];
                wo[96] = o[1][1];
            }
            int u0[2];
            bool on[2];
            v8bf Bop[2][2];
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                u0[c] = __shfl(u_l, 32 * c + col);
                on[c] = __shfl((int)on_l, 32 * c + col) != 0;
                const uint4 *ro = s_win + c * 128 + half * 32 + col;
                Bop[c][0] = __builtin_bit_cast(v8bf, ro[0]);
                Bop[c][1] = __builtin_bit_cast(v8bf, ro[64]);
            }
            uint32_t nn[2] = {0u, 0u};
            const uint4 *qt = s_op + half * 32 + col;
            const int nt = nc >> 5;
            int t = 0;
            for (; t < nt; ++t) {
                if (t > 0 && (t & (kFinUpperTest - 1)) == 0) {
                    // exit test every 64 points: the list is sorted, the 64 entries of a wavefront are of one quality and
                    // die together -- and most of the live prefix sits just above the cut (a few dozen counts of slack),
                    // so most wavefronts leave at the first or second test (every 256 points: 0.30 ms, 977 M evaluations).
                    // A dead entry keeps an upper count below the bound (what it has + what is left < bound)
                    bool any = false;
#pragma unroll
                    for (int c = 0; c < 2; ++c) {
                        int U = 16 * t - (int)nn[c];
                        U += __shfl_xor(U, 32);
                        any = any || (on[c] && !(u0[c] + U + (left0 - 32 * t) < Bnow));
                    }
                    if (__ballot(any) == 0ull)
                        break;
                }
                const uint4 *q = qt + (size_t)t * 128;
                const v8bf A0 = __builtin_bit_cast(v8bf, q[0]), A1 = __builtin_bit_cast(v8bf, q[64]);
                v16f a0, a1;
                a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A0, Bop[0][0], zero, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A0, Bop[1][0], zero, 0, 0, 0);
                a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A1, Bop[0][1], a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A1, Bop[1][1], a1, 0, 0, 0);
                nn[0] += (uint32_t)__builtin_popcount(dense_collect(a0) & 0x55555555u);
                nn[1] += (uint32_t)__builtin_popcount(dense_collect(a1) & 0x55555555u);
            }
            // back to the lane that owns the entry (it read the count, it writes it: a later chunk re-reads its own store)
            int Uc[2];
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                Uc[c] = 16 * t - (int)nn[c];
                Uc[c] += __shfl_xor(Uc[c], 32);
            }
            const int U1 = __shfl(Uc[1], lane & 31);
            if (on_l)
                b.hyp_cnt[rec_l] = u_l + (lane < 32 ? Uc[0] : U1);
            if (STATS)
                evals += (unsigned long long)__popcll(__ballot(on_l)) * (unsigned long long)(32 * t);
        }
    }
    if (STATS && b.stats && lane == 0 && evals)
        atomicAdd(&b.stats[7], evals);
}

// The work list of the exact solve (-> ransac_exact_list_kernel): records the pre-screen could not certify, and approximate
// records whose upper-bound count reaches the pair's final bound -- they may be the winner, so they get their exact F before
// ransac_select_kernel scores everything at or above the bound.  Entries are collected per workgroup in LDS and appended
// with one atomic per flush: a returning atomic per wavefront on the list's single counter stalls every wavefront for a
// memory round trip and serialises in the L2.
// Round 4: grid (kSurvWg, P), a workgroup owns a contiguous stretch of ONE pair's records and a thread four records at a
// time (one 32-bit load of the state bytes, one 128-bit load of the counts, no dependent second round trip: the flat grid of
// round 3 spent 49 iterations of state -> branch -> count per thread, 0.11 ms of latency).  Besides the work list of the exact
// solve the kernel writes the pair's CANDIDATE list for ransac_select_kernel (plist = the first half of clist, dead after the
// list sort; pcount): everything that will be scored exactly -- the work-list entries plus exact records at or above the bound
// (pairs in mode 0) --, so that the selection no longer scans the pair's 50 000 counts itself.
constexpr int kSurvList = 1024;   // per list and workgroup, flushed when the next 1024 records might not fit
constexpr int kSurvWg = 8;
__global__ __launch_bounds__(256) void ransac_survivors_kernel(BatchDev b, RunParams rp, int n_active)
{
    __shared__ uint32_t s_x[kSurvList + 1024], s_c[kSurvList + 1024];
    __shared__ int s_nx, s_nc;
    __shared__ unsigned s_bx, s_bc;
    const int pair = blockIdx.y, tid = threadIdx.x, lane = tid & 63;
    if (b.M[pair] < 8)
        return;
    const int mode = b.mode[pair];
    const int bound = b.bound[pair];
    const int H = rp.num_hypotheses;
    const size_t Hp = (size_t)b.max_groups * kHypPerBlock;
    const uint32_t *st4 = reinterpret_cast<const uint32_t *>(b.hyp_okf + (size_t)pair * Hp);
    const int4 *cnt4 = reinterpret_cast<const int4 *>(b.hyp_cnt + (size_t)pair * Hp);
    uint32_t *plist = b.clist + (size_t)pair * Hp;
    if (tid == 0) {
        s_nx = 0;
        s_nc = 0;
    }
    __syncthreads();
    auto flush = [&]() {   // called by the whole workgroup
        const int nx = s_nx, nc = s_nc;
        if (tid == 0) {
            s_bx = nx ? atomicAdd(&b.xcount[0], (unsigned)nx) : 0u;
            s_bc = nc ? (unsigned)atomicAdd(&b.pcount[pair], nc) : 0u;
        }
        __syncthreads();
        for (int i = tid; i < nx; i += 256)
            b.xlist[s_bx + i] = s_x[i];
        for (int i = tid; i < nc; i += 256)
            plist[s_bc + i] = s_c[i];
        __syncthreads();
        if (tid == 0) {
            s_nx = 0;
            s_nc = 0;
        }
        __syncthreads();
    };
    const int n4 = (int)(Hp / 4);                                   // groups of four records
    const int per_wg = (n4 + gridDim.x - 1) / gridDim.x;
    const int g_begin = blockIdx.x * per_wg, g_end = min(n4, g_begin + per_wg);
    for (int g0 = g_begin; g0 < g_end; g0 += 256) {
        const int g = g0 + tid;
        uint32_t st = 0u;
        int4 c = make_int4(-1, -1, -1, -1);
        if (g < g_end) {
            st = st4[g];
            c = cnt4[g];
        }
        const int cv[4] = {c.x, c.y, c.z, c.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int h = 4 * g + k;
            const int okf = (int)((st >> (8 * k)) & 0xffu);
            const bool live = g < g_end && h < H;
            // the exact solve still owes: records without a certificate, approximate records whose upper count reaches the bound
            const bool need = live && mode != 0 && (okf == kPsNeedExact || (okf == kPsApprox && cv[k] >= bound));
            // scored exactly by the selection: those, and exact records at or above the bound
            const bool cand = need || (live && okf == kPsExact && cv[k] >= bound);
            const unsigned long long mx = __ballot(need), mc = __ballot(cand);
            if (mc) {   // (rare: ~10 of a pair's 50 000)
                const int first = __builtin_ctzll(mc);
                int bx = 0, bc = 0;
                if (lane == first) {
                    bc = atomicAdd(&s_nc, __popcll(mc));
                    bx = mx ? atomicAdd(&s_nx, __popcll(mx)) : 0;
                }
                bc = __shfl(bc, first);
                bx = __shfl(bx, first);
                if (cand)
                    s_c[bc + __popcll(mc & ((1ull << lane) - 1ull))] = (uint32_t)h;
                if (need)
                    s_x[bx + __popcll(mx & ((1ull << lane) - 1ull))] = (uint32_t)((size_t)pair * Hp + h);
            }
        }
        __syncthreads();           // every append of this round is in
        const bool full = s_nc > kSurvList || s_nx > kSurvList;
        __syncthreads();           // ... and read by everyone: the decision is uniform
        if (full)
            flush();
    }
    flush();
}

// grid P, 256 threads.  bound[pair] is now the largest full count (every surviving hypothesis went through the
// atomicMax); hypotheses at that count are collected in index order and scored once more, one per lane over the LDS
// point stream with the fused kernel's exact operations (count AND residual sum in index order), and the best by
// (residual, index) becomes the pair's single WgBest record; finalize_model reduces the records as before.
constexpr int kSelThreads = 256;
constexpr int kSelList = 2048;   // capacity of the candidate list in LDS
constexpr int kSelWaveCount = 96;   // up to this many candidates are counted one per wavefront (lanes = points), more one per lane
constexpr int kSelSeg = 1024;    // hypotheses scanned between two barriers (the list is worked off once a segment might overflow it)
constexpr int kSelSerial = 24;   // up to this many ties are scored one at a time by the whole workgroup

__global__ __launch_bounds__(kSelThreads) void ransac_select_kernel(BatchDev b, RunParams rp, int use_plist)
{
    extern __shared__ __attribute__((aligned(16))) double s_spts[];   // [max_kp][4] points OR [max_kp] residuals
    __shared__ uint32_t s_list[kSelList];
    __shared__ int s_cnt[kSelList];
    __shared__ int s_nlist;
    __shared__ double s_F[9];
    __shared__ int s_tot[4];
    __shared__ Cand s_c[4];
    __shared__ Cand s_best;
    __shared__ double s_bestF[9];
    __shared__ uint32_t s_win;
    const int pair = blockIdx.x, tid = threadIdx.x;
    const int M = min(b.M[pair], b.max_kp);
    const int H = rp.num_hypotheses;
    const int G = (H + kHypPerBlock - 1) / kHypPerBlock;
    WgBest *out = b.wgbest + (size_t)pair * b.max_groups;
    for (int g = 1 + tid; g < G; g += kSelThreads) {
        out[g].count = -1;
        out[g].hyp = 0xffffffffu;
        out[g].residual = 0.0;
    }
    if (tid == 0) {
        s_best.cnt = -2;
        s_best.hyp = 0xffffffffu;
        s_best.res = 0.0;
    }
    if (M < 8) {  // estimator-RANSAC.cpp:25-29
        if (tid == 0) {
            out[0].count = -1;
            out[0].hyp = 0xffffffffu;
            out[0].residual = 0.0;
        }
        return;
    }
    const size_t Hp = (size_t)b.max_groups * kHypPerBlock;
    const double4 *P4 = reinterpret_cast<const double4 *>(b.pts + (size_t)pair * b.max_kp * 4);
    bool staged = false;   // the dynamic LDS block holds the pair's points (lane-per-hypothesis path) or residuals
    __syncthreads();
    const int cmax = b.bound[pair];
    const int32_t *cntp = b.hyp_cnt + (size_t)pair * Hp;
    const uint8_t *okb = b.hyp_okf + (size_t)pair * Hp;   // a listed hypothesis whose exact solve rejected the sample is out
    const double *Fp = b.hyp_F + (size_t)pair * Hp * kHypRec;
    const double thr = pair_max_error_sq(b, rp, pair);
    const double4 *L4 = reinterpret_cast<const double4 *>(s_spts);
    double *s_r = s_spts;
    const int lane = tid & 63, w = tid >> 6;
    int scan = 0;   // position in the pair's candidate list (use_plist) or in its hypothesis range
    const int n_total = use_plist ? b.pcount[pair] : H;
    const uint32_t *plist = b.clist + (size_t)pair * Hp;
    while (scan < n_total) {
        if (tid == 0)
            s_nlist = 0;
        __syncthreads();
        if (use_plist) {
            // the candidates ransac_survivors_kernel listed: everything whose (upper-bound) count reaches the pair's bound
            const int n = min(kSelList, n_total - scan);
            for (int q = tid; q < n; q += kSelThreads)
                s_list[q] = plist[scan + q];
            if (tid == 0)
                s_nlist = n;
            scan += n;
            __syncthreads();
        } else {
            // (no candidate list: the diagnostics ladder's launches) collect the hypotheses whose count reaches the pair's
            // bound, appended per wavefront with one LDS atomic, no barrier inside a segment of kSelSeg hypotheses; the list
            // order is immaterial, candidates are compared by (count, residual, index)
            while (scan < H) {
                const int seg_end = min(H, scan + kSelSeg);
                for (int h = scan + tid; h < seg_end; h += kSelThreads) {
                    const bool flag = cntp[h] >= cmax && okb[h] != kPsInvalid;
                    const unsigned long long bal = __ballot(flag);
                    if (bal) {
                        int base = 0;
                        if (lane == __builtin_ctzll(bal))
                            base = atomicAdd(&s_nlist, __popcll(bal));
                        base = __shfl(base, __builtin_ctzll(bal));
                        if (flag)
                            s_list[base + __popcll(bal & ((1ull << lane) - 1ull))] = (uint32_t)h;
                    }
                }
                scan = seg_end;
                __syncthreads();
                if (s_nlist > kSelList - kSelSeg)
                    break;   // (uniform) the next segment might not fit: work this list off first
            }
        }
        int n_list = s_nlist;
        __syncthreads();
        if (n_list == 0)
            continue;
        // Exact COUNTS first (round 4): with pre-screened counts the list holds every hypothesis whose UPPER bound reaches the
        // pair's bound (~10 per pair on the bench workload, dozens to hundreds elsewhere), but only those that tie at the
        // largest exact count need the residual sum in index order -- a serial chain of one addition per inlier, the
        // latency floor of this kernel (0.25 ms per 512 pairs when every listed hypothesis went through it).  One wavefront
        // per hypothesis, lanes are points: the count is a handful of ballots.  Pairs in mode 0 carry exact counts already.
        if (b.mode[pair] != 0 && n_list > kSelWaveCount) {
            // long lists (pairs with many uncertified hypotheses: ~1000 per pair on the small-baseline sequence): one
            // hypothesis per LANE over the LDS point stream, 256 at a time, counts only
            if (!staged) {
                const double2 *src = reinterpret_cast<const double2 *>(P4);
                double2 *dst = reinterpret_cast<double2 *>(s_spts);
                for (int i = tid; i < 2 * M; i += kSelThreads)
                    dst[i] = src[i];
                staged = true;
                __syncthreads();
            }
            for (int q = tid; q < n_list; q += kSelThreads) {
                const uint32_t h = s_list[q];
                int c = -1;
                if (okb[h] != kPsInvalid) {
                    const double *f = Fp + (size_t)h * kHypRec;
                    double F[9];
#pragma unroll
                    for (int k = 0; k < 9; ++k)
                        F[k] = f[k];
                    c = 0;
#pragma unroll 4
                    for (int i = 0; i < M; ++i) {
                        const double4 p = L4[i];
                        c += epipolar_residual(F, p.x, p.y, p.z, p.w) < thr ? 1 : 0;
                    }
                }
                s_cnt[q] = c;
            }
        } else if (b.mode[pair] != 0) {
            for (int q = w; q < n_list; q += kSelThreads / 64) {
                if (okb[s_list[q]] == kPsInvalid) {   // (wave-uniform) its exact solve rejected the sample: out
                    if (lane == 0)
                        s_cnt[q] = -1;
                    continue;
                }
                const double *f = Fp + (size_t)s_list[q] * kHypRec;
                double F[9];
#pragma unroll
                for (int k = 0; k < 9; ++k)
                    F[k] = f[k];
                int c = 0;
                for (int i = lane; i < M; i += 64) {
                    const double4 p = P4[i];
                    c += epipolar_residual(F, p.x, p.y, p.z, p.w) < thr ? 1 : 0;
                }
#pragma unroll
                for (int o = 32; o > 0; o >>= 1)
                    c += __shfl_xor(c, o);
                if (lane == 0)
                    s_cnt[q] = c;
            }
        } else {
            for (int q = tid; q < n_list; q += kSelThreads)
                s_cnt[q] = cntp[s_list[q]];
        }
        __syncthreads();
        {
            // the largest count of this round against the best so far; the hypotheses at that count, in index order
            int cm = -1;
            for (int q = tid; q < n_list; q += kSelThreads)
                cm = max(cm, s_cnt[q]);
#pragma unroll
            for (int o = 32; o > 0; o >>= 1)
                cm = max(cm, __shfl_xor(cm, o));
            if (lane == 0)
                s_tot[w] = cm;
            __syncthreads();
            const int target = max(max(max(s_tot[0], s_tot[1]), max(s_tot[2], s_tot[3])), s_best.cnt);
            __syncthreads();
            int n_ties = 0;
            for (int q0 = 0; q0 < n_list; q0 += kSelThreads) {
                const int q = q0 + tid;
                const bool tie = q < n_list && s_cnt[q] == target;
                const uint32_t hq = q < n_list ? s_list[q] : 0u;
                const unsigned long long bal = __ballot(tie);
                if (lane == 0)
                    s_tot[w] = __popcll(bal);
                __syncthreads();   // (also: every s_list[q] of this stretch has been read)
                int off = n_ties, tot = 0;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int v = s_tot[k];
                    off += (k < w) ? v : 0;
                    tot += v;
                }
                if (tie)
                    s_list[off + __popcll(bal & ((1ull << lane) - 1ull))] = hq;   // off + ... <= q: in-place compaction
                n_ties += tot;
                __syncthreads();
            }
            n_list = n_ties;
        }
        if (n_list == 0)
            continue;
        if (n_list <= kSelSerial) {
            // few ties (the usual case is one): the whole workgroup scores ONE hypothesis at a time -- residuals of all
            // points in parallel, inliers compacted in index order, then one lane adds them up in that order.  The
            // sequential replacement rule's sum is res = fma(r_i, m_i, res) with m_i in {0, 1} over all i, i.e. the
            // inlier residuals added in index order, one rounding each: the same bits.
            staged = false;
            for (int q = 0; q < n_list; ++q) {
                const uint32_t h = s_list[q];
                if (tid < 9)
                    s_F[tid] = Fp[(size_t)h * kHypRec + tid];
                __syncthreads();
                double F[9];
#pragma unroll
                for (int k = 0; k < 9; ++k)
                    F[k] = s_F[k];
                int n_in = 0;
                for (int start = 0; start < M; start += kSelThreads) {
                    const int i = start + tid;
                    double r = 0.0;
                    bool in = false;
                    if (i < M) {
                        const double4 p = P4[i];
                        r = epipolar_residual(F, p.x, p.y, p.z, p.w);
                        in = r < thr;
                    }
                    const unsigned long long bal = __ballot(in);
                    if (lane == 0)
                        s_tot[w] = __popcll(bal);
                    __syncthreads();
                    int off = n_in, tot = 0;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int v = s_tot[k];
                        off += (k < w) ? v : 0;
                        tot += v;
                    }
                    if (in)
                        s_r[off + __popcll(bal & ((1ull << lane) - 1ull))] = r;
                    n_in += tot;
                    __syncthreads();
                }
                if (tid == 0) {
                    double res = 0.0;
                    int i = 0;
                    for (; i + 8 <= n_in; i += 8) {
                        const double r0 = s_r[i], r1 = s_r[i + 1], r2 = s_r[i + 2], r3 = s_r[i + 3];
                        const double r4 = s_r[i + 4], r5 = s_r[i + 5], r6 = s_r[i + 6], r7 = s_r[i + 7];
                        res += r0; res += r1; res += r2; res += r3;
                        res += r4; res += r5; res += r6; res += r7;
                    }
                    for (; i < n_in; ++i)
                        res += s_r[i];
                    const Cand me{n_in, h, res};
                    if (s_best.cnt < 0 || cand_better(me, s_best)) {
                        s_best = me;
#pragma unroll
                        for (int k = 0; k < 9; ++k)
                            s_bestF[k] = F[k];
                    }
                }
                __syncthreads();
            }
            continue;
        }
        if (!staged) {
            const double2 *src = reinterpret_cast<const double2 *>(P4);
            double2 *dst = reinterpret_cast<double2 *>(s_spts);
            for (int i = tid; i < 2 * M; i += kSelThreads)
                dst[i] = src[i];
            staged = true;
            __syncthreads();
        }
        for (int q0 = 0; q0 < n_list; q0 += kSelThreads) {
            const bool have = q0 + tid < n_list;
            const uint32_t h = have ? s_list[q0 + tid] : 0xffffffffu;
            Cand me{-2, h, 0.0};
            double F[9];
            if (have) {
                const double *f = Fp + (size_t)h * kHypRec;
#pragma unroll
                for (int k = 0; k < 9; ++k)
                    F[k] = f[k];
                int cnt = 0;
                double res = 0.0;
#pragma unroll 4
                for (int i = 0; i < M; ++i) {   // the fused kernel's loop: same operations, same order, same bits
                    const double4 p = L4[i];
                    const double r = epipolar_residual(F, p.x, p.y, p.z, p.w);
                    const bool in = r < thr;
                    cnt += in ? 1 : 0;
                    res += in ? r : 0.0;   // NaN-safe (a NaN residual is no inlier and adds nothing, as in the reference)
                }
                me.cnt = cnt;
                me.res = res;
            }
            Cand red = me;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                Cand other;
                other.cnt = __shfl_xor(red.cnt, o);
                other.hyp = __shfl_xor(red.hyp, o);
                other.res = __shfl_xor(red.res, o);
                if (other.cnt >= 0 && (red.cnt < 0 || cand_better(other, red)))
                    red = other;
            }
            if (lane == 0)
                s_c[w] = red;
            __syncthreads();
            if (tid == 0) {
                Cand best = s_best;
                bool repl = false;
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (s_c[k].cnt >= 0 && (best.cnt < 0 || cand_better(s_c[k], best))) {
                        best = s_c[k];
                        repl = true;
                    }
                s_best = best;
                s_win = repl ? best.hyp : 0xfffffffeu;
            }
            __syncthreads();
            if (have && h == s_win) {
#pragma unroll
                for (int k = 0; k < 9; ++k)
                    s_bestF[k] = F[k];
            }
            __syncthreads();
        }
    }
    if (tid == 0) {
        const Cand best = s_best;
        out[0].count = best.cnt >= 0 ? best.cnt : -1;
        out[0].hyp = best.cnt >= 0 ? best.hyp : 0xffffffffu;
        out[0].residual = best.cnt >= 0 ? best.res : 0.0;
        if (best.cnt >= 0) {
#pragma unroll
            for (int k = 0; k < 9; ++k)
                out[0].F[k] = s_bestF[k];
        }
    }
}


// find_fundamental_matrix on one explicit sample (single lane); diagnostics / API parity only.
__global__ __launch_bounds__(64, 1) void fundamental_kernel(const double *p1, const double *p2, double *Fout, int *okout)
{
    if (threadIdx.x != 0)
        return;
    double x1[8], y1[8], x2[8], y2[8], F[9];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        x1[k] = p1[2 * k]; y1[k] = p1[2 * k + 1];
        x2[k] = p2[2 * k]; y2[k] = p2[2 * k + 1];
    }
    unsigned rot = 0, pairs = 0;
    bool bad = false;
    // the pair step of the RANSAC solve (unscaled sequences, sqrt-free test, 9x9 and 3x3), so that the bitwise F
    // tests of this entry point cover exactly the arithmetic the hot kernel runs
    bool ok = eight_point<16 + 32 + 128 + 1024>(x1, y1, x2, y2, F, rot, pairs, bad);
    if (bad)
        ok = eight_point<16>(x1, y1, x2, y2, F, rot, pairs, bad);
#pragma unroll
    for (int k = 0; k < 9; ++k)
        Fout[k] = F[k];
    *okout = ok ? 1 : 0;
}

// ---------------------------------------------------------------------------------------------
// finalize, three launches so that the 4 x M_inl triangulations of ONE pair spread over the chip (a single pair
// at a time is BASELINE configs[1]; with one workgroup per pair they took 0.19 ms of a 0.5 ms step):
//   finalize_model   grid P          arg-best, inlier mask + ordered inlier list, E projection, decomposition
//   triangulate      grid (x, P)     one (candidate, inlier) item per lane: 4x4 DLT + Jacobi SVD + cheirality
//   finalize_select  grid P          candidate selection, ordered compaction, pose
// ---------------------------------------------------------------------------------------------
constexpr int kFinThreads = 256;

// ordered compaction of {i in [0, n) : pred(i)}; emit(i, position); returns the count (block-uniform)
template <typename Pred, typename Emit>
__device__ __forceinline__ int block_compact(int n, int *s_tot, Pred pred, Emit emit)
{
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    int basepos = 0;
    for (int start = 0; start < n; start += kFinThreads) {
        const int i = start + tid;
        const bool flag = (i < n) && pred(i);
        const unsigned long long bal = __ballot(flag);
        const int pre = __popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0)
            s_tot[w] = __popcll(bal);
        __syncthreads();
        int off = basepos, tot = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int v = s_tot[k];
            off += (k < w) ? v : 0;
            tot += v;
        }
        if (flag)
            emit(i, off + pre);
        basepos += tot;
        __syncthreads();
    }
    return basepos;
}

// triangulation of ONE (candidate c, inlier j): 4x4 DLT + SVD on the lane (sfm-solve.cpp:134-227), cheirality flag and point
// into the candidate's scratch rows.  R / Rr / T: the candidate's raw and rectified rotation (row-major 9) and +-t.
__device__ __forceinline__ void triangulate_item(const BatchDev &b, int pair, int c, int j, const double *R, const double *Rr,
                                                 double t0, double t1, double t2)
{
    const size_t base = (size_t)pair * b.max_kp;
    const int i = b.inl[base + j];
    const double4 p = *reinterpret_cast<const double4 *>(b.pts + (base + i) * 4);
    double At[4][4];  // At[col][row] of A
    auto design = [&]() {
        At[0][0] = -1.0; At[1][0] = 0.0;  At[2][0] = p.x; At[3][0] = 0.0;
        At[0][1] = 0.0;  At[1][1] = -1.0; At[2][1] = p.y; At[3][1] = 0.0;
        At[0][2] = p.z * Rr[6] - Rr[0]; At[1][2] = p.z * Rr[7] - Rr[1]; At[2][2] = p.z * Rr[8] - Rr[2]; At[3][2] = p.z * t2 - t0;
        At[0][3] = p.w * Rr[6] - Rr[3]; At[1][3] = p.w * Rr[7] - Rr[4]; At[2][3] = p.w * Rr[8] - Rr[5]; At[3][3] = p.w * t2 - t1;
    };
    double X[4];
    unsigned rot = 0, prs = 0;
    bool bad = false;
    design();
    svd4_last_vt_row<true>(At, X, rot, prs, bad);
    if (__builtin_expect(__any(bad), 0)) {   // a range guard of the unscaled sequences was violated: full sequences
        design();
        svd4_last_vt_row<false>(At, X, rot, prs, bad);
    }
    bool okp = !(dabs(X[3]) < kTol);
    const double scale = 1.0 / X[3];
    const double px = X[0] * scale, py = X[1] * scale, pz = X[2] * scale;
    okp = okp && !(pz < kTol);
    const double z2 = ((R[6] * px + R[7] * py) + R[8] * pz) + t2;
    okp = okp && !(z2 < kTol);
    b.okf[((size_t)pair * 4 + c) * b.max_kp + j] = okp ? 1 : 0;
    double *dst = b.cand_pts + (((size_t)pair * 4 + c) * b.max_kp + j) * 3;
    dst[0] = px; dst[1] = py; dst[2] = pz;
}

__global__ __launch_bounds__(kFinThreads) void finalize_model_kernel(BatchDev b, RunParams rp, int mode)
{
    __shared__ double s_F[9], s_E[9];
    __shared__ int s_tot[4];
    __shared__ int s_proceed;
    __shared__ Cand s_red[4];
    __shared__ uint32_t s_grp[4];
    const int pair = blockIdx.x, tid = threadIdx.x;
    const size_t base = (size_t)pair * b.max_kp;
    const int M = min(b.M[pair], b.max_kp);
    mvs_pair_result *res = b.results + pair;
    FinModel *fm = b.fin + pair;
    const double *P = b.pts + base * 4;
    uint8_t *mask = b.mask + base;

    if (tid == 0)
        s_proceed = 0;
    __syncthreads();

    if (mode == kFinalizeFull) {
        // ---- arg-best over the workgroup records (sequential-replacement order) ----
        const int G = (rp.num_hypotheses + kHypPerBlock - 1) / kHypPerBlock;
        Cand me{-2, 0xffffffffu, 0.0};
        uint32_t mygroup = 0;
        if (M >= 8) {
            const WgBest *wb = b.wgbest + (size_t)pair * b.max_groups;
            for (int g = tid; g < G; g += kFinThreads) {
                Cand c{wb[g].count, wb[g].hyp, wb[g].residual};
                if (c.cnt >= 0 && (me.cnt < -1 || cand_better(c, me))) {
                    me = c;
                    mygroup = g;
                }
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            Cand other;
            other.cnt = __shfl_xor(me.cnt, o);
            other.hyp = __shfl_xor(me.hyp, o);
            other.res = __shfl_xor(me.res, o);
            const uint32_t og = __shfl_xor(mygroup, o);
            if (other.cnt >= 0 && (me.cnt < -1 || cand_better(other, me))) {
                me = other;
                mygroup = og;
            }
        }
        if ((tid & 63) == 0) {
            s_red[tid >> 6] = me;
            s_grp[tid >> 6] = mygroup;
        }
        __syncthreads();
        if (tid == 0) {
            Cand best = s_red[0];
            uint32_t bg = s_grp[0];
            for (int w = 1; w < 4; ++w)
                if (s_red[w].cnt >= 0 && (best.cnt < -1 || cand_better(s_red[w], best))) {
                    best = s_red[w];
                    bg = s_grp[w];
                }
            res->valid = 0;
            res->n_matches = M;
            res->n_inliers = 0;
            res->n_points = 0;
            res->best_hyp = best.cnt >= 0 ? (int)best.hyp : -1;
            res->best_count = best.cnt >= 0 ? best.cnt : 0;
            res->best_residual = best.cnt >= 0 ? best.res : 0.0;
            if (best.cnt >= 0) {
                const WgBest *wb = b.wgbest + (size_t)pair * b.max_groups + bg;
#pragma unroll
                for (int k = 0; k < 9; ++k) {
                    s_F[k] = wb->F[k];
                    res->F[k] = wb->F[k];
                }
                s_proceed = 1;
            }
        }
        __syncthreads();
        for (int i = M + tid; i < b.max_kp; i += kFinThreads)   // rows past the match list: cleared (deterministic downloads)
            mask[i] = 0;
        if (!s_proceed) {
            for (int i = tid; i < M; i += kFinThreads)
                mask[i] = 0;
            if (tid == 0) {
                fm->proceed = 0;
                fm->n_inl = 0;
                fm->ncand = 0;
            }
            return;
        }
        // ---- inlier mask of the winner (estimator-RANSAC.cpp:100-129) ----
        double F[9];
#pragma unroll
        for (int k = 0; k < 9; ++k)
            F[k] = s_F[k];
        const double thr = pair_max_error_sq(b, rp, pair);
        for (int i = tid; i < M; i += kFinThreads) {
            const double4 p = *reinterpret_cast<const double4 *>(P + (size_t)i * 4);
            mask[i] = epipolar_residual(F, p.x, p.y, p.z, p.w) < thr ? 1 : 0;
        }
        __syncthreads();
    } else if (tid == 0) {
        res->valid = 0;
        res->n_matches = M;
        res->n_points = 0;
        res->best_hyp = -1;
        res->best_count = 0;
        res->best_residual = 0.0;
    }

    // ---- ordered inlier list ----
    uint16_t *inl = b.inl + base;
    const int n_inl = block_compact(
        M, s_tot, [&](int i) { return mask[i] != 0; }, [&](int i, int pos) { inl[pos] = (uint16_t)i; });

    // ---- E projection + decomposition (single lane; sfm-solve.cpp:74-84,97-127) ----
    if (tid == 0) {
        res->n_inliers = n_inl;
        unsigned rot = 0, prs = 0;
        bool go = true;
        int ncand = 4;
        if (mode == kFinalizeFull) {
            double Fm[3][3], w[3], U[3][3], Vt[3][3];
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j)
                    Fm[i][j] = s_F[i * 3 + j];
            svd3_full(Fm, w, U, Vt, rot, prs);
            const double v = dsqrt(w[0] * w[1]);
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const double a = U[i][0] * v, c = U[i][1] * v;
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const double e = a * Vt[0][j] + c * Vt[1][j];
                    s_E[i * 3 + j] = e;
                    res->E[i * 3 + j] = e;
                }
            }
            // compute() returns best_count > 0 (estimator-RANSAC.cpp:89); sfm_solve needs >= min inliers (:330)
            go = (res->best_count > 0) && (n_inl >= rp.min_inliers);
        } else if (mode == kFinalizeFromE) {
#pragma unroll
            for (int k = 0; k < 9; ++k)
                s_E[k] = res->E[k];
        }
        if (mode == kFinalizeTriangulate) {
            double R[3][3];
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    R[i][j] = res->R1to2[i * 3 + j];
                    fm->R[0][i * 3 + j] = R[i][j];
                }
            rectify3(R);
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j)
                    fm->Rr[0][i * 3 + j] = R[i][j];
#pragma unroll
            for (int k = 0; k < 3; ++k)
                fm->T[k] = res->t1to2[k];
            ncand = 1;
        } else if (go) {
            double Em[3][3], w[3], U[3][3], Vt[3][3], V[3][3];
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j)
                    Em[i][j] = s_E[i * 3 + j];
            svd3_full(Em, w, U, Vt, rot, prs);
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j)
                    V[i][j] = Vt[j][i];
            if (det3(U) < 0.0) {
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int j = 0; j < 3; ++j)
                        U[i][j] = -U[i][j];
            }
            if (det3(V) < 0.0) {
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int j = 0; j < 3; ++j)
                        V[i][j] = -V[i][j];
            }
            double Ra[3][3], Rb[3][3];
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    Ra[i][j] = (U[i][1] * V[j][0] + (-U[i][0]) * V[j][1]) + U[i][2] * V[j][2];
                    Rb[i][j] = ((-U[i][1]) * V[j][0] + U[i][0] * V[j][1]) + U[i][2] * V[j][2];
                    fm->R[0][i * 3 + j] = Ra[i][j];
                    fm->R[1][i * 3 + j] = Rb[i][j];
                }
            // S = U Z U^T, t = (-S12, S02, -S01)
            fm->T[0] = -((-U[1][1]) * U[2][0] + U[1][0] * U[2][1]);
            fm->T[1] = ((-U[0][1]) * U[2][0] + U[0][0] * U[2][1]);
            fm->T[2] = -((-U[0][1]) * U[1][0] + U[0][0] * U[1][1]);
            rectify3(Ra);
            rectify3(Rb);
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    fm->Rr[0][i * 3 + j] = Ra[i][j];
                    fm->Rr[1][i * 3 + j] = Rb[i][j];
                }
            ncand = 4;
        }
        fm->proceed = (go && n_inl > 0) ? 1 : 0;
        fm->n_inl = n_inl;
        fm->ncand = ncand;
        fm->npre = min(n_inl, kFinThreads / ncand);
        s_proceed = fm->proceed;
    }
    __syncthreads();   // (workgroup-scope fence: the lane-0 stores to *fm above are visible to the whole workgroup)
    if (!s_proceed)
        return;
    // ---- the prefix: every candidate on the first 256 / ncand inliers (wavefront = candidate when there are four) ----
    {
        const int ncand = fm->ncand, npre = fm->npre;
        const int per = kFinThreads / ncand;
        const int c = tid / per, j = tid - c * per;
        if (tid < 4)
            s_tot[tid] = 0;
        __syncthreads();
        if (j < npre) {
            const bool flip = (c & 1) != 0;
            triangulate_item(b, pair, c, j, fm->R[c >> 1], fm->Rr[c >> 1], flip ? -fm->T[0] : fm->T[0],
                             flip ? -fm->T[1] : fm->T[1], flip ? -fm->T[2] : fm->T[2]);
            if (b.okf[((size_t)pair * 4 + c) * b.max_kp + j])
                atomicAdd(&s_tot[c], 1);
        }
        __syncthreads();
        if (tid == 0) {
            int best = -1, bc = 0;
            for (int k = 0; k < ncand; ++k) {
                fm->pre_cnt[k] = s_tot[k];
                if (s_tot[k] > best) {
                    best = s_tot[k];
                    bc = k;
                }
            }
            for (int k = ncand; k < 4; ++k)
                fm->pre_cnt[k] = 0;
            fm->best_c = bc;
        }
    }
}

// Lazy candidate evaluation (round 5).  recover_pose_and_points (sfm-solve.cpp:250-280) triangulates every inlier under all
// four (R, t) candidates and keeps the one with strictly more points in front of both cameras, first candidate on ties.  Three
// of the four lose by a wide margin on any real pair, and to know that a candidate loses it is enough that ALL its remaining
// points could not lift it past the winner.  Every evaluated point is the contract's own arithmetic; only the number of points
// evaluated for the losers changes:
//   finalize_model   every candidate on the PREFIX (the first 256 / ncand inliers): pre_cnt[c]; best_c = its arg-max;
//   triangulate      best_c on the rest of the inliers (grid (ceil(max_kp / 256), P));
//   finalize_select  count* of best_c; a candidate c is COMPLETED there (rare) iff pre_cnt[c] + (n_inl - npre) > count*, or
//                    == count* with c < best_c -- otherwise its full count is provably not the winner's and its prefix count
//                    stands in for it in the reference's selection loop (strict >, candidate order), which picks the same winner.
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void triangulate_kernel(BatchDev b)
{
    const int pair = blockIdx.y;
    const FinModel *fm = b.fin + pair;
    if (!fm->proceed)
        return;
    const int j = fm->npre + blockIdx.x * 256 + threadIdx.x;
    if (j >= fm->n_inl)
        return;
    const int c = fm->best_c;
    const bool flip = (c & 1) != 0;
    triangulate_item(b, pair, c, j, fm->R[c >> 1], fm->Rr[c >> 1], flip ? -fm->T[0] : fm->T[0], flip ? -fm->T[1] : fm->T[1],
                     flip ? -fm->T[2] : fm->T[2]);
}

__global__ __launch_bounds__(kFinThreads) void finalize_select_kernel(BatchDev b)
{
    __shared__ int s_tot[4];
    __shared__ int s_cnt[4];
    __shared__ int s_win;
    const int pair = blockIdx.x, tid = threadIdx.x;
    const FinModel *fm = b.fin + pair;
    const size_t base = (size_t)pair * b.max_kp;
    // rows [n_points, max_kp) of points / point_idx are cleared on every path (deterministic whole-capacity downloads)
    auto clear_tail = [&](int from) {
        for (int i = from + tid; i < b.max_kp; i += kFinThreads) {
            double *d = b.points + (base + i) * 3;
            d[0] = 0.0; d[1] = 0.0; d[2] = 0.0;
            b.point_idx[base + i] = 0;
        }
    };
    if (!fm->proceed) {
        clear_tail(0);
        return;
    }
    mvs_pair_result *res = b.results + pair;
    const int n_inl = fm->n_inl, ncand = fm->ncand;
    const uint8_t *okf = b.okf + (size_t)pair * 4 * b.max_kp;
    // ---- candidate selection: strictly more points wins, order (Ra,t),(Ra,-t),(Rb,t),(Rb,-t) (sfm-solve.cpp:259-281).
    // best_c has been triangulated on every inlier, the others on the prefix only (see triangulate_kernel): count* first,
    // then any candidate whose remaining points could still lift it to the winner's place is completed here (rare).
    const int npre = fm->npre, bc = fm->best_c, rem = n_inl - npre;
    auto count_rest = [&](int c) {   // points of candidate c behind the prefix, added to s_cnt[c]
        int cnt = 0;
        for (int j = npre + tid; j < n_inl; j += kFinThreads)
            cnt += okf[(size_t)c * b.max_kp + j];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1)
            cnt += __shfl_xor(cnt, o);
        if ((tid & 63) == 0 && cnt)
            atomicAdd(&s_cnt[c], cnt);
    };
    if (tid < 4)
        s_cnt[tid] = tid < ncand ? fm->pre_cnt[tid] : 0;
    __syncthreads();
    count_rest(bc);
    __syncthreads();
    {
        const int cstar = s_cnt[bc];
        for (int c = 0; c < ncand; ++c) {
            if (c == bc)
                continue;
            const int ub = fm->pre_cnt[c] + rem;
            if (!(ub > cstar || (ub == cstar && c < bc)))
                continue;            // cannot take best_c's place whatever its remaining points do (workgroup-uniform)
            const bool flip = (c & 1) != 0;
            for (int j0 = npre; j0 < n_inl; j0 += kFinThreads)
                if (j0 + tid < n_inl)
                    triangulate_item(b, pair, c, j0 + tid, fm->R[c >> 1], fm->Rr[c >> 1], flip ? -fm->T[0] : fm->T[0],
                                     flip ? -fm->T[1] : fm->T[1], flip ? -fm->T[2] : fm->T[2]);
            __syncthreads();         // (workgroup-scope fence: the flags just written are read by other lanes below)
            count_rest(c);
        }
    }
    __syncthreads();
    if (tid == 0) {
        int best = 0, win = -1;
        for (int c = 0; c < ncand; ++c)
            if (s_cnt[c] > best) {
                best = s_cnt[c];
                win = c;
            }
        s_win = win;
    }
    __syncthreads();
    const int win = s_win;
    if (win < 0) {
        clear_tail(0);
        return;  // recover_pose_and_points returned false
    }
    // ---- compact the winner's points in index order ----
    const double *src = b.cand_pts + ((size_t)pair * 4 + win) * b.max_kp * 3;
    const uint8_t *okw = okf + (size_t)win * b.max_kp;
    const uint16_t *inl = b.inl + base;
    const int n_pts = block_compact(
        n_inl, s_tot, [&](int j) { return okw[j] != 0; },
        [&](int j, int pos) {
            double *d = b.points + (base + pos) * 3;
            d[0] = src[j * 3]; d[1] = src[j * 3 + 1]; d[2] = src[j * 3 + 2];
            b.point_idx[base + pos] = inl[j];
        });
    clear_tail(n_pts);
    // ---- pose2in1 = SE3(SO3(R), t).inverse()  (sfm-solve.cpp:364; lie-group.hpp:212-216) ----
    if (tid == 0) {
        const double *Rw = fm->R[win >> 1];
        const bool flip = (win & 1) != 0;
        const double t[3] = {flip ? -fm->T[0] : fm->T[0], flip ? -fm->T[1] : fm->T[1], flip ? -fm->T[2] : fm->T[2]};
        double Rr[3][3], RT[3][3];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                Rr[i][j] = Rw[i * 3 + j];
                res->R1to2[i * 3 + j] = Rw[i * 3 + j];
            }
        rectify3(Rr);
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j)
                RT[i][j] = Rr[j][i];
        rectify3(RT);
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            res->t1to2[i] = t[i];
            res->t[i] = -((RT[i][0] * t[0] + RT[i][1] * t[1]) + RT[i][2] * t[2]);
#pragma unroll
            for (int j = 0; j < 3; ++j)
                res->R[i * 3 + j] = RT[i][j];
        }
        res->n_points = n_pts;
        res->valid = 1;
    }
}

// The retired kernels of the experiment ladder (rounds 1-3: ransac_solve / ransac_score / ransac_solve_av / ransac_count and the
// device-side probes of the matrix-core tile and of the unscaled sequences) live in their own file, which ONLY the diagnostics
// build compiles: libmvslam_hip.so contains none of it (tests/test_abi.py reads its symbol table).
#ifdef MVS_DEBUG_HOOKS
#include "kernels_experiments.hip"
#endif

// ---------------------------------------------------------------------------------------------
// launch wrappers
// ---------------------------------------------------------------------------------------------
constexpr int kCntThreads = 768;   // 12 wavefronts: two workgroups fit a CU (2 x 64 KB of LDS, 6 of the 7 wavefronts per SIMD the
constexpr int kCntPpl = 2;         // 70 registers allow) against one workgroup of 1024 (4 per SIMD): 42.7 -> 42.3 ms per 512 pairs
constexpr int kSolveBlock = 64;    // one wavefront per workgroup: every SIMD refills on its own (45.4 -> 44.5 ms, same bits)
#ifdef MVS_DEBUG_HOOKS
static int kSplitMinPairs = 3;     // (diagnostics: settable, tools/single_pair.py times a single pair on the pre-screened stage)
void set_split_min_pairs(int v) { kSplitMinPairs = v; }
#else
constexpr int kSplitMinPairs = 3;  // one or two pairs stay on the fused kernel (latency: fewer launches)
#endif

// Which launches take the pre-screened stage: three or more pairs; two pairs when their hypotheses fill the chip more than once
// on the fused kernel (2 x 50 000: 0.35 ms against 0.40, profiles/r04_single_pair_paths.json); one pair never (0.32 against 0.25:
// the stage is a chain of latencies there).  (Diagnostics: a split minimum other than 3 is taken literally.)
// ("the chip": one fused workgroup per compute unit at one wavefront per SIMD -- the device's own count, BatchDev::cu_count.)
static bool prescreened_launch(const BatchDev &b, int n_active, int H)
{
    const int cus = b.cu_count > 0 ? b.cu_count : 256;
    return n_active >= kSplitMinPairs || (kSplitMinPairs == 3 && n_active == 2 && 2 * ((H + kHypPerBlock - 1) / kHypPerBlock) > cus);
}


constexpr int kCnt32Threads = 768;
constexpr int kCnt32Slots = 4;     // hypotheses a wavefront carries at a time
constexpr int kCnt32Ppl = 4;       // single-precision counting: four points per lane and block (scalar work per evaluation halves)
static size_t count32_lds_bytes(int max_kp)
{
    const int bw = 64 * kCnt32Ppl;
    return (size_t)((max_kp + bw - 1) / bw) * bw * 4 * sizeof(float);
}

static size_t count_lds_bytes(int max_kp)
{
    const int bw = 64 * kCntPpl;
    return (size_t)((max_kp + bw - 1) / bw) * bw * 4 * sizeof(double);
}

bool kernel_desc(int id, int max_kp, int desc_words, KernelDesc *out)
{
    KernelDesc d{nullptr, nullptr, 0, 0};
    switch (id) {
    case kKMatchTopk:
        d.name = desc_words == 4 ? "match_topk_kernel<4>" : desc_words == 16 ? "match_topk_kernel<16>" : "match_mfma_kernel<true>";
        d.fn = desc_words == 4    ? reinterpret_cast<const void *>(match_topk_kernel<4>)
               : desc_words == 16 ? reinterpret_cast<const void *>(match_topk_kernel<16>)
                                  : reinterpret_cast<const void *>(match_mfma_kernel<true>);
        d.threads = desc_words == 8 ? kMmThreads : 1024;
        break;
    case kKMatchTopkVec:
        d.name = "match_topk_kernel<8>";
        d.fn = reinterpret_cast<const void *>(match_topk_kernel<8>);
        d.threads = 1024;
        break;
    case kKMatchCompact:
        d.name = "match_compact_kernel";
        d.fn = reinterpret_cast<const void *>(match_compact_kernel);
        d.threads = 1024;
        break;
    case kKRansacFused:
        d.name = "ransac_kernel<false, 1272>";
        d.fn = reinterpret_cast<const void *>(ransac_kernel<false, 248 + 1024>);
        d.threads = kHypPerBlock;
        break;
    case kKRansacSolve:
        d.name = "ransac_solve_list_kernel<1264>";   // (+ the one-workgroup mode0_list_kernel in front of it)
        d.fn = reinterpret_cast<const void *>(ransac_solve_list_kernel<240 + 1024>);
        d.threads = kSolveBlock;
        break;
    case kKRansacSelect:
        d.name = "ransac_select_kernel";
        d.fn = reinterpret_cast<const void *>(ransac_select_kernel);
        d.threads = kSelThreads;
        d.dynamic_lds = (size_t)max_kp * 4 * sizeof(double);
        break;
    case kKPairPrepare:
        d.name = "pair_prepare_kernel";
        d.fn = reinterpret_cast<const void *>(pair_prepare_kernel);
        d.threads = 256;
        break;
    case kKRansacPrescreen:
        d.name = "ransac_prescreen_kernel";
        d.fn = reinterpret_cast<const void *>(ransac_prescreen_kernel);
        d.threads = 64;
        break;
    case kKRansacExactList:
        d.name = "ransac_exact_list_kernel<1264>";
        d.fn = reinterpret_cast<const void *>(ransac_exact_list_kernel<240 + 1024>);
        d.threads = 64;
        break;
    case kKRansacCount2:
        d.name = "ransac_count2_kernel<768, 2>";
        d.fn = reinterpret_cast<const void *>(ransac_count2_kernel<kCntThreads, kCntPpl>);
        d.threads = kCntThreads;
        d.dynamic_lds = count_lds_bytes(max_kp);
        break;
    case kKRansacCountPilot:
        d.name = "ransac_finish_mfma_kernel<false, true>";   // the pilot
        d.fn = reinterpret_cast<const void *>(ransac_finish_mfma_kernel<false, true>);
        d.threads = kFinishThreads;
        d.dynamic_lds = (size_t)kDenseChunk * 64;
        break;
    case kKRansacCountFinish:
        d.name = "ransac_finish_mfma_kernel<false, false>";
        d.fn = reinterpret_cast<const void *>(ransac_finish_mfma_kernel<false>);
        d.threads = kFinishThreads;
        d.dynamic_lds = (size_t)kDenseChunk * 64;
        break;
    case kKRansacCountFinishRest:
        d.name = "ransac_finish_upper_kernel<false>";
        d.fn = reinterpret_cast<const void *>(ransac_finish_upper_kernel<false>);
        d.threads = kFinUpperThreads;
        d.dynamic_lds = (size_t)kFinUpperChunk * 64 + (size_t)(kFinUpperThreads / 64) * kDenseWin * 16;
        break;
    case kKRansacCountMfma:
        d.name = "ransac_count_mfma_kernel<false, 512, 4, 672>";
        d.fn = reinterpret_cast<const void *>(ransac_count_mfma_kernel<false, kDenseThreads, kDenseBatches>);
        d.threads = kDenseThreads;
        d.dynamic_lds = (size_t)kDenseChunkD * 64 + (size_t)(kDenseThreads / 64) * kDenseWin * 16;
        break;
    case kKRansacSurvivors:
        d.name = "ransac_survivors_kernel";
        d.fn = reinterpret_cast<const void *>(ransac_survivors_kernel);
        d.threads = 256;
        break;
    case kKFinModel:
        d.name = "finalize_model_kernel";
        d.fn = reinterpret_cast<const void *>(finalize_model_kernel);
        d.threads = kFinThreads;
        break;
    case kKTriangulate:
        d.name = "triangulate_kernel";
        d.fn = reinterpret_cast<const void *>(triangulate_kernel);
        d.threads = 256;
        break;
    case kKFinSelect:
        d.name = "finalize_select_kernel";
        d.fn = reinterpret_cast<const void *>(finalize_select_kernel);
        d.threads = kFinThreads;
        break;
#ifdef MVS_DEBUG_HOOKS   // kernels of the experiment ladder: ids behind kKernelCountProduct, diagnostics build only
    case kKRansacScore:
        d.name = "ransac_score_kernel";
        d.fn = reinterpret_cast<const void *>(ransac_score_kernel);
        d.threads = kHypPerBlock;
        break;
    case kKRansacCount:
        d.name = "ransac_count_kernel<768, 2>";
        d.fn = reinterpret_cast<const void *>(ransac_count_kernel<kCntThreads, kCntPpl>);
        d.threads = kCntThreads;
        d.dynamic_lds = count_lds_bytes(max_kp);
        break;
    case kKRansacCount32:
        d.name = "ransac_count32_kernel<768, 4, 4, 1, false>";
        d.fn = reinterpret_cast<const void *>(ransac_count32_kernel<kCnt32Threads, kCnt32Ppl, kCnt32Slots, 1>);
        d.threads = kCnt32Threads;
        d.dynamic_lds = count32_lds_bytes(max_kp);
        break;
#endif
    default: return false;
    }
    *out = d;
    return true;
}

// The opt-in to more than 64 KB of dynamic LDS is a per-device function attribute.  Called from mvs_ctx_create (once per
// context, i.e. per (thread, device)); the result is checked there, so a part without 160 KB of LDS fails at create time
// with a clear message instead of at the first launch.
hipError_t prepare_kernels()
{
    const void *fns[] = {
#ifdef MVS_DEBUG_HOOKS
                         reinterpret_cast<const void *>(ransac_count_kernel<kCntThreads, kCntPpl>),
                         reinterpret_cast<const void *>(ransac_count_kernel<kCntThreads, kCntPpl, true>),
                         reinterpret_cast<const void *>(ransac_count32_kernel<kCnt32Threads, kCnt32Ppl, kCnt32Slots, 1>),
                         reinterpret_cast<const void *>(ransac_count32_kernel<kCnt32Threads, kCnt32Ppl, kCnt32Slots, 1, true>),
                         reinterpret_cast<const void *>(ransac_count32_kernel<kCnt32Threads, kCnt32Ppl, kCnt32Slots, 2>),
                         reinterpret_cast<const void *>(ransac_count32_kernel<kCnt32Threads, kCnt32Ppl, kCnt32Slots, 2, true>),
#endif
                         reinterpret_cast<const void *>(ransac_count2_kernel<kCntThreads, kCntPpl>),
                         reinterpret_cast<const void *>(ransac_count2_kernel<kCntThreads, kCntPpl, true>),
                         reinterpret_cast<const void *>(ransac_count32_kernel<kCnt32Threads, kCnt32Ppl, kCnt32Slots, 0>),
                         reinterpret_cast<const void *>(ransac_count32_kernel<kCnt32Threads, kCnt32Ppl, kCnt32Slots, 0, true>),
                         reinterpret_cast<const void *>(ransac_count_mfma_kernel<false, kDenseThreads, kDenseBatches>),
                         reinterpret_cast<const void *>(ransac_count_mfma_kernel<true, kDenseThreads, kDenseBatches>),
#ifdef MVS_DEBUG_HOOKS
                         reinterpret_cast<const void *>(ransac_count_mfma_kernel<false, 256, 8, 768>),
                         reinterpret_cast<const void *>(ransac_count_mfma_kernel<false, 384, 6, 768>),
                         reinterpret_cast<const void *>(ransac_count_mfma_kernel<false, 512, 4, 768>),
                         reinterpret_cast<const void *>(ransac_count_mfma_kernel<false, 512, 4, 640>),
                         reinterpret_cast<const void *>(ransac_count_mfma_kernel<false, 512, 4, 672>),
                         reinterpret_cast<const void *>(ransac_count_mfma_kernel<false, 512, 2, 672>),
#endif
                         reinterpret_cast<const void *>(ransac_finish_mfma_kernel<false>),
                         reinterpret_cast<const void *>(ransac_finish_mfma_kernel<true>),
                         reinterpret_cast<const void *>(ransac_finish_mfma_kernel<false, true>),
                         reinterpret_cast<const void *>(ransac_finish_mfma_kernel<true, true>),
                         reinterpret_cast<const void *>(ransac_finish_upper_kernel<false>),
                         reinterpret_cast<const void *>(ransac_finish_upper_kernel<true>),
                         reinterpret_cast<const void *>(ransac_select_kernel)};
    for (const void *f : fns) {
        const hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, kMaxKp * 32);
        if (e != hipSuccess)
            return e;
    }
    return hipSuccess;
}

// The product library has ONE path and no process-global mutable state: the switches below are compile-time constants there.
// Only the diagnostics build (-DMVS_DEBUG_HOOKS: kernels_dbg.o -> libmvslam_hip_dbg.so) turns them into variables with
// setters, and only that build contains the experiment ladder's kernels (tools/ab_ransac.py, tests/prescreen_gpu_check.py).
#ifdef MVS_DEBUG_HOOKS
static int g_match_mfma = 1;   // 1 = by batch size (below); diagnostics: 0 = always the VALU kernel, 2 = always the matrix-core one
void set_match_mfma(int v) { g_match_mfma = v; }
#else
constexpr int g_match_mfma = 1;
#endif

// The cap of match_mfma_kernel<true>: the smallest integer distance C with ratio * C > max_dist, evaluated exactly as the
// kernel evaluates the ratio test (float distance -> double); -1 when the call has no distance limit (max_dist < 0), no usable
// ratio, or a limit so wide that random rows would reach it (Hamming distances of unrelated 256-bit descriptors are 128 +- 8).
static int match_cap_distance(double ratio, double max_dist)
{
    if (!(max_dist >= 0.0) || !(ratio > 0.0))
        return -1;
    for (int c = 0; c <= 96; ++c)
        if (ratio * (double)(float)c > max_dist)
            return c;
    return -1;
}

void launch_match_topk(const BatchDev &b, const RunParams &rp, int n_active, hipStream_t stream, LaunchTimer *lt)
{
    const dim3 grid((b.max_kp + 63) / 64, n_active), block(1024);
    const double ratio = rp.ratio, md = rp.max_dist;
    const int mm_groups8 = ((b.max_kp + kMmQueries - 1) / kMmQueries) * n_active;
    const bool mfma8 = b.desc_words == 8 && (g_match_mfma == 2 || (g_match_mfma == 1 && mm_groups8 >= 512));
    if (lt) lt->mark(b.desc_words == 8 && !mfma8 ? kKMatchTopkVec : kKMatchTopk);
    switch (b.desc_words) {
    case 4: hipLaunchKernelGGL(match_topk_kernel<4>, grid, block, 0, stream, b, ratio, md); break;
    case 8: {
        // the matrix-core kernel takes 256 queries per workgroup over all trains: throughput for a batch (0.55 against 1.05 ms
        // per 512 pairs), but a long serial walk when there are only a handful of workgroups -- one pair: 60 us against 22 for
        // the vector kernel with its 64 queries per workgroup.  Below two workgroups per CU the vector kernel runs.
        if (mfma8) {
            const dim3 mgrid((b.max_kp + kMmQueries - 1) / kMmQueries, n_active);
            const int cap = match_cap_distance(ratio, md);
            if (cap >= 0)
                hipLaunchKernelGGL(match_mfma_kernel<true>, mgrid, dim3(kMmThreads), 0, stream, b, ratio, md, cap);
            else
                hipLaunchKernelGGL(match_mfma_kernel<false>, mgrid, dim3(kMmThreads), 0, stream, b, ratio, md, 0);
        } else
            hipLaunchKernelGGL(match_topk_kernel<8>, grid, block, 0, stream, b, ratio, md);
        break;
    }
    case 16: hipLaunchKernelGGL(match_topk_kernel<16>, grid, block, 0, stream, b, ratio, md); break;
    default: break;
    }
}

void launch_match_compact(const BatchDev &b, const RunParams &, int n_active, hipStream_t stream, LaunchTimer *lt)
{
    if (lt) lt->mark(kKMatchCompact);
    hipLaunchKernelGGL(match_compact_kernel, dim3(n_active), dim3(1024), 0, stream, b);
}

void launch_prep_points(const BatchDev &b, const double *uv1, const double *uv2, int n_active, hipStream_t stream)
{
    hipLaunchKernelGGL(prep_points_kernel, dim3((b.max_kp + 255) / 256, n_active), dim3(256), 0, stream, b, uv1, uv2);
}

// 9000 (default, round 3) = the pre-screened stage: pair_prepare -> prescreen (+ the exact solve for pairs the probe sends
// there) -> exact solve of the hypotheses without a certificate -> count with per-hypothesis thresholds -> survivors ->
// their exact solve -> select.
// 120 fused; 632 = solve + hypothesis-per-lane scoring as two launches (round 1); 1784 = solve (with the sqrt-free
// convergence test, bit 128 of the kernel's VAR, for the 9x9 and -- kernel bit 1024 -- the 3x3 SVD) + pruned
// point-per-lane scoring (bit 1024 of the launch variant): ransac_count + ransac_select
#ifdef MVS_DEBUG_HOOKS
static int g_ransac_variant = 9000;
void set_ransac_variant(int v) { g_ransac_variant = v; }
int get_ransac_variant() { return g_ransac_variant; }
#endif

template <int VAR>
static void launch_ransac_var(const BatchDev &b, const RunParams &rp, dim3 grid, dim3 block, bool stats, hipStream_t stream,
                              LaunchTimer *lt)
{
    if (lt) lt->mark(kKRansacFused);
    if (stats)
        hipLaunchKernelGGL((ransac_kernel<true, VAR>), grid, block, 0, stream, b, rp);
    else
        hipLaunchKernelGGL((ransac_kernel<false, VAR>), grid, block, 0, stream, b, rp);
}

#ifdef MVS_DEBUG_HOOKS   // round 2's pruned scoring (variants 1656 / 1784 / 3832)
static void launch_pruned_scoring(const BatchDev &b, const RunParams &rp, int n_active, hipStream_t stream, LaunchTimer *lt,
                                  bool stats = false)
{
    // enough workgroups to fill the chip for a small launch, few enough that every wavefront works through many
    // groups of four hypotheses (the bound only helps once the first groups have finished)
    const int n_groups4 = (rp.num_hypotheses + kCntSlots - 1) / kCntSlots;
    const int wpw = kCntThreads / 64;
    int wg = (512 + n_active - 1) / n_active;
    wg = std::max(wg, 4);
    wg = std::min(wg, std::max(1, (n_groups4 + wpw - 1) / wpw));
    const size_t lds_cnt = count_lds_bytes(b.max_kp);
    const size_t lds_sel = (size_t)b.max_kp * 4 * sizeof(double);
    const dim3 grid(wg, n_active);
    if (lt) lt->mark(kKRansacCount);
    if (stats)
        hipLaunchKernelGGL((ransac_count_kernel<kCntThreads, kCntPpl, true>), grid, dim3(kCntThreads), lds_cnt, stream, b, rp, wg);
    else
        hipLaunchKernelGGL((ransac_count_kernel<kCntThreads, kCntPpl>), grid, dim3(kCntThreads), lds_cnt, stream, b, rp, wg);
    if (lt) lt->mark(kKRansacSelect);
    hipLaunchKernelGGL(ransac_select_kernel, dim3(n_active), dim3(kSelThreads), lds_sel, stream, b, rp, 0);
}

#endif

#ifdef MVS_DEBUG_HOOKS
static int g_count_dense = 1;   // 1 = pilot + dense matrix-core phase + matrix-core finish; diagnostics: 0 = one
                                // ransac_count32 launch, 2 = pilot + dense phase + the vector finish (ransac_count32, phase 2)
static int g_dense_margin = kDenseMargin;   // diagnostics: mvs_debug_set_count_dense(1000 + margin)
static int g_pilot_hyp = kPilotMfmaHyp;     // diagnostics: mvs_debug_set_count_dense(100000 + hypotheses of the pilot)
void set_count_dense(int v)
{
    if (v >= 100000)
        g_pilot_hyp = v - 100000;
    else if (v >= 1000)
        g_dense_margin = v - 1000;
    else
        g_count_dense = v;
}
static int g_force_mode = -1;   // diagnostics: -1 = the probe decides, 0 / 1 = every pair exact / pre-screened
void set_prescreen_force(int m) { g_force_mode = m; }

void launch_prescreen_only(const BatchDev &b, const RunParams &rp, int n_active, int mode, hipStream_t stream)
{
    hipLaunchKernelGGL(pair_prepare_kernel, dim3(n_active), dim3(256), 0, stream, b, rp, mode);
    hipLaunchKernelGGL(ransac_prescreen_kernel, dim3((rp.num_hypotheses + 63) / 64, n_active), dim3(64), 0, stream, b, rp);
}
#else
constexpr int g_count_dense = 1;
constexpr int g_dense_margin = kDenseMargin;
constexpr int g_pilot_hyp = kPilotMfmaHyp;
constexpr int g_force_mode = -1;
#endif

// the counting part of the pre-screened stage: upper / lower count bounds of every record, the pair's bound (mode 1: pilot ->
// dense matrix-core phase -> sorted list -> matrix-core finish; modes 0 / 2: ransac_count2_kernel)
static void launch_counting(const BatchDev &b, const RunParams &rp, int n_active, hipStream_t stream, LaunchTimer *lt, bool stats)
{
    const int H = rp.num_hypotheses;
    const int n_groups4 = (H + kCntSlots - 1) / kCntSlots;
    const int wpw = kCntThreads / 64;
    int wg = (512 + n_active - 1) / n_active;
    wg = std::max(wg, 4);
    wg = std::min(wg, std::max(1, (n_groups4 + wpw - 1) / wpw));
    const size_t lds_cnt = count_lds_bytes(b.max_kp);
#ifdef MVS_DEBUG_HOOKS
    const size_t lds_c32 = count32_lds_bytes(b.max_kp);   // the vector counting kernels are diagnostics-only since round 4
#endif
    // counting: single precision for the pairs in mode 1, double precision for the others (each launch's workgroups leave
    // at once for the pairs of the other kind)
    // single-precision counting of the pairs in mode 1, three launches: pilot (the first 256 hypotheses in full -> the pair's
    // first bound) -> dense phase on the matrix cores in split bf16 (every hypothesis x the points that must be seen before
    // anything can be dropped, no exit tests) -> finish (the listed hypotheses that can still reach the bound, from there on).
    // g_count_dense = 0 (diagnostics library only): ransac_count32_kernel over everything in one launch -- byte-identical
    // results, 5.2 ms against 0.2 + 2.0 + 1.4 per 512 pairs (profiles/r03_count32_experiments.md).
#ifdef MVS_DEBUG_HOOKS
    if (!g_count_dense) {
        if (lt) lt->mark(kKRansacCount32);
        if (stats)
            hipLaunchKernelGGL((ransac_count32_kernel<kCnt32Threads, kCnt32Ppl, kCnt32Slots, 1, true>), dim3(wg, n_active),
                               dim3(kCnt32Threads), lds_c32, stream, b, rp, wg);
        else
            hipLaunchKernelGGL((ransac_count32_kernel<kCnt32Threads, kCnt32Ppl, kCnt32Slots, 1>), dim3(wg, n_active),
                               dim3(kCnt32Threads), lds_c32, stream, b, rp, wg);
    } else
#endif
    {
        if (lt) lt->mark(kKRansacCountPilot);
        const dim3 pilot_grid(n_active, (std::min(H, g_pilot_hyp) + kFinishThreads - 1) / kFinishThreads);
#ifdef MVS_DEBUG_HOOKS
        if (g_count_dense == 3) {   // diagnostics: round 3's vector pilot over the first 256 hypotheses
            const int wg_pilot = std::max(1, std::min(wg, (kPilotHyp / kCnt32Slots) / (kCnt32Threads / 64)));
            hipLaunchKernelGGL((ransac_count32_kernel<kCnt32Threads, kCnt32Ppl, kCnt32Slots, 0>), dim3(wg_pilot, n_active),
                               dim3(kCnt32Threads), lds_c32, stream, b, rp, wg_pilot);
        } else
#endif
        if (stats)
            hipLaunchKernelGGL((ransac_finish_mfma_kernel<true, true>), pilot_grid, dim3(kFinishThreads), (size_t)kDenseChunk * 64,
                               stream, b, rp, 0);
        else
            hipLaunchKernelGGL((ransac_finish_mfma_kernel<false, true>), pilot_grid, dim3(kFinishThreads), (size_t)kDenseChunk * 64,
                               stream, b, rp, 0);
        if (lt) lt->mark(kKRansacCountMfma);
        auto dense = [&](auto kern, int threads, int batches, int chunk = kDenseChunkD) {
            const size_t lds = (size_t)chunk * 64 + (size_t)(threads / 64) * kDenseWin * 16;
            const dim3 grid(((H + threads - 1) / threads + batches - 1) / batches, n_active);
            hipLaunchKernelGGL(kern, grid, dim3(threads), lds, stream, b, rp, g_dense_margin);
        };
#ifdef MVS_DEBUG_HOOKS
        // A/B of the dense phase's shape: mvs_debug_set_count_dense(10 + k)
        if (g_count_dense == 10) dense(ransac_count_mfma_kernel<false, 256, 8, 768>, 256, 8, 768);
        else if (g_count_dense == 11) dense(ransac_count_mfma_kernel<false, 384, 6, 768>, 384, 6, 768);
        else if (g_count_dense == 15) dense(ransac_count_mfma_kernel<false, 512, 4, 768>, 512, 4, 768);
        else if (g_count_dense == 16) dense(ransac_count_mfma_kernel<false, 512, 4, 640>, 512, 4, 640);
        else if (g_count_dense == 17) dense(ransac_count_mfma_kernel<false, 512, 4, 672>, 512, 4, 672);
        else if (g_count_dense == 18) dense(ransac_count_mfma_kernel<false, 512, 2, 672>, 512, 2, 672);
        else
#endif
        if (stats)
            dense(ransac_count_mfma_kernel<true, kDenseThreads, kDenseBatches>, kDenseThreads, kDenseBatches);
        else
            dense(ransac_count_mfma_kernel<false, kDenseThreads, kDenseBatches>, kDenseThreads, kDenseBatches);
        if (lt) lt->mark(kKRansacCountFinish);
        const dim3 fin_grid(n_active, (H + kFinishThreads - 1) / kFinishThreads);   // workgroups past the list's end leave at once
        if (g_count_dense != 2)
            hipLaunchKernelGGL(ransac_list_sort_kernel, dim3(n_active), dim3(kSortThreads), 0, stream, b);
#ifdef MVS_DEBUG_HOOKS
        if (g_count_dense == 2) {   // diagnostics: the vector finish (ransac_count32_kernel, phase 2)
            hipLaunchKernelGGL((ransac_count32_kernel<kCnt32Threads, kCnt32Ppl, kCnt32Slots, 2>), dim3(wg, n_active),
                               dim3(kCnt32Threads), lds_c32, stream, b, rp, wg);
        } else
#endif
        {
            // two launches: the first batch of every pair (the 256 largest partial counts: the winner is nearly always among
            // them, so the pair's bound is final afterwards: upper and lower bounds over every point), then the rest of the
            // list: upper counts only, over the points behind the dense phase
            if (stats)
                hipLaunchKernelGGL(ransac_finish_mfma_kernel<true>, dim3(n_active, 1), dim3(kFinishThreads), (size_t)kDenseChunk * 64,
                                   stream, b, rp, 0);
            else
                hipLaunchKernelGGL(ransac_finish_mfma_kernel<false>, dim3(n_active, 1), dim3(kFinishThreads), (size_t)kDenseChunk * 64,
                                   stream, b, rp, 0);
            if (fin_grid.y > 1) {
                const size_t lds_up = (size_t)kFinUpperChunk * 64 + (size_t)(kFinUpperThreads / 64) * kDenseWin * 16;
                if (lt) lt->mark(kKRansacCountFinishRest);
                if (stats)
                    hipLaunchKernelGGL(ransac_finish_upper_kernel<true>, dim3(n_active, kFinUpperWg), dim3(kFinUpperThreads), lds_up,
                                       stream, b, rp);
                else
                    hipLaunchKernelGGL(ransac_finish_upper_kernel<false>, dim3(n_active, kFinUpperWg), dim3(kFinUpperThreads), lds_up,
                                       stream, b, rp);
            }
        }
    }
    if (lt) lt->mark(kKRansacCount2);
    if (stats)
        hipLaunchKernelGGL((ransac_count2_kernel<kCntThreads, kCntPpl, true>), dim3(wg, n_active), dim3(kCntThreads), lds_cnt,
                           stream, b, rp, wg);
    else
        hipLaunchKernelGGL((ransac_count2_kernel<kCntThreads, kCntPpl>), dim3(wg, n_active), dim3(kCntThreads), lds_cnt, stream,
                           b, rp, wg);
}

// the pre-screened RANSAC stage (variant 9000)
static void launch_prescreened(const BatchDev &b, const RunParams &rp, int n_active, hipStream_t stream, LaunchTimer *lt,
                               bool stats)
{
    const int H = rp.num_hypotheses;
    const int G = (H + kHypPerBlock - 1) / kHypPerBlock;
    if (lt) lt->mark(kKPairPrepare);
    hipLaunchKernelGGL(pair_prepare_kernel, dim3(n_active), dim3(256), 0, stream, b, rp, g_force_mode);
    if (lt) lt->mark(kKRansacPrescreen);
    hipLaunchKernelGGL(ransac_prescreen_kernel, dim3((H + 63) / 64, n_active), dim3(64), 0, stream, b, rp);
    // pairs the probe did not certify: every hypothesis through the exact solve, a persistent grid over the list of those pairs
    if (lt) lt->mark(kKRansacSolve);
    hipLaunchKernelGGL(mode0_list_kernel, dim3(1), dim3(256), 0, stream, b, n_active);
    static_assert(kSolveBlock == 64, "ransac_solve_list_kernel takes blocks of 64 hypotheses");
    hipLaunchKernelGGL((ransac_solve_list_kernel<240 + 1024>), dim3(2048), dim3(64), 0, stream, b, rp,
                       G * (kHypPerBlock / kSolveBlock));
    launch_counting(b, rp, n_active, stream, lt, stats);
    if (lt) lt->mark(kKRansacSurvivors);
    hipLaunchKernelGGL(ransac_survivors_kernel, dim3(kSurvWg, n_active), dim3(256), 0, stream, b, rp, n_active);
    // one exact solve over the whole work list: what the pre-screen flagged + the survivors of the counting
    if (lt) lt->mark(kKRansacExactList);
    hipLaunchKernelGGL((ransac_exact_list_kernel<240 + 1024>), dim3(1024), dim3(64), 0, stream, b, rp, 0);
    if (lt) lt->mark(kKRansacSelect);
    hipLaunchKernelGGL(ransac_select_kernel, dim3(n_active), dim3(kSelThreads), (size_t)b.max_kp * 4 * sizeof(double), stream,
                       b, rp, 1);
}

// the pre-screened stage, or -- for one or two pairs, the per-hypothesis tables and the instrumented replay's rotation
// counters -- the fused hypothesis-per-lane kernel
static void launch_ransac_product(const BatchDev &b, const RunParams &rp, int n_active, bool stats, hipStream_t stream,
                                  LaunchTimer *lt, dim3 grid, dim3 block)
{
    const bool split_ok = !stats && b.hyp_F && prescreened_launch(b, n_active, rp.num_hypotheses);
    if (!split_ok || b.hyp_count) {
        if (!stats)
            launch_ransac_var<248 + 1024>(b, rp, grid, block, false, stream, lt);
        else
            launch_ransac_var<120>(b, rp, grid, block, true, stream, lt);
        if (stats && b.hyp_F && !b.hyp_count && prescreened_launch(b, n_active, rp.num_hypotheses))
            launch_prescreened(b, rp, n_active, stream, nullptr, true);   // + the product path's own counters
    } else {
        launch_prescreened(b, rp, n_active, stream, lt, false);
    }
}

void launch_ransac(const BatchDev &b, const RunParams &rp, int n_active, bool stats, hipStream_t stream, LaunchTimer *lt)
{
    const int G = (rp.num_hypotheses + kHypPerBlock - 1) / kHypPerBlock;
    const dim3 grid(G, n_active), block(kHypPerBlock);
#ifndef MVS_DEBUG_HOOKS
    launch_ransac_product(b, rp, n_active, stats, stream, lt, grid, block);
#else
    // diagnostics build: the experiment ladder of rounds 1-2 behind mvs_debug_set_ransac_variant (tools/ab_ransac.py)
    const bool split_ok = !stats && b.hyp_F && prescreened_launch(b, n_active, rp.num_hypotheses);
    switch (g_ransac_variant) {
    case 9000: launch_ransac_product(b, rp, n_active, stats, stream, lt, grid, block); break;
    case 0: launch_ransac_var<0>(b, rp, grid, block, stats, stream, lt); break;
    case 376: launch_ransac_var<376>(b, rp, grid, block, stats, stream, lt); break;   // timing experiment: no V rotations
    case 632:
        if (!split_ok) {
            launch_ransac_var<120>(b, rp, grid, block, stats, stream, lt);
        } else {
            hipLaunchKernelGGL((ransac_solve_kernel<112>), grid, block, 0, stream, b, rp, 0);
            hipLaunchKernelGGL(ransac_score_kernel, grid, block, 0, stream, b, rp);
        }
        break;
    case 760:   // 632 + sqrt-free convergence test
        if (!split_ok) {
            launch_ransac_var<120>(b, rp, grid, block, stats, stream, lt);
        } else {
            hipLaunchKernelGGL((ransac_solve_kernel<240>), grid, block, 0, stream, b, rp, 0);
            hipLaunchKernelGGL(ransac_score_kernel, grid, block, 0, stream, b, rp);
        }
        break;
    case 3832:  // 1784 with the solve as A / V wavefront pairs
        if (!split_ok || b.hyp_count) {
            launch_ransac_var<120>(b, rp, grid, block, stats, stream, lt);
        } else {
            hipLaunchKernelGGL((ransac_solve_av_kernel<240>), grid, dim3(512), 0, stream, b, rp);
            launch_pruned_scoring(b, rp, n_active, stream, lt);
        }
        break;
    case 1656:  // 632 + pruned scoring
    case 1784:  // 760 + pruned scoring (+ the 3x3 SVD on the unscaled sequences): the default
        if (!split_ok) {
            // one or two pairs, per-hypothesis tables: the fused kernel (with the same sqrt-free pair step for 1784);
            // the instrumented replay stays on variant 120, whose counters the flop model was derived with
            if (g_ransac_variant == 1784 && !stats)
                launch_ransac_var<248 + 1024>(b, rp, grid, block, false, stream, lt);
            else
                launch_ransac_var<120>(b, rp, grid, block, stats, stream, lt);
            if (stats && b.hyp_F && !b.hyp_count && prescreened_launch(b, n_active, rp.num_hypotheses) && g_ransac_variant == 1784) {
                // the instrumented replay also runs the product path once with the counting kernel's evaluation counter
                // (stats[2]): the roofline quotes EXECUTED evaluations for the pruned kernel, not the H x M it avoids
                hipLaunchKernelGGL((ransac_solve_kernel<240 + 1024>), dim3(G * (kHypPerBlock / kSolveBlock), n_active),
                                   dim3(kSolveBlock), 0, stream, b, rp, 0);
                launch_pruned_scoring(b, rp, n_active, stream, nullptr, true);
            }
        } else {
            if (lt) lt->mark(kKRansacSolve);
            if (g_ransac_variant == 1784)
                hipLaunchKernelGGL((ransac_solve_kernel<240 + 1024>), dim3(G * (kHypPerBlock / kSolveBlock), n_active),
                                   dim3(kSolveBlock), 0, stream, b, rp, 0);
            else
                hipLaunchKernelGGL((ransac_solve_kernel<112>), grid, block, 0, stream, b, rp, 0);
            if (b.hyp_count) {
                if (lt) lt->mark(kKRansacScore);
                hipLaunchKernelGGL(ransac_score_kernel, grid, block, 0, stream, b, rp);
            } else {
                launch_pruned_scoring(b, rp, n_active, stream, lt);
            }
        }
        break;
    default: launch_ransac_var<120>(b, rp, grid, block, stats, stream, lt); break;
    }
#endif
}

void launch_finalize(const BatchDev &b, const RunParams &rp, int n_active, int mode, hipStream_t stream, LaunchTimer *lt)
{
    if (lt) lt->mark(kKFinModel);
    hipLaunchKernelGGL(finalize_model_kernel, dim3(n_active), dim3(kFinThreads), 0, stream, b, rp, mode);
    if (lt) lt->mark(kKTriangulate);
    hipLaunchKernelGGL(triangulate_kernel, dim3((b.max_kp + 255) / 256, n_active), dim3(256), 0, stream, b);
    if (lt) lt->mark(kKFinSelect);
    hipLaunchKernelGGL(finalize_select_kernel, dim3(n_active), dim3(kFinThreads), 0, stream, b);
}

// full-population audit, worst-case-construction probes (diagnostics build only: tests/audit_gpu_check.py,
// tests/constants_gpu_check.py)
#ifdef MVS_DEBUG_HOOKS
#include "kernels_audit.hip"
#endif


// ---- single-shot glue (the reference's one-pair-at-a-time call pattern: front-end/image-pair.cpp:30-71) ----------------------
// The per-pair scalars of pair 0 travel as KERNEL ARGUMENTS instead of five small host-to-device copies, and the outputs of
// pair 0 are gathered into one contiguous block for ONE device-to-host copy instead of five: every copy command costs the
// stream several microseconds that a 0.25 ms call notices (shim ImagePair ctor 0.38 -> 0.36 ms with the faster kernels of
// round 4; see profiles/r04_image_pair_latency.json).
__global__ __launch_bounds__(256) void single_params_kernel(BatchDev b, SingleParams sp, const uint4 *in)
{
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t == 0) {
        const_cast<int32_t *>(b.n1)[0] = sp.n1;
        const_cast<int32_t *>(b.n2)[0] = sp.n2;
        const_cast<int64_t *>(b.gidx)[0] = sp.gidx;
    }
    if (t < 9) {
        const_cast<double *>(b.K)[t] = sp.K[t];
        const_cast<double *>(b.Kinv)[t] = sp.Kinv[t];
    }
    if (in) {
        // the pair's descriptors and keypoints arrived as ONE block [desc1 | desc2 | kp1 | kp2] (16-byte aligned parts)
        uint4 *dst[4] = {reinterpret_cast<uint4 *>(const_cast<uint32_t *>(b.desc1)), reinterpret_cast<uint4 *>(const_cast<uint32_t *>(b.desc2)),
                         reinterpret_cast<uint4 *>(const_cast<float *>(b.kp1)), reinterpret_cast<uint4 *>(const_cast<float *>(b.kp2))};
        size_t off = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const size_t n16 = sp.part_bytes[k] / 16;   // parts are multiples of 8 bytes: whole 16-byte words, then one half
            for (size_t i = t; i < n16; i += (size_t)gridDim.x * 256)
                dst[k][i] = in[off + i];
            if (t == 0 && (sp.part_bytes[k] & 8u))
                reinterpret_cast<uint2 *>(dst[k] + n16)[0] = reinterpret_cast<const uint2 *>(in + off + n16)[0];
            off += (sp.part_bytes[k] + 15) / 16;
        }
    }
}
void launch_single_params(const BatchDev &b, const SingleParams &sp, const void *in, hipStream_t stream)
{
    hipLaunchKernelGGL(single_params_kernel, dim3(in ? 32 : 1), dim3(256), 0, stream, b, sp, reinterpret_cast<const uint4 *>(in));
}
// out: [result, 512 B][mask rows][points rows x 3 f64][point_idx rows x i32][matches rows x 16 B], each part 16-byte aligned
// (single_layout); parts with a zero flag are skipped
__global__ __launch_bounds__(256) void single_gather_kernel(BatchDev b, int rows, int flags, unsigned char *out)
{
    const SingleLayout L = single_layout(rows, flags);
    const int tid = blockIdx.x * 256 + threadIdx.x, nth = gridDim.x * 256;
    const uint32_t *r32 = reinterpret_cast<const uint32_t *>(b.results);
    for (int i = tid; i < (int)(sizeof(mvs_pair_result) / 4); i += nth)
        reinterpret_cast<uint32_t *>(out)[i] = r32[i];
    if (flags & 1)
        for (int i = tid; i < rows; i += nth)
            out[L.mask + i] = b.mask[i];
    if (flags & 2)
        for (int i = tid; i < rows * 3; i += nth)
            reinterpret_cast<double *>(out + L.points)[i] = b.points[i];
    if (flags & 4)
        for (int i = tid; i < rows; i += nth)
            reinterpret_cast<int32_t *>(out + L.idx)[i] = b.point_idx[i];
    if (flags & 8)
        for (int i = tid; i < rows; i += nth)
            reinterpret_cast<mvs_match *>(out + L.matches)[i] = b.matches[i];
}
void launch_single_gather(const BatchDev &b, int rows, int flags, unsigned char *out, hipStream_t stream)
{
    hipLaunchKernelGGL(single_gather_kernel, dim3(16), dim3(256), 0, stream, b, rows, flags, out);
}

void launch_fundamental(const double *p1, const double *p2, double *F, int *ok, hipStream_t stream)
{
    hipLaunchKernelGGL(fundamental_kernel, dim3(1), dim3(64), 0, stream, p1, p2, F, ok);
}

}  // namespace mvs
