// splat_kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels for the rasterisation hot
// path of thomasantony/splat.  Compiled with -ffp-contract=off: the reference (Rust/nalgebra)
// never fuses a*b+c, and image parity depends on the f32 operation order.
//
//   K0 cov3d_kernel       compute_cov3d                 src/gaussians.rs:101-113, 446-462
//   K1 preprocess_kernel  Pipeline::vertex, once/Gaussian src/pipelines.rs:96-125, 17-51;
//                                                       src/gaussians.rs:40-99, 114-161
//   -- scan_kernel        per-(tile,sub-bucket) exclusive offsets
//   K2 emit_kernel        instance expansion            src/pipelines.rs:69-79 (+ euc bbox clamp)
//   K3 sort_tiles_*       per-tile painter's order      src/gaussians.rs:302-303 (stable asc. z)
//   K4 composite_kernel   euc raster + fragment + blend src/pipelines.rs:127-168
#include <algorithm>
#include <cstddef>
#include <cstdlib>
#include <type_traits>

#include "splat_internal.h"

namespace splat {

// ---------------------------------------------------------------------------
// small column-major matrix helpers; products accumulate left to right like
// nalgebra's gemv-per-column (y_i = ((A_i0 x_0 + A_i1 x_1) + A_i2 x_2) + ...)
// ---------------------------------------------------------------------------
struct Mat3 { float m[9]; };
#define M3(A, r, c) ((A).m[(c) * 3 + (r)])

__device__ __forceinline__ Mat3 mat3_mul(const Mat3& a, const Mat3& b) {
    Mat3 c;
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            float acc = M3(a, i, 0) * M3(b, 0, j);
            acc = M3(a, i, 1) * M3(b, 1, j) + acc;
            acc = M3(a, i, 2) * M3(b, 2, j) + acc;
            M3(c, i, j) = acc;
        }
    return c;
}
__device__ __forceinline__ Mat3 mat3_t(const Mat3& a) {
    Mat3 t;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) M3(t, i, j) = M3(a, j, i);
    return t;
}
__device__ __forceinline__ void mat4_vec(const float* m, float x, float y, float z, float w, float out[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float acc = m[0 + i] * x;
        acc = m[4 + i] * y + acc;
        acc = m[8 + i] * z + acc;
        acc = m[12 + i] * w + acc;
        out[i] = acc;
    }
}
__device__ __forceinline__ bool finitef(float v) { return fabsf(v) <= 3.402823466e+38f; }

// Wave64 reductions / scans on the DPP path (row shifts inside the 16-lane rows, then the two row broadcasts of gfx9):
// six VALU instructions and no LDS traffic, where __shfl_xor / __shfl_up are a ds_bpermute plus address arithmetic per
// step.  K1 reduces a bounding box and scans a tile count per wave.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_or_identity(int identity, int v) { return __builtin_amdgcn_update_dpp(identity, v, CTRL, ROW_MASK, 0xf, false); }
// op over the whole wave, valid in lane 63 only (read it with wave_last)
template <typename Op>
__device__ __forceinline__ int wave_reduce_to_last(int v, int identity, Op op) {
    v = op(v, dpp_or_identity<0x111, 0xf>(identity, v));     // row_shr:1
    v = op(v, dpp_or_identity<0x112, 0xf>(identity, v));     // row_shr:2
    v = op(v, dpp_or_identity<0x114, 0xf>(identity, v));     // row_shr:4
    v = op(v, dpp_or_identity<0x118, 0xf>(identity, v));     // row_shr:8   -> lane 15 of every row holds its row
    v = op(v, dpp_or_identity<0x142, 0xa>(identity, v));     // row_bcast:15 into rows 1 and 3
    v = op(v, dpp_or_identity<0x143, 0xc>(identity, v));     // row_bcast:31 into rows 2 and 3 -> lane 63 holds the wave
    return v;
}
__device__ __forceinline__ int wave_last(int v) { return __builtin_amdgcn_readlane(v, 63); }
// inclusive prefix sum over the lanes of the wave
__device__ __forceinline__ unsigned int wave_inclusive_sum(unsigned int x) {
    int v = (int)x;
    v += dpp_or_identity<0x111, 0xf>(0, v);
    v += dpp_or_identity<0x112, 0xf>(0, v);
    v += dpp_or_identity<0x114, 0xf>(0, v);
    v += dpp_or_identity<0x118, 0xf>(0, v);
    v += dpp_or_identity<0x142, 0xa>(0, v);
    v += dpp_or_identity<0x143, 0xc>(0, v);
    return (unsigned int)v;
}

// ---------------------------------------------------------------------------
// Scene packing (upload time).  GaussianList buffers -> 16 planes of float4, plane p of
// Gaussian i at planes[p*n + i] so every per-frame load is a coalesced 16 B/lane stream.
// Float slots: 0-2 xyz, 3 opacity, 4-12 cov3d (col-major 3x3, all nine: the reference's
// R*S*R^T is not bit-symmetric), 13-39 sh[0..27), 40-60 sh[27..48), 61-63 zero.
// ---------------------------------------------------------------------------
// Gaussian slot j holds original Gaussian perm[j]: the scene is stored in Morton order of position
// so that consecutive threads project to neighbouring tiles (binning aggregates in LDS).
__global__ __launch_bounds__(256) void pack_scene_kernel(uint64_t n, const float* __restrict__ pos4,
                                                         const float* __restrict__ cov3d,
                                                         const float* __restrict__ opacity,
                                                         const float* __restrict__ sh,
                                                         const unsigned int* __restrict__ perm,
                                                         float4* __restrict__ planes) {
    uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const uint64_t i = perm[j];
    float F[64];
    F[0] = pos4[4 * i + 0]; F[1] = pos4[4 * i + 1]; F[2] = pos4[4 * i + 2]; F[3] = opacity[i];
#pragma unroll
    for (int k = 0; k < 9; ++k) F[4 + k] = cov3d[9 * i + k];
#pragma unroll
    for (int k = 0; k < 48; ++k) F[13 + k] = sh[48 * i + k];
    F[61] = F[62] = F[63] = 0.0f;
#pragma unroll
    for (int p = 0; p < SCENE_PLANES; ++p) planes[(uint64_t)p * n + j] = make_float4(F[4 * p], F[4 * p + 1], F[4 * p + 2], F[4 * p + 3]);
}

// K0 -- compute_cov3d, src/gaussians.rs:101-113.  rot = (i,j,k,w).
__global__ __launch_bounds__(256) void cov3d_kernel(uint64_t n, const float* __restrict__ scales3,
                                                    const float* __restrict__ rot4, float* __restrict__ out) {
    uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    float q0 = rot4[4 * t], q1 = rot4[4 * t + 1], q2 = rot4[4 * t + 2], q3 = rot4[4 * t + 3];
    // UnitQuaternion::from_quaternion: q / |q|, nalgebra 4-vector dot = (q0q0+q2q2)+(q1q1+q3q3)
    float a = q0 * q0, b = q1 * q1, c = q2 * q2, d = q3 * q3;
    a += c; b += d;
    float nrm = sqrtf(a + b);
    float i = q0 / nrm, j = q1 / nrm, k = q2 / nrm, w = q3 / nrm;
    float ww = w * w, ii = i * i, jj = j * j, kk = k * k;
    float ij = i * j * 2.0f, wk = w * k * 2.0f, wj = w * j * 2.0f;
    float ik = i * k * 2.0f, jk = j * k * 2.0f, wi = w * i * 2.0f;
    Mat3 R, S;
    M3(R, 0, 0) = ww + ii - jj - kk; M3(R, 0, 1) = ij - wk;           M3(R, 0, 2) = wj + ik;
    M3(R, 1, 0) = wk + ij;           M3(R, 1, 1) = ww - ii + jj - kk; M3(R, 1, 2) = jk - wi;
    M3(R, 2, 0) = ik - wj;           M3(R, 2, 1) = wi + jk;           M3(R, 2, 2) = ww - ii - jj + kk;
#pragma unroll
    for (int e = 0; e < 9; ++e) S.m[e] = 0.0f;
    float s0 = scales3[3 * t], s1 = scales3[3 * t + 1], s2 = scales3[3 * t + 2];
    M3(S, 0, 0) = s0 * s0; M3(S, 1, 1) = s1 * s1; M3(S, 2, 2) = s2 * s2;
    Mat3 cov = mat3_mul(mat3_mul(R, S), mat3_t(R));
#pragma unroll
    for (int e = 0; e < 9; ++e) out[9 * t + e] = cov.m[e];
}

// Exactly covered pixel interval {p in [lo_lim,hi_lim] : |p + off - c| <= h}; false when empty.
__device__ __forceinline__ bool covered_interval(float c, float h, float off, int lo_lim, int hi_lim, int* lo, int* hi) {
    float flo = c - h - off, fhi = c + h - off;
    if (!(fhi >= (float)lo_lim - 2.0f) || !(flo <= (float)hi_lim + 2.0f)) return false;
    int a = (int)fmaxf(floorf(flo) - 1.0f, (float)lo_lim);
    int b = (int)fminf(ceilf(fhi) + 1.0f, (float)hi_lim);
    // (the candidates start at most a few pixels outside the interval: two or three trips each.  Left to itself the
    // compiler vectorises these search loops eight candidates wide -- ~150 instructions per loop, a quarter of K1)
#pragma clang loop vectorize(disable) interleave(disable) unroll(disable)
    while (a <= b && !(fabsf(((float)a + off) - c) <= h)) ++a;
#pragma clang loop vectorize(disable) interleave(disable) unroll(disable)
    while (b >= a && !(fabsf(((float)b + off) - c) <= h)) --b;
    if (a > b) return false;
    *lo = a; *hi = b;
    return true;
}

// SH basis constants, src/gaussians.rs:11-26
#define SH_C0 0.28209479177387814f
#define SH_C1 0.4886025119029199f
#define SH_C2_0 1.0925484305920792f
#define SH_C2_1 (-1.0925484305920792f)
#define SH_C2_2 0.31539156525252005f
#define SH_C2_3 (-1.0925484305920792f)
#define SH_C2_4 0.5462742152960396f
#define SH_C3_0 (-0.5900435899266435f)
#define SH_C3_1 2.890611442640554f
#define SH_C3_2 (-0.4570457994644658f)
#define SH_C3_3 0.3731763325901154f
#define SH_C3_4 (-0.4570457994644658f)
#define SH_C3_5 1.445305721320277f
#define SH_C3_6 (-0.5900435899266435f)

// ---------------------------------------------------------------------------
// Block-cooperative binning shared by K1 (count) and K2 (emit).
// The scene is stored in Morton order, so the 256 Gaussians of a block overlap a small set of
// tiles.  Their (Gaussian, tile) pairs are first accumulated in an LDS table over the block's
// tile bounding box, then ONE global atomic per touched tile adds the count (K1) or reserves a
// slot range (K2).  Gaussians with > AGG_MAX_TILES tiles (close-ups) are expanded by the whole
// block, one tile per thread, straight on the global counters.
// ---------------------------------------------------------------------------
constexpr int AGG_CAP = 4096;          // LDS table entries (16 KB)
constexpr int AGG_MAX_TILES = 64;

constexpr int BIG_CAP = 256;           // close-up Gaussians handled per block-round

template <int CAP>
struct BinSharedT {
    static constexpr int cap = CAP;
    // The close-up arrays share the table's memory: close-ups are expanded after the table has done its work
    // (a workgroup barrier separates the two uses), and 6 KB less LDS per workgroup is room for another
    // kernel's workgroups on the CU (the compositor of the previous frame runs beside K1).
    union {
        unsigned int table[CAP];
        struct {
            unsigned int big[BIG_CAP][4];  // the block's big rectangles: x0 | y0 << 16, width, tile count, first flat index
            unsigned long long bigkey[BIG_CAP];
        };
    };
    int box[4];                        // min tx, min ty, max tx, max ty of the aggregated Gaussians
    unsigned int nbig;
    unsigned int nvis, nsing;          // block totals for the frame statistics (BucketBinner)
    unsigned int bigtotal;             // tiles of all big rectangles together
};
using BinShared = BinSharedT<AGG_CAP>;       // two-pass binning: dense table over the block's tile bounding box
// One-pass binning hashes tile coordinates into a HASH_DIM x HASH_DIM table (see BucketBinner).  32 x 32 tiles =
// 512 x 512 pixels covers the footprint of 256 Morton neighbours at every pose measured; a block that does not
// fit places its pairs one atomic each.  Against 64 x 64: a quarter of the table to clear and to scan per block,
// four reservations per thread to keep in registers instead of sixteen, 10 KB less LDS per workgroup.
#ifndef SPLAT_HASH_BITS
#define SPLAT_HASH_BITS 5
#endif
constexpr int HASH_BITS = SPLAT_HASH_BITS, HASH_DIM = 1 << HASH_BITS, HASH_CAP = HASH_DIM * HASH_DIM;
using BucketShared = BinSharedT<(HASH_CAP * 4 > BIG_CAP * 24 ? HASH_CAP : BIG_CAP * 6)>;   // (the close-up arrays need 6 KB)

// Close-ups (more than AGG_MAX_TILES tiles): the tiles of ALL the block's big rectangles form one
// flat index space that the 256 threads stride over, so every returning global atomic of the
// block is in flight at once (rectangle by rectangle it was one memory round trip per rectangle:
// 0.6-13 k cycles per block on C3, and unbounded for a camera inside the scene).
// Call with sh.nbig rectangles stored (add_big) and a barrier passed; ends without a barrier.
template <typename SH>
__device__ __forceinline__ void add_big(SH& sh, int tx0, int tx1, int ty0, int ty1, unsigned long long key) {
    const unsigned int k = atomicAdd(&sh.nbig, 1u);
    const unsigned int w = (unsigned int)(tx1 - tx0 + 1);
    sh.big[k][0] = (unsigned int)tx0 | ((unsigned int)ty0 << 16); sh.big[k][1] = w;
    sh.big[k][2] = w * (unsigned int)(ty1 - ty0 + 1);
    sh.bigkey[k] = key;
}
template <typename SH, typename F>
__device__ __forceinline__ void expand_big(SH& sh, int tiles_x, F f) {
    const unsigned int tid = threadIdx.x, nbig = sh.nbig;
    if (tid == 0) {
        unsigned int run = 0;
        for (unsigned int k = 0; k < nbig; ++k) { sh.big[k][3] = run; run += sh.big[k][2]; }
        sh.bigtotal = run;
    }
    __syncthreads();
    const unsigned int total = sh.bigtotal;
    for (unsigned int i = tid; i < total; i += 256u) {
        unsigned int lo = 0, hi = nbig;                    // last rectangle whose first flat index is <= i
        while (hi - lo > 1u) { const unsigned int mid = (lo + hi) >> 1; if (sh.big[mid][3] <= i) lo = mid; else hi = mid; }
        const unsigned int e = i - sh.big[lo][3], w = sh.big[lo][1], xy = sh.big[lo][0];
        const unsigned int ty = (xy >> 16) + e / w, tx = (xy & 0xffffu) + e % w;
        f((unsigned int)(ty * (unsigned int)tiles_x + tx), sh.bigkey[lo]);
    }
}

// G Gaussians per thread per call: the fixed costs (bbox reduce, table zero/flush, ~6 barriers)
// are paid once per 256*G Gaussians.
// bin_preinit: K1<BUCKET> zeroes the table and resets box / nbig / counters before its long vertex
// stage (BucketBinner expects that and a barrier).
__device__ __forceinline__ void bin_preinit(BucketShared& sh) {
    for (int e = (int)threadIdx.x; e < HASH_CAP; e += 256) sh.table[e] = 0;
    if (threadIdx.x == 0) { sh.box[0] = 0x7fffffff; sh.box[1] = 0x7fffffff; sh.box[2] = -1; sh.box[3] = -1; sh.nbig = 0; sh.nvis = 0; sh.nsing = 0; }
}
template <bool EMIT, int G>
__device__ __forceinline__ void bin_block(BinShared& sh, const bool (&vis)[G], const int (&tx0)[G], const int (&tx1)[G],
                                          const int (&ty0)[G], const int (&ty1)[G], int tiles_x,
                                          unsigned int* __restrict__ gcount, unsigned long long* __restrict__ keys,
                                          const unsigned long long (&key)[G], unsigned int bcap = 0) {
    const unsigned int tid = threadIdx.x;
    // where entry `slot` of `tile` lives: absolute (gcount = running cursors, two-pass path) or inside
    // the tile's fixed-stride bucket (gcount = counts from zero, one-pass path; entries past the
    // bucket's capacity are dropped and the frame is flagged by the scan)
    auto put = [&](unsigned int tile, unsigned int slot, unsigned long long k) {
        if (bcap == 0) keys[slot] = k;
        else if (slot < bcap) keys[(size_t)tile * bcap + slot] = k;
    };
    bool small[G], big[G];
    // f(tx, ty, key) for every tile of the thread's aggregated rectangle.  A rectangle of up to LANE_T
    // tiles is walked by its own lane; larger ones (<= AGG_MAX_TILES = one wave) are spread over the
    // lanes of the wave, one tile each -- otherwise the lane that owns a 60-tile splat keeps its
    // wave in the loop sixty rounds (measured: the scatter phase of such a block 42 k cycles vs 3 k).
    constexpr int LANE_T = 9;
    auto each_tile = [&](int g, auto f) {
        const unsigned int lane = tid & 63u;
        const int w = tx1[g] - tx0[g] + 1, nt = small[g] ? w * (ty1[g] - ty0[g] + 1) : 0;
        if (nt > 0 && nt <= LANE_T)
            for (int ty = ty0[g]; ty <= ty1[g]; ++ty)
                for (int tx = tx0[g]; tx <= tx1[g]; ++tx) f(tx, ty, key[g]);
        unsigned long long m = __builtin_amdgcn_ballot_w64(nt > LANE_T);
        while (m) {
            const int src = __builtin_ctzll(m);
            m &= m - 1ull;
            const int X0 = __builtin_amdgcn_readlane(tx0[g], src), Y0 = __builtin_amdgcn_readlane(ty0[g], src);
            const int W = __builtin_amdgcn_readlane(w, src), N = __builtin_amdgcn_readlane(nt, src);
            const unsigned int klo = (unsigned int)__builtin_amdgcn_readlane((int)(unsigned int)key[g], src);
            const unsigned int khi = (unsigned int)__builtin_amdgcn_readlane((int)(unsigned int)(key[g] >> 32), src);
            if ((int)lane < N) {
                const int dy = (int)lane / W, dx = (int)lane - dy * W;
                f(X0 + dx, Y0 + dy, ((unsigned long long)khi << 32) | klo);
            }
        }
    };
    if (tid == 0) { sh.box[0] = 0x7fffffff; sh.box[1] = 0x7fffffff; sh.box[2] = -1; sh.box[3] = -1; sh.nbig = 0; }
    __syncthreads();
    {   // block bounding box of the aggregated rectangles: wave reduce, then one LDS atomic per wave
        int a = 0x7fffffff, b = 0x7fffffff, c = -1, d = -1;
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const int ntiles = vis[g] ? (tx1[g] - tx0[g] + 1) * (ty1[g] - ty0[g] + 1) : 0;
            small[g] = vis[g] && ntiles <= AGG_MAX_TILES;
            big[g] = vis[g] && !small[g];
            if (small[g]) { a = min(a, tx0[g]); b = min(b, ty0[g]); c = max(c, tx1[g]); d = max(d, ty1[g]); }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            a = min(a, __shfl_xor(a, o)); b = min(b, __shfl_xor(b, o));
            c = max(c, __shfl_xor(c, o)); d = max(d, __shfl_xor(d, o));
        }
        if ((tid & 63u) == 0 && c >= 0) {
            atomicMin(&sh.box[0], a); atomicMin(&sh.box[1], b); atomicMax(&sh.box[2], c); atomicMax(&sh.box[3], d);
        }
    }
    __syncthreads();
    const int bx0 = sh.box[0], by0 = sh.box[1], bw = sh.box[2] - bx0 + 1, bh = sh.box[3] - by0 + 1;
    const int area = (sh.box[2] >= 0) ? bw * bh : 0;
    const bool agg = area > 0 && area <= AGG_CAP;
    if (agg) {
        for (int e = (int)tid; e < area; e += 256) sh.table[e] = 0;
        __syncthreads();
#pragma unroll
        for (int g = 0; g < G; ++g)
            each_tile(g, [&](int tx, int ty, unsigned long long) { atomicAdd(&sh.table[(ty - by0) * bw + (tx - bx0)], 1u); });
        __syncthreads();
        for (int e = (int)tid; e < area; e += 256) {
            unsigned int c = sh.table[e];
            if (c) {
                unsigned int tile = (unsigned int)((by0 + e / bw) * tiles_x + bx0 + e % bw);
                unsigned int base = atomicAdd(&gcount[tile], c);     // (issuing several before waiting: measured no gain)
                if (EMIT) sh.table[e] = base;
            }
        }
        if (EMIT) {
            __syncthreads();
#pragma unroll
            for (int g = 0; g < G; ++g)
                each_tile(g, [&](int tx, int ty, unsigned long long k) {
                    const unsigned int slot = atomicAdd(&sh.table[(ty - by0) * bw + (tx - bx0)], 1u);
                    put((unsigned int)(ty * tiles_x + tx), slot, k);
                });
        }
    } else {
#pragma unroll
        for (int g = 0; g < G; ++g)     // bounding box larger than the table (incoherent block): direct
            each_tile(g, [&](int tx, int ty, unsigned long long k) {
                const unsigned int tile = (unsigned int)(ty * tiles_x + tx);
                unsigned int slot = atomicAdd(&gcount[tile], 1u);
                if (EMIT) put(tile, slot, k);
            });
    }
    // close-ups: every thread takes tiles of each big rectangle (in rounds of BIG_CAP)
#pragma unroll
    for (int g = 0; g < G; ++g) {
        if (__syncthreads_or(big[g] ? 1 : 0) == 0) continue;
        if (tid == 0) sh.nbig = 0;
        __syncthreads();
        if (big[g]) add_big(sh, tx0[g], tx1[g], ty0[g], ty1[g], key[g]);
        __syncthreads();
        expand_big(sh, tiles_x, [&](unsigned int tile, unsigned long long kk) {
            unsigned int slot = atomicAdd(&gcount[tile], 1u);
            if (EMIT) put(tile, slot, kk);
        });
        __syncthreads();
    }
}

// One-pass binning for K1<BUCKET>: the same aggregation as bin_block<EMIT>, reorganised to need
// three barriers instead of six (a K1 block is latency bound: every barrier also waits for the
// slowest of its four waves).  The LDS table is addressed by a hash of the tile coordinates,
//     slot(tx, ty) = (ty mod D) * D + ((tx + 17 ty) mod D),      D = HASH_DIM
// which is collision-free whenever the block's tile bounding box is at most D x D -- so counting can
// start before the bounding box is known; the box (LDS atomics, no barrier of its own) is only
// needed afterwards, to validate the hash and to turn slots back into tiles.  A block that fails the
// validation (incoherent: its Gaussians are spread over more than D tiles in x or y) places its
// pairs with one global atomic each.  Block statistics ride on the same barriers.
//
// Two phases, so that the caller can put independent work between them:
//   reserve()  count the block's pairs per tile in LDS, then ISSUE one returning global atomic per touched
//              tile (the reservation of the block's run in that tile's bucket).  The results stay in
//              registers: nothing waits for them yet.
//   place()    wait for the reservations, hand out the slots with LDS atomics, store the keys.
// K1 evaluates the SH colour and writes the record in between: the round trip to the memory-side atomic
// unit (a third of a block's life when it was waited for on the spot) hides under the SH planes' loads.
// The caller has run bin_preinit() and a barrier.
// Workspace of the flattened tile expansion (per wave): see BucketBinner::flat_count.
constexpr int FLAT_CAP = 512;              // (Gaussian, tile) pairs of one wave that go through it (a wave of C3 has ~350)
// (The run-start marks -- a byte per pair: value = source lane + 1 -- live in the 2 KB of BucketShared's union behind the
// table, which only the close-up arrays use, later: 10.4 instead of 12.5 KB per workgroup, and a K1 workgroup fits in the
// 11.5 KB of LDS that seven compositor workgroups leave of a CU.)
struct FlatShared {
    unsigned short exp[4][FLAT_CAP];       // per pair: table slot | source lane << 10, written by the count pass for the hand-out pass
};
static_assert(BucketShared::cap >= HASH_CAP + 4 * (FLAT_CAP / 4), "the marks fit behind the table");

struct BucketBinner {
    static constexpr int LANE_T = 9;       // see bin_block
    static constexpr int NRES = HASH_CAP / 256;
    BucketShared& sh;
    FlatShared& fs;
    const int tiles_x;
    unsigned int* __restrict__ const gcount;      // per tile: the CURSOR of its region (initialised to the region's start)
    unsigned long long* __restrict__ const keys;
    const unsigned int bcap;                      // entries of the key buffer: nothing is stored at or beyond it
    // LARGE splats -- wider or taller than the table's window, or of more than large_tiles tiles -- are not expanded here: one
    // entry (key, tile rectangle) goes to the frame's large list, and bin_large_kernel, behind this launch, bins the list TILE
    // by tile (see there).  large_list == nullptr: the block expands them itself, one returning atomic per pair (expand_big).
    uint4* __restrict__ const large_list;
    unsigned int* __restrict__ const large_count;
    const unsigned int large_cap;                 // entries of the list (the scene's Gaussians: it cannot overflow)
    const int large_tiles;                        // 0: only the window decides
    // (A run that outgrows its tile's region spills into the next tile's.  Nobody looks: the scan sees the cursor beyond
    // the region's end and the frame is skipped as a whole -- sort and compositor never read its lists.)
    // this thread's rectangle
    int tx0, tx1, ty0, ty1, w, ntiles;
    bool small, big, hashed, any;
    bool serial;                           // a small rectangle that did not fit the wave's flattened expansion
    unsigned int flat_total;               // pairs of this wave in the flattened expansion (wave-uniform)
    int bx0, by0;
    unsigned int res[NRES];                // reservations in flight: position of the block's run in the tile's region, per table slot

    __device__ __forceinline__ BucketBinner(BucketShared& s, FlatShared& f, int tiles_x_, unsigned int* g, unsigned long long* k, unsigned int cap,
                                            uint4* ll, unsigned int* lc, unsigned int lcap, int lt)
        : sh(s), fs(f), tiles_x(tiles_x_), gcount(g), keys(k), bcap(cap), large_list(ll), large_count(lc), large_cap(lcap), large_tiles(lt) {}
    // this wave's large splats, appended to the frame's list: one returning atomic per wave that has any (wave-uniform control flow)
    __device__ __forceinline__ void append_large(unsigned long long key) const {
        const unsigned long long m = __builtin_amdgcn_ballot_w64(big);
        if (m == 0ull) return;
        const unsigned int lane = threadIdx.x & 63u;
        unsigned int base = 0u;
        if (lane == (unsigned int)__builtin_ctzll(m)) base = __hip_atomic_fetch_add(large_count, (unsigned int)__builtin_popcountll(m), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        base = (unsigned int)__builtin_amdgcn_readlane((int)base, __builtin_ctzll(m));
        if (big) {
            const unsigned int at = base + __builtin_amdgcn_mbcnt_hi((unsigned int)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)m, 0u));
            if (at < large_cap) {          // two planes: rectangles, then keys (bin_large_kernel's filter reads the first only)
                uint2* const rects = reinterpret_cast<uint2*>(large_list);
                rects[at] = make_uint2((unsigned int)tx0 | ((unsigned int)ty0 << 16), (unsigned int)tx1 | ((unsigned int)ty1 << 16));
                rects[(size_t)large_cap + at] = make_uint2((unsigned int)key, (unsigned int)(key >> 32));
            }
        }
    }
    __device__ __forceinline__ static unsigned int hslot(int tx, int ty) { return (unsigned int)(((ty & (HASH_DIM - 1)) << HASH_BITS) | ((tx + 17 * ty) & (HASH_DIM - 1))); }
    __device__ __forceinline__ void put_at(unsigned int pos, unsigned long long k) const {
        if (pos < bcap) keys[pos] = k;
    }
    // one pair straight on the tile's cursor (blocks the table cannot hold, close-ups)
    __device__ __forceinline__ void put_direct(unsigned int tile, unsigned long long k) const { put_at(atomicAdd(&gcount[tile], 1u), k); }
    // f(tx, ty, key) for every tile of the thread's aggregated rectangle (see bin_block's each_tile).  Rectangles of
    // more than LANE_T tiles are spread over the lanes of their wave, 64 tiles per round.
    template <typename F>
    __device__ __forceinline__ void each_tile(bool mine, unsigned long long key, F f) const {
        const unsigned int lane = threadIdx.x & 63u;
        if (mine && ntiles <= LANE_T)
            for (int ty = ty0; ty <= ty1; ++ty)
                for (int tx = tx0; tx <= tx1; ++tx) f(tx, ty, key);
        unsigned long long m = __builtin_amdgcn_ballot_w64(mine && ntiles > LANE_T);
        while (m) {
            const int src = __builtin_ctzll(m);
            m &= m - 1ull;
            const int X0 = __builtin_amdgcn_readlane(tx0, src), Y0 = __builtin_amdgcn_readlane(ty0, src);
            const int W = __builtin_amdgcn_readlane(w, src), N = __builtin_amdgcn_readlane(ntiles, src);
            const unsigned int klo = (unsigned int)__builtin_amdgcn_readlane((int)(unsigned int)key, src);
            const unsigned int khi = (unsigned int)__builtin_amdgcn_readlane((int)(unsigned int)(key >> 32), src);
            // (W <= HASH_DIM: t / W by a float reciprocal, exact for t < 2^20 after the one-step correction)
            const float rW = __builtin_amdgcn_rcpf((float)W);
            for (int t0 = 0; t0 < N; t0 += 64) {
                const int t = t0 + (int)lane;
                if (t < N) {
                    int dy = (int)((float)t * rW);
                    int dx = t - dy * W;
                    if (dx < 0) { --dy; dx += W; } else if (dx >= W) { ++dy; dx -= W; }
                    f(X0 + dx, Y0 + dy, ((unsigned long long)khi << 32) | klo);
                }
            }
        }
    }
    // The tiles of the wave's small rectangles as ONE index space, 64 per step with every lane busy: pair f belongs to
    // the source lane whose run [P, P + ntiles) contains it.  (Rectangle by rectangle -- a lane walking its own up to
    // nine tiles while its neighbours with three wait, then one step per larger rectangle with a fifth of the lanes
    // active -- a wave took ~25 steps per pass where its ~350 pairs fill six; the two passes over the table were 20 k of
    // a block's 53 k cycles.)  The source of f: every source lane marks the first index of its run (a byte in LDS);
    // within a step the lanes find the latest mark at or before their index with a ballot, a carry links the steps.
    // flat_count adds the pairs to the table and leaves (slot, source lane) per pair for flat_handout, which hands
    // out the positions and stores the keys.  Runs that end beyond FLAT_CAP take the rectangle-by-rectangle path.
    __device__ __forceinline__ void flat_count() {
        const unsigned int lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
        const unsigned int n = small ? (unsigned int)ntiles : 0u;
        const unsigned int inc = wave_inclusive_sum(n);
        const bool flat = n != 0u && inc <= (unsigned int)FLAT_CAP;
        serial = n != 0u && !flat;
        const unsigned long long fm = __builtin_amdgcn_ballot_w64(flat);
        flat_total = fm ? (unsigned int)__builtin_amdgcn_readlane((int)inc, 63 - __builtin_clzll(fm)) : 0u;
        if (flat_total == 0u) return;
        const unsigned int P = inc - n;
        unsigned int* const markw = sh.table + HASH_CAP + wave * (FLAT_CAP / 4);
        unsigned char* const mark = reinterpret_cast<unsigned char*>(markw);
        markw[lane] = 0u; markw[lane + 64u] = 0u;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (flat) mark[P] = (unsigned char)(lane + 1u);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const int pack = tx0 | (ty0 << 12) | ((w - 1) << 24);     // tile coordinates < 4096 (65535 / 16), w <= HASH_DIM
        int carry = 0;
        for (unsigned int base = 0; base < flat_total; base += 64u) {
            const unsigned int f = base + lane;
            const int m = f < flat_total ? (int)mark[f] : 0;
            const unsigned long long q = __builtin_amdgcn_ballot_w64(m != 0) & ((2ull << lane) - 1ull);
            const int at = q ? 63 - __builtin_clzll(q) : 0;
            const int sm = __builtin_amdgcn_ds_bpermute(at << 2, m);
            const int src = q ? sm - 1 : carry;
            carry = __builtin_amdgcn_readlane(src, 63);
            const int Ps = __builtin_amdgcn_ds_bpermute(src << 2, (int)P), pk = __builtin_amdgcn_ds_bpermute(src << 2, pack);
            const int t = (int)f - Ps, W = (pk >> 24) + 1;
            int dy = (int)((float)t * __builtin_amdgcn_rcpf((float)W));      // (t < 1024, W <= 32: exact after the one-step correction)
            int dx = t - dy * W;
            if (dx < 0) { --dy; dx += W; } else if (dx >= W) { ++dy; dx -= W; }
            const unsigned int slot = hslot((pk & 0xfff) + dx, ((pk >> 12) & 0xfff) + dy);
            if (f < flat_total) {
                atomicAdd(&sh.table[slot], 1u);
                fs.exp[wave][f] = (unsigned short)(slot | ((unsigned int)src << 10));
            }
        }
    }
    // (hashed blocks only: the table slot names the tile)
    __device__ __forceinline__ void flat_handout(unsigned long long key) const {
        const unsigned int lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
        const unsigned int i0 = blockIdx.x * 256u + (threadIdx.x & ~63u);     // scene slot of the wave's lane 0 (a key's low half is its thread's slot)
        for (unsigned int base = 0; base < flat_total; base += 64u) {
            const unsigned int f = base + lane;
            const unsigned int e = f < flat_total ? (unsigned int)fs.exp[wave][f] : 0u;
            const unsigned int slot = e & (unsigned int)(HASH_CAP - 1), src = e >> 10;
            const unsigned int khi = (unsigned int)__builtin_amdgcn_ds_bpermute((int)(src << 2), (int)(unsigned int)(key >> 32));
#if SPLAT_EXP_K1DUMMY
            // timing experiment (VERDICT r5 item 1's gate): the arithmetic per-pair staging verdicts would cost here -- eight
            // more cross-lane reads of the source Gaussian's record and SPLAT_EXP_K1DUMMY VALU instructions in four chains
            {
                float c0 = __uint_as_float(khi), c1 = c0 + 1.0f, c2 = c0 + 2.0f, c3 = c0 + 3.0f;
#pragma unroll
                for (int q = 0; q < 8; ++q) c0 += __uint_as_float((unsigned int)__builtin_amdgcn_ds_bpermute((int)(src << 2), (int)__float_as_uint(c1 + (float)q)));
#pragma unroll
                for (int q = 0; q < SPLAT_EXP_K1DUMMY / 4; ++q) { c0 = fmaf(c0, 1.0001f, c1); c1 = fmaf(c1, 0.9999f, c2); c2 = fmaf(c2, 1.0002f, c3); c3 = fmaf(c3, 0.9998f, c0); }
                asm volatile("" :: "v"(c0), "v"(c1), "v"(c2), "v"(c3));
            }
#endif
            if (f < flat_total) {
                const unsigned int pos = atomicAdd(&sh.table[slot], 1u);      // (the table holds positions in the key buffer)
                put_at(pos, ((unsigned long long)khi << 32) | (unsigned long long)(i0 + src));
            }
        }
    }
    __device__ __forceinline__ void count(bool vis, bool singular, int tx0_, int tx1_, int ty0_, int ty1_) {
        const unsigned int tid = threadIdx.x, lane = tid & 63u;
        tx0 = tx0_; tx1 = tx1_; ty0 = ty0_; ty1 = ty1_;
        w = tx1 - tx0 + 1; ntiles = vis ? w * (ty1 - ty0 + 1) : 0;
#if SPLAT_EXP_K1DROP == 1       // timing experiments (invalid frames): what the close-up path costs -- its rectangles dropped
        if (vis && !(w <= HASH_DIM && (ty1 - ty0) < HASH_DIM)) { vis = false; ntiles = 0; }
#elif SPLAT_EXP_K1DROP >= 2     // ... and what every rectangle of more than SPLAT_EXP_K1DROP tiles costs
        if (ntiles > SPLAT_EXP_K1DROP) { vis = false; ntiles = 0; }
#endif
        // Through the hashed table: every rectangle that fits its HASH_DIM x HASH_DIM window (512 x 512 pixels).  Only a
        // splat wider or taller than that is a close-up for the slow path below.  (The limit used to be 64 tiles, one
        // lane-spread round; at the C3 bench pose the 0.45 % of the Gaussians above it put 8 % of the pairs -- and three
        // more barriers, a serial prefix and an exposed atomic round trip -- into most blocks: K1 0.169 -> 0.13 ms.)
        const bool fits = w <= HASH_DIM && (ty1 - ty0) < HASH_DIM, few = large_tiles <= 0 || ntiles <= large_tiles;
        small = vis && fits && (large_list == nullptr || few); big = vis && !small;
        if (large_count != nullptr) {
            // The host decides from counts whether frames keep a list at all (include/splat_policy.h): word 1 counts the splats
            // outside the window -- the ones a block without a list pays one atomic per pair for -- and, when there is no list this
            // frame, word 0 the splats that would have been listed (append_large counts those itself).
            const unsigned long long wm = __builtin_amdgcn_ballot_w64(vis && !fits);
            if (wm != 0ull && (threadIdx.x & 63u) == 0u) (void)__hip_atomic_fetch_add(large_count + 1, (unsigned int)__builtin_popcountll(wm), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (large_list == nullptr) {
                const unsigned long long lm = __builtin_amdgcn_ballot_w64(vis && !(fits && few));
                if (lm != 0ull && (threadIdx.x & 63u) == 0u) (void)__hip_atomic_fetch_add(large_count, (unsigned int)__builtin_popcountll(lm), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        {   // block statistics and bounding box of the aggregated rectangles: wave reduce, LDS atomics by lane 0
            const unsigned int nv = (unsigned int)__builtin_popcountll(__builtin_amdgcn_ballot_w64(vis));
            const unsigned int ns = (unsigned int)__builtin_popcountll(__builtin_amdgcn_ballot_w64(singular));
            auto mn = [](int x, int y) { return min(x, y); };
            auto mx = [](int x, int y) { return max(x, y); };
            const int a = wave_last(wave_reduce_to_last(small ? tx0 : 0x7fffffff, 0x7fffffff, mn));
            const int b = wave_last(wave_reduce_to_last(small ? ty0 : 0x7fffffff, 0x7fffffff, mn));
            const int c = wave_last(wave_reduce_to_last(small ? tx1 : -1, -1, mx));
            const int d = wave_last(wave_reduce_to_last(small ? ty1 : -1, -1, mx));
            if (lane == 0) {
                if (nv) atomicAdd(&sh.nvis, nv);
                if (ns) atomicAdd(&sh.nsing, ns);
                if (c >= 0) { atomicMin(&sh.box[0], a); atomicMin(&sh.box[1], b); atomicMax(&sh.box[2], c); atomicMax(&sh.box[3], d); }
            }
        }
        flat_count();
        each_tile(serial, 0ull, [&](int tx, int ty, unsigned long long) { atomicAdd(&sh.table[hslot(tx, ty)], 1u); });
    }
    __device__ __forceinline__ void issue(unsigned int* __restrict__ blockinfo) {
        const unsigned int tid = threadIdx.x;
        __syncthreads();
        bx0 = sh.box[0]; by0 = sh.box[1];
        any = sh.box[2] >= 0;
        hashed = any && (sh.box[2] - bx0) < HASH_DIM && (sh.box[3] - by0) < HASH_DIM;
        // block statistics as a plain store (summed on the host when statistics are asked for): 5860
        // atomics per frame on one counter are not free on the memory-side atomic unit the reservations use
        if (tid == 0) blockinfo[blockIdx.x] = sh.nvis | (sh.nsing << 9);
#pragma unroll
        for (int q = 0; q < NRES; ++q) res[q] = 0xffffffffu;
        if (hashed) {
            // reserve every touched tile's run in its bucket: one returning global atomic per (block, tile).
            // All counts and addresses first, THEN the atomics back to back: interleaved, the register allocator put a
            // result register next to the 64-bit operand of the following slot's address arithmetic, and the hardware
            // waited out the first atomic's round trip (s_waitcnt vmcnt(0)) before it could issue the second -- two
            // serialised memory round trips per block in front of the SH loads.
            unsigned int cnt[NRES];
            unsigned long long addr[NRES];
#pragma unroll
            for (int q = 0; q < NRES; ++q) {
                const int e = (int)tid + 256 * q;
                cnt[q] = sh.table[e];
                const int ty = by0 + (((e >> HASH_BITS) - by0) & (HASH_DIM - 1));
                const int tx = bx0 + (((e & (HASH_DIM - 1)) - 17 * ty - bx0) & (HASH_DIM - 1));
                addr[q] = (unsigned long long)(gcount + (unsigned int)(ty * tiles_x + tx));     // (only dereferenced where cnt != 0)
            }
            static_assert(NRES == 4, "the scheduling fence below names four addresses");
            asm volatile("" : "+v"(addr[0]), "+v"(addr[1]), "+v"(addr[2]), "+v"(addr[3]), "+v"(cnt[0]), "+v"(cnt[1]), "+v"(cnt[2]), "+v"(cnt[3]));
#pragma unroll
            for (int q = 0; q < NRES; ++q)
                if (cnt[q])     // (a global-address-space pointer: through a generic one this is a FLAT atomic, which every later LDS wait would wait for)
                    res[q] = __hip_atomic_fetch_add((__attribute__((address_space(1))) unsigned int*)addr[q], cnt[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);

        }
    }
    // COUNT flavour of K1 (no keys, no records: the pairs per tile and nothing else): the block's counts go out as plain
    // (non-returning) atomics, the pairs the table does not hold one atomic each.  Ends the block's work.
    __device__ __forceinline__ void count_only() {
        const unsigned int tid = threadIdx.x;
        __syncthreads();
        bx0 = sh.box[0]; by0 = sh.box[1];
        any = sh.box[2] >= 0;
        hashed = any && (sh.box[2] - bx0) < HASH_DIM && (sh.box[3] - by0) < HASH_DIM;
        if (hashed) {
#pragma unroll
            for (int q = 0; q < NRES; ++q) {
                const int e = (int)tid + 256 * q;
                const unsigned int cnt = sh.table[e];
                const int ty = by0 + (((e >> HASH_BITS) - by0) & (HASH_DIM - 1));
                const int tx = bx0 + (((e & (HASH_DIM - 1)) - 17 * ty - bx0) & (HASH_DIM - 1));
                if (cnt) (void)__hip_atomic_fetch_add(gcount + (unsigned int)(ty * tiles_x + tx), cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        } else if (any) {
            each_tile(small, 0ull, [&](int tx, int ty, unsigned long long) { (void)__hip_atomic_fetch_add(gcount + (unsigned int)(ty * tiles_x + tx), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); });
        }
        if (large_list != nullptr) { append_large(0ull); return; }      // (counted tile by tile by bin_large_kernel<true>)
        if (__syncthreads_or(big ? 1 : 0) == 0) return;
        if (big) add_big(sh, tx0, tx1, ty0, ty1, 0ull);
        __syncthreads();
        expand_big(sh, tiles_x, [&](unsigned int tile, unsigned long long) { (void)__hip_atomic_fetch_add(gcount + tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); });
    }
    // place() in two steps, so that the caller can put its record store between them: publish() waits for the
    // reservations (issued before the SH planes' loads, so they have arrived when those have) and hands them to the
    // block through the table; a store issued BEFORE that wait would have to complete first (gfx9 counts loads,
    // atomics and stores in one counter).
    __device__ __forceinline__ void publish() {
        if (hashed) {
#pragma unroll
            for (int q = 0; q < NRES; ++q) sh.table[(int)threadIdx.x + 256 * q] = res[q];     // (slots nobody touched are never read)
        }
    }
    template <typename AFTER>
    __device__ __forceinline__ void place(unsigned long long key, AFTER after_barrier) {
        if (hashed) {
            __syncthreads();
            after_barrier();
            flat_handout(key);
            each_tile(serial, key, [&](int tx, int ty, unsigned long long k) { put_at(atomicAdd(&sh.table[hslot(tx, ty)], 1u), k); });
        } else {
            after_barrier();
            if (any) each_tile(small, key, [&](int tx, int ty, unsigned long long k) { put_direct((unsigned int)(ty * tiles_x + tx), k); });
        }
        // large splats: to the frame's list (bin_large_kernel bins them tile by tile) ...
        if (large_list != nullptr) { append_large(key); return; }
        // ... or, without a list, close-ups (wider or taller than the table's window): the whole block takes the tiles of each, one per thread
        if (__syncthreads_or(big ? 1 : 0) == 0) return;
        if (big) add_big(sh, tx0, tx1, ty0, ty1, key);
        __syncthreads();
        expand_big(sh, tiles_x, [&](unsigned int tile, unsigned long long kk) { put_direct(tile, kk); });
    }
};

struct NoBinner { template <typename... A> __device__ __forceinline__ NoBinner(A&&...) {} };   // two-pass binning: K1 only counts

// order-preserving u32 of an f32 (ascending z == far first in a right-handed view)
__device__ __forceinline__ unsigned int depth_key(float z) {
    unsigned int u = __float_as_uint(z);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// Can ANY Gaussian of a block land on this context's slab?  Conservative test on the block's
// upload-time bounds: every Gaussian centre lies in the AABB, so its clip coordinates lie in the
// convex hull of the eight projected corners (while w keeps its sign), and its 3-sigma half
// extents obey h <= 3 sqrt((focal/z)^2 |v|^2 ||cov3d||_F + lowpass), v the view-matrix row.
// Anything that cannot be bounded (w or view z changing sign inside the box, non-finite bounds)
// answers "maybe".  Margins: 1 px + 0.1 % on screen, 1e-4 on NDC z.  Lanes 0-7 of each wave take one
// corner each; the result is wave-uniform and identical in the block's four waves.
__device__ __forceinline__ bool block_may_reach_slab(const BlockBounds& bb, const FrameConst& fc) {
    const unsigned int lane = threadIdx.x & 63u;
    const float px = (lane & 1u) ? bb.hi[0] : bb.lo[0], py = (lane & 2u) ? bb.hi[1] : bb.lo[1], pz = (lane & 4u) ? bb.hi[2] : bb.lo[2];
    float pc[4], q[4];
    mat4_vec(fc.view, px, py, pz, 1.0f, pc);
    mat4_vec(fc.proj, pc[0], pc[1], pc[2], pc[3], q);
    const float sx = (q[0] / q[3] * 0.5f + 0.5f) * fc.w;
    const float sy = fc.y_up ? (q[1] / q[3] * -0.5f + 0.5f) * fc.h : (q[1] / q[3] * 0.5f + 0.5f) * fc.h;
    const float sz = q[2] / q[3];
    float lo[5] = {sx, sy, sz, q[3], pc[2]}, hi[5] = {sx, sy, sz, q[3], pc[2]};
    bool finite = finitef(sx) && finitef(sy) && finitef(sz) && finitef(q[3]) && finitef(pc[2]);
#pragma unroll
    for (int o = 1; o < 8; o <<= 1) {
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            lo[k] = fminf(lo[k], __shfl_xor(lo[k], o));
            hi[k] = fmaxf(hi[k], __shfl_xor(hi[k], o));
        }
        finite = finite & (__shfl_xor((int)finite, o) != 0);
    }
    if (!finite) return true;
    if (!(lo[3] > 0.0f || hi[3] < 0.0f) || !(lo[4] > 0.0f || hi[4] < 0.0f)) return true;   // w or view z crosses zero
    const float wabs = fminf(fabsf(lo[3]), fabsf(hi[3])), wmax = fmaxf(fabsf(lo[3]), fabsf(hi[3]));
    if (!(wabs > 1e-3f * wmax)) return true;                                               // projection too steep to trust
    const float zabs = fminf(fabsf(lo[4]), fabsf(hi[4]));
    float nx = fc.view[0] * fc.view[0] + fc.view[4] * fc.view[4] + fc.view[8] * fc.view[8];
    float ny = fc.view[1] * fc.view[1] + fc.view[5] * fc.view[5] + fc.view[9] * fc.view[9];
    if (fc.corrected) {
        // the Jacobian row is (f/z)(1, 0, -tx/z) with |tx/z| <= 1.3 htan: |row V|^2 <= |row|^2 ||V||_F^2
        const float nz = fc.view[2] * fc.view[2] + fc.view[6] * fc.view[6] + fc.view[10] * fc.view[10];
        const float ht = fmaxf(fc.htanx, fc.htany);
        nx = ny = (nx + ny + nz) * (1.0f + 1.69f * ht * ht);
    }
    const float fz = fc.focal / zabs, s = fz * fz * bb.fmax;
    const float hx = 3.0f * sqrtf(s * nx + fc.lowpass) * 1.001f + 1.0f, hy = 3.0f * sqrtf(s * ny + fc.lowpass) * 1.001f + 1.0f;
    auto pad = [](float v) { return 1e-3f * fabsf(v); };
    // written so that a NaN anywhere falls through to "maybe"
    if (hi[0] + hx + pad(hi[0]) < 0.0f || lo[0] - hx - pad(lo[0]) > fc.w) return false;
    if (hi[1] + hy + pad(hi[1]) < (float)fc.row_px0 || lo[1] - hy - pad(lo[1]) > (float)fc.row_px1) return false;
    if (fc.zclip && (hi[2] + 1e-4f + pad(hi[2]) * 0.1f < fc.zmin || lo[2] - 1e-4f - pad(lo[2]) * 0.1f > fc.zmax)) return false;
    return true;
}

// K1 -- one thread per Gaussian (slot j of the Morton-ordered scene, original index orig[j]): the
// whole vertex stage, a 48-B record, the exactly covered pixel rectangle, and per-tile counts.
// BUCKET = false (two-pass binning): counts only; depth/rect/vislist feed K2, which emits the keys
//          into exactly sized lists after the scan.
// BUCKET = true  (one-pass binning): every tile owns a REGION of the key buffer, keys[layout[t] .. layout[t + 1]) --
//          sized from the list the tile had two frames earlier (layout_kernel) -- and counts[t] is the region's cursor;
//          the reservation that counts a (Gaussian, tile) pair also places its key, so there is no K2 and no
//          second read of the per-Gaussian data.  fc.bucket_cap = entries of the key buffer.
// CORRECTED = SPLAT_MODE_CORRECTED_PROJECTION, a compile-time flavour: in the reference's projection the clamped tx/ty
// and the Jacobian's shear entries only reach the discarded third column of cov (src/gaussians.rs:133-151), so the
// compiler drops them -- two divisions, the clamps and a third of the products -- when it can see that.
// (Eight blocks per CU instead of seven -- __launch_bounds__(256, 8): 64 VGPRs, five of them spilled -- measured in round 6:
// K1 alone 0.146 -> 0.160 ms, the frame 3175 -> 3090 frames/s.  Not taken.)
// COUNT (with BUCKET): the pairs per tile and nothing else -- geometry planes only, no SH, no record, no key: what a frame
// needs to build regions that fit exactly ITS camera before it bins (enqueue_frame: count-first frames, the bootstrap).
template <bool BUCKET, bool CORRECTED = false, bool COUNT = false>
__global__ __launch_bounds__(256) void preprocess_kernel(uint64_t n, const float4* __restrict__ planes,
                                                         const unsigned int* __restrict__ orig, FrameConst fc,
                                                         Rec* __restrict__ recs, float* __restrict__ depth,
                                                         ushort4* __restrict__ rect, unsigned int* __restrict__ counts,
                                                         unsigned int* __restrict__ vislist,
                                                         unsigned long long* __restrict__ keys,
                                                         const BlockBounds* __restrict__ bounds,
                                                         unsigned int* __restrict__ blockinfo,
                                                         FrameStatus* __restrict__ status,
                                                         uint4* __restrict__ large_list, unsigned int* __restrict__ large_count) {
    __shared__ std::conditional_t<BUCKET, BucketShared, BinShared> sh;
    __shared__ std::conditional_t<BUCKET, FlatShared, unsigned int> fsh;
    __shared__ unsigned int swave[4];
    __shared__ unsigned int sbase, sreach;
    // K1 is a chain of dependent phases with little arithmetic in each; beside the compositor's
    // throughput-bound waves every one of its instructions queued behind theirs.  One priority step above the
    // compositor's default: K1 0.38 -> 0.32 ms inside the pipeline, +4 % frames/s on C3.
    // (the REDO launches behind a frame's scan -- count pass and second binning pass of a frame whose lists outgrew regions
    // sized from earlier frames -- leave at once unless that scan said so)
    if (fc.redo_only && status->overflow != 2u) return;
    __builtin_amdgcn_s_setprio(1);
#if SPLAT_K1X == 30 || SPLAT_K1X == 31
    unsigned long long* const stamps = reinterpret_cast<unsigned long long*>(depth) + ((size_t)blockIdx.x * 4u + (threadIdx.x >> 6)) * 8u;
#define STAMP_(k) do { if ((threadIdx.x & 63u) == 0u) stamps[k] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define STAMP_(k) do { } while (0)
#endif
#if SPLAT_K1X == 31
#define STAMP(k) do { if ((k) == 0) STAMP_(0); else if ((k) == 1) STAMP_(4); else if ((k) == 2) STAMP_(5); else if ((k) == 4) STAMP_(6); else if ((k) == 7) STAMP_(7); } while (0)
#define STAMPF(k) STAMP_(k)
#else
#define STAMP(k) STAMP_(k)
#define STAMPF(k) do { } while (0)
#endif
    STAMP(0);
#if SPLAT_K1X == 30 || SPLAT_K1X == 31
    if ((threadIdx.x & 63u) == 0u)      // HW_REG_HW_ID (id 4), all 32 bits; HW_REG_XCC_ID (id 20)
        reinterpret_cast<unsigned long long*>(rect)[(size_t)blockIdx.x * 4u + (threadIdx.x >> 6)] =
            ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32) | (unsigned int)__builtin_amdgcn_s_getreg((31 << 11) | 4);
#endif
    if constexpr (BUCKET) {
        // Block culling rides on the barrier the table reset needs anyway: wave 0 alone runs the bounds test (a
        // few hundred instructions that used to be issued by all four waves) while the others clear the table.
        if (fc.cull_blocks && threadIdx.x < 64u) {
            const bool reach = block_may_reach_slab(bounds[blockIdx.x], fc);
            // (the flag is a plain store: thousands of culled blocks retire within microseconds, and that
            // many atomics on one counter took longer than the blocks they counted)
            if (threadIdx.x == 0) { sreach = reach ? 1u : 0u; blockinfo[blockIdx.x] = reach ? 0u : 0x80000000u; }   // (binning overwrites a 0 with its counts)
        }
        bin_preinit(sh);
        __syncthreads();
        if (fc.cull_blocks && sreach == 0u) return;          // whole block off this context's slab / target: nothing to read
        STAMPF(1);
    } else if (fc.cull_blocks) {
        const bool reach = block_may_reach_slab(bounds[blockIdx.x], fc);
        if (threadIdx.x == 0) blockinfo[blockIdx.x] = reach ? 0u : 0x80000000u;
        if (!reach) return;
    }
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    bool singular = false, in_slab = false;
    int tx0 = 0, tx1 = -1, ty0 = 0, ty1 = -1;
    float F[64];
    float cx = 0, cy = 0, hx = 0, hy = 0, ca = 0, cb = 0, cc = 0, zview = 0;
    if (i < n) {
        // geometry first: position, opacity, cov3d live in planes 0-3 (64 B); the SH planes are
        // only fetched for Gaussians that reach this context's slab
        // (requesting them before the table reset / culling test / barrier of the prologue, so that its ~8 k cycles run
        // under the loads' flight: measured twice, rounds 2 and 3 -- no gain, 0.147 -> 0.149 ms)
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            float4 v = planes[(uint64_t)p * n + i];
            F[4 * p] = v.x; F[4 * p + 1] = v.y; F[4 * p + 2] = v.z; F[4 * p + 3] = v.w;
        }
        const float px = F[0], py = F[1], pz = F[2];
        asm volatile("" : "+v"(F[12]));
        STAMPF(2);

        // project_cov3d_to_screen                                            src/gaussians.rs:114-161
        float pc[4];
        mat4_vec(fc.view, px, py, pz, 1.0f, pc);
        float limx = 1.3f * fc.htanx, limy = 1.3f * fc.htany;
        float txtz = pc[0] / pc[2], tytz = pc[1] / pc[2];
        float tx = fminf(limx, fmaxf(-limx, txtz)) * pc[2];
        float ty = fminf(limy, fmaxf(-limy, tytz)) * pc[2];
        float tz = pc[2];
        Mat3 J;
        M3(J, 0, 0) = fc.focal / tz; M3(J, 0, 1) = 0.0f;          M3(J, 0, 2) = -(fc.focal * tx) / (tz * tz);
        M3(J, 1, 0) = 0.0f;          M3(J, 1, 1) = fc.focal / tz; M3(J, 1, 2) = -(fc.focal * ty) / (tz * tz);
        M3(J, 2, 0) = 0.0f;          M3(J, 2, 1) = 0.0f;          M3(J, 2, 2) = 0.0f;
        Mat3 Wm;   // viewmatrix.fixed_view::<3,3>(0,0).transpose()
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c) M3(Wm, r, c) = fc.view[r * 4 + c];
        Mat3 T = CORRECTED ? mat3_mul(Wm, mat3_t(J)) : mat3_mul(Wm, J);   // corrected: the shear column reaches the 2x2 block
        Mat3 Sg;
#pragma unroll
        for (int e = 0; e < 9; ++e) Sg.m[e] = F[4 + e];
        Mat3 cov = mat3_mul(mat3_mul(mat3_t(T), mat3_t(Sg)), T);
        float m11 = M3(cov, 0, 0) + fc.lowpass, m21 = M3(cov, 1, 0), m12 = M3(cov, 0, 1), m22 = M3(cov, 1, 1) + fc.lowpass;

        // gaussian_vertex_shader                                             src/pipelines.rs:17-51
        float det = m11 * m22 - m21 * m12;       // nalgebra 2x2 try_inverse
        float q[4];
        mat4_vec(fc.proj, pc[0], pc[1], pc[2], pc[3], q);
        float ndcx = q[0] / q[3], ndcy = q[1] / q[3], ndcz = q[2] / q[3];
        ca = m22 / det; cb = -m12 / det; cc = m11 / det;
        hx = 3.0f * sqrtf(m11); hy = 3.0f * sqrtf(m22);
        // euc: NDC -> target pixels
        cx = (ndcx * 0.5f + 0.5f) * fc.w;
        cy = fc.y_up ? (ndcy * -0.5f + 0.5f) * fc.h : (ndcy * 0.5f + 0.5f) * fc.h;


        singular = (det == 0.0f);
        asm volatile("" : "+v"(ca), "+v"(cb), "+v"(cc), "+v"(cx), "+v"(cy), "+v"(hx), "+v"(hy));
        STAMPF(3);
        bool vis = !singular && finitef(cx) && finitef(cy) && finitef(hx) && finitef(hy) && finitef(ca) && finitef(cb) &&
                   finitef(cc) && finitef(ndcz);
        if (vis && fc.zclip) vis = (fc.zmin <= ndcz) && (ndcz <= fc.zmax);
        // exactly covered pixel ranges on the whole target; the slab then bounds the rows
        int fx0 = 1, fx1 = 0, fy0 = 1, fy1 = 0;
        const float off = fc.sample_half ? 0.5f : 0.0f;
        bool on_target = vis && covered_interval(cx, hx, off, 0, fc.W - 1, &fx0, &fx1) &&
                         covered_interval(cy, hy, off, 0, fc.H - 1, &fy0, &fy1);
        if (on_target) {
            int y0 = max(fy0, fc.row_px0), y1 = min(fy1, fc.row_px1 - 1);
            in_slab = y0 <= y1;
            tx0 = fx0 >> 4; tx1 = fx1 >> 4; ty0 = (y0 >> 4) - fc.tile_row0; ty1 = (y1 >> 4) - fc.tile_row0;
        }
        zview = pc[2];
        if (!BUCKET) {
            depth[i] = pc[2];
            rect[i] = on_target ? make_ushort4((unsigned short)fx0, (unsigned short)fx1, (unsigned short)fy0, (unsigned short)fy1)
                                : make_ushort4(1, 0, 1, 0);
        }
    }
    std::conditional_t<BUCKET, BucketBinner, NoBinner> binner(sh, fsh, fc.tiles_x, counts, keys, fc.bucket_cap, large_list, large_count, (unsigned int)min(n, (uint64_t)0xffffffffu), fc.large_tiles);
    // The SH planes' loads go out BEFORE the reservations: vector memory returns in order, so loads issued behind
    // the returning atomics could not be consumed before those have made their round trip to the L2 atomic unit
    // (queued behind every other block's atomics on the same hot tile counters) -- the SH stage, the record store
    // and the whole rest of the block used to wait that out.  In this order the reservations are in flight under
    // the SH arithmetic instead.
    Rec r;
    if constexpr (COUNT) {
        static_assert(BUCKET || !COUNT, "the count flavour belongs to one-pass binning");
        binner.count(in_slab, singular, tx0, tx1, ty0, ty1);
        binner.count_only();
        return;
    }
    if (in_slab) {
#pragma unroll
        for (int p = 4; p < LIVE_PLANES; ++p) {
            float4 v = planes[(uint64_t)p * n + i];
            F[4 * p] = v.x; F[4 * p + 1] = v.y; F[4 * p + 2] = v.z; F[4 * p + 3] = v.w;
        }
    }
    if constexpr (BUCKET) {
        STAMP(1);
        binner.count(in_slab, singular, tx0, tx1, ty0, ty1);     // the block's pairs per tile, in LDS -- under the loads
        STAMP(2);
        // (all six planes in flight -- left alone, the compiler sinks each plane's load into the sh_dim branch that uses
        // it: three dependent round trips -- and landed, before the barrier of the count and the atomics behind it)
        asm volatile("" : "+v"(F[16]), "+v"(F[20]), "+v"(F[24]), "+v"(F[28]), "+v"(F[32]), "+v"(F[36]));
        STAMP(3);
        binner.issue(blockinfo);            // reservations in flight from here on
        STAMP(4);
    }
    if (in_slab) {
        const float px = F[0], py = F[1], pz = F[2], opacity = F[3];
        const float* sh = F + 13;

        // ray_direction = (position - camera.position).normalize()            src/pipelines.rs:99
        float dxw = px - fc.cam[0], dyw = py - fc.cam[1], dzw = pz - fc.cam[2];
        float nrm = sqrtf((dxw * dxw + dyw * dyw) + dzw * dzw);
        float x = dxw / nrm, y = dyw / nrm, z = dzw / nrm;
        if constexpr (!BUCKET) asm volatile("" : "+v"(F[16]), "+v"(F[20]), "+v"(F[24]), "+v"(F[28]), "+v"(F[32]), "+v"(F[36]));   // (all six SH planes in flight at once)

        // eval_spherical_harmonics                                           src/gaussians.rs:40-99
        float col[3];
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) col[ch] = SH_C0 * sh[ch];
        if (fc.sh_dim > 3) {
            float k1 = SH_C1 * y, k2 = SH_C1 * z, k3 = SH_C1 * x;
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) col[ch] = col[ch] - k1 * sh[3 + ch] + k2 * sh[6 + ch] - k3 * sh[9 + ch];
            if (fc.sh_dim > 12) {
                float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
                float k4 = SH_C2_0 * xy, k5 = SH_C2_1 * yz, k6 = SH_C2_2 * (2.0f * zz - xx - yy);
                float k7 = SH_C2_3 * xz, k8 = SH_C2_4 * (xx - yy);
#pragma unroll
                for (int ch = 0; ch < 3; ++ch)
                    col[ch] = col[ch] + k4 * sh[12 + ch] + k5 * sh[15 + ch] + k6 * sh[18 + ch] + k7 * sh[21 + ch] +
                              k8 * sh[24 + ch];
                if (fc.sh_dim > 27) {
#pragma unroll
                    for (int p = LIVE_PLANES; p < SCENE_PLANES; ++p) {
                        float4 v = planes[(uint64_t)p * n + i];
                        F[4 * p] = v.x; F[4 * p + 1] = v.y; F[4 * p + 2] = v.z; F[4 * p + 3] = v.w;
                    }
                    float k9 = SH_C3_0 * y * (3.0f * xx - yy), k10 = SH_C3_1 * xy * z;
                    float k11 = SH_C3_2 * y * (4.0f * zz - xx - yy);
                    float k12 = SH_C3_3 * z * (2.0f * zz - 3.0f * xx - 3.0f * yy);
                    float k13 = SH_C3_4 * x * (4.0f * zz - xx - yy), k14 = SH_C3_5 * z * (xx - yy);
                    float k15 = SH_C3_6 * x * (xx - 3.0f * yy);
#pragma unroll
                    for (int ch = 0; ch < 3; ++ch)
                        col[ch] = col[ch] + k9 * sh[27 + ch] + k10 * sh[30 + ch] + k11 * sh[33 + ch] + k12 * sh[36 + ch] +
                                  k13 * sh[39 + ch] + k14 * sh[42 + ch] + k15 * sh[45 + ch];
                }
            }
        }
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) col[ch] = col[ch] + 0.5f;   // HALF, no clamp

        // fragments with power < pthr have alpha < 1/255 for certain (margin 1e-3 >> f32 error)
        float pthr = (opacity > 0.0f) ? (logf(1.0f / (255.0f * opacity)) - 1e-3f)
                                      : ((opacity <= 0.0f) ? 3.0e38f : -3.0e38f);
        // A colour that is not finite (SH coefficients of +-inf / NaN) is stored as the finite value that blends
        // to the same byte for every ACCEPTED fragment (alpha >= 1/255: +inf and FLT_MAX both saturate to 255,
        // -inf / NaN / -FLT_MAX all end at 0 through the saturating cast, src/pipelines.rs:159-161), so that a
        // REJECTED fragment's 0 * colour is 0 as in the reference, whose rejected fragment is (0,0,0,0)
        // (src/pipelines.rs:135-143), and not 0 * inf = NaN.  Finite colours are untouched.
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
            const float v = col[ch];
            col[ch] = (v != v) ? -3.402823466e+38f : fminf(fmaxf(v, -3.402823466e+38f), 3.402823466e+38f);
        }
        r.a = make_float4(cx, cy, hx, hy);
        r.b = make_float4(ca, fc.y_up ? cb : -cb, cc, opacity);   // cross term carries the y-axis sign (exact)
        r.c = make_float4(col[0], col[1], col[2], pthr);
    }
    // the reservations have arrived with the SH planes (issued before them): hand them to the block BEFORE the record
    // store is issued -- the wait for a reservation issued after a store would include that store's completion
    STAMP(5);
    if constexpr (BUCKET) binner.publish();
    STAMP(6);
    auto store_record = [&] { if (in_slab) recs[i] = r; };   // slot order: coalesced here, Morton-local for the compositor's gathers
    if constexpr (!BUCKET) store_record();
    if constexpr (BUCKET) {
        const unsigned long long key = in_slab ? (((unsigned long long)depth_key(zview) << 32) | (unsigned long long)(unsigned int)i) : 0ull;
        binner.place(key, store_record);        // (the record store goes out behind the barrier of the hand-out)
        STAMP(7);
    } else {
        // compact the slots that reach the slab into vislist (K2 runs over those only)
        {
            const unsigned int lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
            const unsigned long long m = __builtin_amdgcn_ballot_w64(in_slab);
            if (lane == 0) swave[wave] = (unsigned int)__builtin_popcountll(m);
            const int nsing = __syncthreads_count(singular);
            const unsigned int nvis = swave[0] + swave[1] + swave[2] + swave[3];
            if (threadIdx.x == 0) {
                sbase = nvis ? (unsigned int)atomicAdd(&status->n_visible, (unsigned long long)nvis) : 0u;
                if (nsing) atomicAdd(&status->n_singular, (unsigned long long)nsing);
            }
            __syncthreads();
            if (in_slab) {
                unsigned int r = sbase + (unsigned int)__builtin_popcountll(m & ((1ull << lane) - 1ull));
                for (unsigned int w = 0; w < wave; ++w) r += swave[w];
                vislist[r] = (unsigned int)i;
            }
        }
        const bool v1[1] = {in_slab};
        const int a1[1] = {tx0}, b1[1] = {tx1}, c1[1] = {ty0}, d1[1] = {ty1};
        const unsigned long long k1[1] = {0ull};
        bin_block<false, 1>(sh, v1, a1, b1, c1, d1, fc.tiles_x, counts, nullptr, k1);
    }
}

// LARGE SPLATS, binned tile by tile.  K1 expands a Gaussian's (Gaussian, tile) pairs block by block: the pairs of 256 Morton
// neighbours meet in an LDS table and cost one returning global atomic per (block, tile).  A splat that covers hundreds or
// thousands of tiles -- a camera inside the scene: 0.5 % of C3's visible Gaussians carry 18 % of the pairs there, on C5 40 % --
// has no neighbours to share with: one atomic per pair on the same few thousand tile cursors from every block of the launch,
// 3-8 G atomics/s, and K1 took 1.1 ms instead of 0.25 on C3 from inside the cloud, 13.5 instead of 1.75 on C5
// (profiles/r07_k1_closeup.txt).  Such splats go to a list instead (BucketBinner::append_large: 16 B each), and this kernel,
// launched behind K1 on the same stream, turns the list inside out: one workgroup per 4 x 4 TILES streams the whole list
// (coalesced, L2-resident), keeps the entries whose rectangle meets its tiles, and appends their keys to those tiles' regions --
// the cursors are this workgroup's alone by now (K1 has ended), so the only atomics are LDS ones on sixteen counters.
// Replaces the per-splat row loops of euc's rasteriser (src/pipelines.rs:80-84) for the splats that fill the screen.
// COUNT: the count flavour's twin (adds the pairs to the counts, stores nothing).  A frame without large splats: every workgroup
// reads one word and leaves.
constexpr int LARGE_G = 4;                 // tiles per side of a workgroup's group (2 / 6 / 8 measured: profiles/r07_bin_large_v2_ab.txt)
typedef unsigned short LargeU16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned int pk_min_u16(unsigned int a, unsigned int b) {
    return __builtin_bit_cast(unsigned int, __builtin_elementwise_min(__builtin_bit_cast(LargeU16x2, a), __builtin_bit_cast(LargeU16x2, b)));
}
__device__ __forceinline__ unsigned int pk_max_u16(unsigned int a, unsigned int b) {
    return __builtin_bit_cast(unsigned int, __builtin_elementwise_max(__builtin_bit_cast(LargeU16x2, a), __builtin_bit_cast(LargeU16x2, b)));
}
// The list is two planes of large_cap entries: the tile rectangles {x0 | y0 << 16, x1 | y1 << 16} -- all the filter reads, 8 B an
// entry, so that a third of a million entries (C5 from inside) stay in an XCD's L2 -- and behind them the keys {slot, depth key}.
template <bool COUNT>
__global__ __launch_bounds__(256) void bin_large_kernel(const uint4* __restrict__ list, const unsigned int* __restrict__ count_p, unsigned int large_cap,
                                                        unsigned int* __restrict__ cursors, unsigned long long* __restrict__ keys,
                                                        unsigned int bcap, int tiles_x, int tile_rows,
                                                        const FrameStatus* __restrict__ status, unsigned int redo_only) {
    if (redo_only && status->overflow != 2u) return;          // (a redo launch of a frame that needs none)
    const unsigned int n = min((unsigned int)__builtin_amdgcn_readfirstlane((int)*count_p), large_cap);
    if (n == 0u) return;
    const uint2* __restrict__ const rects = reinterpret_cast<const uint2*>(list);
    const uint2* __restrict__ const lkeys = rects + large_cap;
    constexpr int NT = LARGE_G * LARGE_G;
    constexpr int CH = 8;                                      // tiles handled at a time: their ballots live in scalar registers (16: 42 of them spilled)
    static_assert(NT % CH == 0, "tiles of a group are handled CH at a time");
    __shared__ unsigned int cnt[NT];                           // keys appended so far, per tile of the group
    __shared__ unsigned int first[NT];                         // the tile's cursor when this launch began
    __shared__ uint4 queue[4][128];                            // per wave: {rectangle, entry} that meet the group, waiting for a full batch
    const unsigned int tid = threadIdx.x, lane = tid & 63u, wave = (unsigned int)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
    const int groups_x = (tiles_x + LARGE_G - 1) / LARGE_G;
    const int gx = (int)(blockIdx.x % (unsigned int)groups_x), gy = (int)(blockIdx.x / (unsigned int)groups_x);
    const int tx_lo = gx * LARGE_G, ty_lo = gy * LARGE_G;
    const int tx_hi = min(tx_lo + LARGE_G - 1, tiles_x - 1), ty_hi = min(ty_lo + LARGE_G - 1, tile_rows - 1);
    if (tid < (unsigned int)NT) {
        const int tx = tx_lo + (int)(tid % LARGE_G), ty = ty_lo + (int)(tid / LARGE_G);
        first[tid] = (tx <= tx_hi && ty <= ty_hi) ? cursors[ty * tiles_x + tx] : 0u;
        cnt[tid] = 0u;
    }
    __syncthreads();
    uint4* const q = queue[wave];
    // The keys of up to 64 queued entries, eight tiles at a time: eight ballots find the entries that cover each tile, lane u
    // reserves tile u's run behind its cursor -- ONE LDS atomic instruction for the eight (one per tile and batch, one after the
    // other, was a chain of sixteen LDS round trips a batch) -- and the covering lanes store their keys.
    auto flush = [&](unsigned int k) {
        const uint4 e = lane < k ? q[lane] : make_uint4(0xffffffffu, 0u, 0u, 0u);       // (x0 = y0 = 65535: covers nothing)
        uint2 kk = make_uint2(0u, 0u);
        if constexpr (!COUNT) { if (lane < k) kk = lkeys[e.z]; }
#pragma unroll
        for (int c0 = 0; c0 < NT; c0 += CH) {
            unsigned long long m[CH];
            unsigned int mine = 0u, cbits = 0u;
#pragma unroll
            for (int u = 0; u < CH; ++u) {
                const unsigned int tp = (unsigned int)(tx_lo + (c0 + u) % LARGE_G) | ((unsigned int)(ty_lo + (c0 + u) / LARGE_G) << 16);
                // x0 <= tx && y0 <= ty && tx <= x1 && ty <= y1   (x1 < tiles_x, y1 < tile_rows: a tile beyond the grid is never covered)
                const bool c = pk_max_u16(e.x, tp) == tp && pk_min_u16(e.y, tp) == tp;
                m[u] = __builtin_amdgcn_ballot_w64(c);
                if (lane == (unsigned int)u) mine = (unsigned int)__builtin_popcountll(m[u]);
                cbits |= c ? (1u << u) : 0u;
            }
            unsigned int base = 0u;
            if (lane < (unsigned int)CH) base = first[c0 + lane] + (mine ? atomicAdd(&cnt[c0 + lane], mine) : 0u);
            if constexpr (!COUNT) {
#pragma unroll
                for (int u = 0; u < CH; ++u) {
                    if (m[u] == 0ull) continue;
                    const unsigned int bt = (unsigned int)__builtin_amdgcn_readlane((int)base, u);
                    if (cbits & (1u << u)) {
                        const unsigned int pos = bt + __builtin_amdgcn_mbcnt_hi((unsigned int)(m[u] >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)m[u], 0u));
                        if (pos < bcap) keys[pos] = ((unsigned long long)kk.y << 32) | (unsigned long long)kk.x;      // (see BucketBinner::put_at)
                    }
                }
            }
        }
    };
    const unsigned int hi_p = (unsigned int)tx_hi | ((unsigned int)ty_hi << 16), lo_p = (unsigned int)tx_lo | ((unsigned int)ty_lo << 16);
    unsigned int qn = 0u;                                      // entries in this wave's queue (uniform)
    constexpr int AHEAD = 4;                                   // rectangles a lane has in flight
    for (unsigned int i0 = wave * 64u; i0 < n; i0 += 256u * AHEAD) {   // (wave w takes entries [64 w + 256 j, + 64): the four waves never meet)
        uint2 r[AHEAD];
#pragma unroll
        for (int a = 0; a < AHEAD; ++a) {
            const unsigned int i = i0 + 256u * a + lane;
            r[a] = i < n ? rects[i] : make_uint2(0xffffffffu, 0u);
        }
#pragma unroll
        for (int a = 0; a < AHEAD; ++a) {
            const unsigned int i = i0 + 256u * a + lane;
            // x0 <= tx_hi && y0 <= ty_hi && x1 >= tx_lo && y1 >= ty_lo
            const bool pass = pk_min_u16(r[a].x, hi_p) == r[a].x && pk_max_u16(r[a].y, lo_p) == r[a].y;
            const unsigned long long m = __builtin_amdgcn_ballot_w64(pass);
            if (m == 0ull) continue;
            if (pass) q[qn + __builtin_amdgcn_mbcnt_hi((unsigned int)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)m, 0u))] = make_uint4(r[a].x, r[a].y, i, 0u);
            qn += (unsigned int)__builtin_popcountll(m);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (qn >= 64u) {
                flush(64u);
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                const uint4 rest = q[64u + lane];                  // (at most 63 are left: they move to the front)
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                qn -= 64u;
                if (lane < qn) q[lane] = rest;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
        }
    }
    if (qn != 0u) flush(qn);
    __syncthreads();
    if (tid < (unsigned int)NT) {
        const int tx = tx_lo + (int)(tid % LARGE_G), ty = ty_lo + (int)(tid / LARGE_G);
        if (tx <= tx_hi && ty <= ty_hi && cnt[tid] != 0u) cursors[ty * tiles_x + tx] = first[tid] + cnt[tid];
    }
}

// Exclusive scan of the per-tile counts by ONE 1024-thread workgroup, 1024 tiles per step with a
// running carry.  Writes offsets[0..m] and cursor[0..m), zeroes counts for the next frame,
// reports D / overflow / longest list, and emits `order`: tile ids by descending list-length
// class, so the sort and composite grids start their longest tiles first (no long tail).
__global__ __launch_bounds__(1024) void scan_kernel(unsigned int m, unsigned int* __restrict__ counts,
                                                    unsigned int* __restrict__ offsets, unsigned int* __restrict__ cursor,
                                                    unsigned int* __restrict__ order, unsigned int* __restrict__ lens,
                                                    FrameStatus* __restrict__ status,
                                                    unsigned long long capacity, unsigned int bucket_cap,
                                                    unsigned int grid_big, unsigned int grid_mid, unsigned int grid_long,
                                                    FrameStatus* __restrict__ host_status) {
    constexpr int NCLS = 64;
    __shared__ unsigned int wsum[16];
    __shared__ unsigned int hist[NCLS];
    __shared__ unsigned int start[NCLS];
    const unsigned int tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    if (tid < NCLS) hist[tid] = 0;
    unsigned int carry = 0, mx = 0;
    auto cls_of = [](unsigned int c) -> unsigned int {
        if (c == 0) return 0u;
        unsigned int msb = 31u - (unsigned int)__clz((int)c);
        unsigned int half = msb ? ((c >> (msb - 1)) & 1u) : 0u;
        return min(1u + 2u * msb + half, (unsigned int)NCLS - 1u);
    };
    __syncthreads();
    for (unsigned int base = 0; base < m; base += 1024) {
        const unsigned int k = base + tid;
        unsigned int c = 0;
        if (k < m) { c = counts[k]; counts[k] = 0; }
        mx = max(mx, c);
        unsigned int v = c;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            unsigned int t = (unsigned int)__shfl_up((int)v, o);
            if ((int)lane >= o) v += t;
        }
        if (lane == 63) wsum[wave] = v;
        __syncthreads();
        unsigned int woff = 0, total = 0;
#pragma unroll
        for (unsigned int w = 0; w < 16; ++w) { unsigned int x = wsum[w]; total += x; if (w < wave) woff += x; }
        const unsigned int excl = carry + woff + v - c;
        if (k < m) {
            // one-pass binning: list k starts at its bucket and is at most bucket_cap long
            offsets[k] = bucket_cap ? k * bucket_cap : excl;
            cursor[k] = excl;
            const unsigned int len = bucket_cap ? min(c, bucket_cap) : c;
            lens[k] = len;
            atomicAdd(&hist[cls_of(len)], 1u);
        }
        carry += total;
        __syncthreads();
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = max(mx, (unsigned int)__shfl_xor((int)mx, o));
    if (lane == 0) wsum[wave] = mx;
    __syncthreads();
    if (tid == 0) {
        for (int w = 0; w < 16; ++w) mx = max(mx, wsum[w]);
        offsets[m] = carry;
        status->n_pairs = carry;
        status->overflow = bucket_cap ? ((mx > bucket_cap) ? 2u : 0u) : (((unsigned long long)carry > capacity) ? 1u : 0u);
        status->max_tile_len = mx;
        unsigned int run = 0;
        for (int cidx = NCLS - 1; cidx >= 0; --cidx) { start[cidx] = run; run += hist[cidx]; }
        // The sort launches for long lists cover only a prefix of `order` (their workgroups need most
        // of a CU's LDS just to start, and thousands of them doing nothing starve under a concurrent
        // compositor); the prefixes were sized from an earlier frame.  Lists >= 8192 / >= 2048 keys
        // form whole length classes, i.e. exact prefixes of `order`.
        const unsigned int ge8192 = start[cls_of(8192u)] + hist[cls_of(8192u)];
        const unsigned int ge2048 = start[cls_of(2048u)] + hist[cls_of(2048u)];
        const unsigned int ge16384 = start[cls_of(16384u)] + hist[cls_of(16384u)];
        status->n_ge8192 = ge8192; status->n_ge2048 = ge2048; status->n_ge16384 = ge16384;
        if (status->overflow == 0 && (ge8192 > grid_big || ge2048 > grid_mid || ge16384 > grid_long)) status->overflow = 3u;
        status->n_fallback = 0; status->n_sort_fallback = 0; status->n_near_tiles = 0; status->n_near_fallback = 0; status->n_large = 0; status->n_window = 0;
        status->redone = 0u;                          // (the ring entry may have carried a redone one-pass frame before)
        status->arrived = 1u;
        if (host_status) *host_status = *status;      // (n_visible / n_singular: K1's atomics, complete before this kernel)
    }
    __syncthreads();   // lens[] written above by this workgroup are visible to it
    for (unsigned int k = tid; k < m; k += 1024) order[atomicAdd(&start[cls_of(lens[k])], 1u)] = k;
}

// The key-buffer regions of a slot's NEXT one-pass frame, from the lists of this one: tile k gets room for its list
// plus a half plus 512 keys (a camera that moves between the two frames shifts lists from tile to tile), regions back to
// back: next_layout[0 .. m], and next_counts[k] = next_layout[k] -- the cursors start at their regions.  (Per-tile
// buckets of one fixed stride -- what this replaces -- had to hold the LONGEST list in every tile: 1.07 GB per frame slot
// at 1080p, 9 GB at 4K with 6 M Gaussians, times two buffers, times four slots in flight; the regions take ~1.6 x the
// frame's pairs.)  A layout that would outgrow the buffer is cut off at key_entries -- the tiles behind the cut get
// empty regions, their frame is skipped like any other that outgrows its storage -- and layout_total tells the host
// how many entries the layout asked for.  One workgroup; every thread a contiguous run of tiles.
// An empty current layout (all zeros, cursors from zero) makes this the bootstrap: a frame whose keys all fell on the
// floor still counted its pairs exactly.
__device__ __forceinline__ unsigned int region_for(unsigned int len) { return (len + (len >> 1) + 512u + 63u) & ~63u; }
template <int NT>
__device__ __forceinline__ void build_layout(unsigned int m, const unsigned int* __restrict__ counts, const unsigned int* __restrict__ layout,
                                             unsigned int* __restrict__ next_layout, unsigned int* __restrict__ next_counts,
                                             unsigned int key_entries, FrameStatus* __restrict__ status, FrameStatus* __restrict__ host_status,
                                             unsigned long long* wsum /* LDS, NT / 64 */, float spare_max,
                                             unsigned int tiles_x = 0u, unsigned int motion_radius = 0u, unsigned short* mv = nullptr /* LDS, 2 m entries */) {
    const unsigned int tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const unsigned int C = (m + NT - 1u) / NT, k0 = min(tid * C, m), k1 = min(k0 + C, m);
    // A MOVING camera (motion_radius > 0 tiles: how far the image shifts until the frame that will use these regions): a tile's
    // margin is sized from the longest list within that distance, not from its own -- the list that will be over this tile two
    // frames on is over a neighbour now.  (Sized from its own list, a tile beside a surface got 512 keys of room x the spare
    // factor and the surface's 10 000-key list two frames later: on the trained-like scene at 3 degrees a frame nearly every
    // frame outgrew its regions and paid a count pass or a second binning.)  Separable maximum through LDS: rows, then columns;
    // lengths saturate at 65535 keys.  mv[0 .. m) ends up holding the filtered length of every tile.
    const bool moving = motion_radius != 0u && mv != nullptr && tiles_x != 0u;
    if (moving) {
        unsigned short* const raw = mv;
        unsigned short* const hmx = mv + m;
        for (unsigned int k = tid; k < m; k += NT) raw[k] = (unsigned short)min(counts[k] - layout[k], 65535u);
        __syncthreads();
        const int R = (int)motion_radius, TX = (int)tiles_x, TY = (int)((m + tiles_x - 1u) / tiles_x);
        // (the loops run over the clipped window only -- no per-tap clamp -- and are unrolled eight taps at a time so that the LDS
        // reads of a tile are in flight together: as plain loops they were a chain of dependent reads, as 25 fixed taps with
        // clamped indices five instructions a tap; 37 us of a synchronous frame's scan launch either way)
        for (unsigned int k = tid; k < m; k += NT) {
            const int ty = (int)(k / tiles_x), tx = (int)(k - (unsigned int)ty * tiles_x);
            const unsigned short* const rowp = raw + ty * TX;
            const int hi = min(tx + R, TX - 1);
            unsigned int v = 0u;
#pragma unroll 8
            for (int x = max(tx - R, 0); x <= hi; ++x) v = max(v, (unsigned int)rowp[x]);
            hmx[k] = (unsigned short)v;
        }
        __syncthreads();
        for (unsigned int k = tid; k < m; k += NT) {
            const int ty = (int)(k / tiles_x), tx = (int)(k - (unsigned int)ty * tiles_x);
            const unsigned short* const colp = hmx + tx;
            const int hi = min(ty + R, TY - 1);
            unsigned int v = 0u;
#pragma unroll 8
            for (int y = max(ty - R, 0); y <= hi; ++y) v = max(v, (unsigned int)colp[y * TX]);
            raw[k] = (unsigned short)v;            // (every thread reads hmx only: raw is free to take the result)
        }
        __syncthreads();
    }
    // what a tile's REGION is sized from: its own list, or the longest list around it
    auto sized_from = [&](unsigned int k, unsigned int len) -> unsigned int { return moving ? max(len, (unsigned int)mv[k]) : len; };
    // A thread's tiles are visited three times (sum of the regions; sum of the grown regions; offsets).  Up to 32 tiles per
    // thread (1080p with 256 threads) their regions stay in registers after the first visit; more are re-read, eight
    // tiles' loads in flight at a time -- the kernel is a chain of round trips to L2 otherwise, and the scan's launch
    // (the frame's chain K1 -> scan -> sort) ends when this workgroup does.
    constexpr unsigned int KEEP = 32u;
    const bool keep = C <= KEEP;
    unsigned int kept[KEEP];                       // the tiles' list lengths
    bool loaded = false;
    auto each_len = [&](auto&& each) {
        if (keep) {
            if (!loaded) {
#pragma unroll
                for (unsigned int u = 0; u < KEEP; ++u) { const unsigned int k = k0 + u; kept[u] = k < k1 ? counts[k] - layout[k] : 0u; }
                loaded = true;
            }
#pragma unroll
            for (unsigned int u = 0; u < KEEP; ++u) if (k0 + u < k1) each(k0 + u, kept[u]);
            return;
        }
        for (unsigned int kb = k0; kb < k1; kb += 8u) {
            unsigned int c[8], l[8];
#pragma unroll
            for (unsigned int u = 0; u < 8u; ++u) { const unsigned int k = kb + u; c[u] = k < k1 ? counts[k] : 0u; l[u] = k < k1 ? layout[k] : 0u; }
#pragma unroll
            for (unsigned int u = 0; u < 8u; ++u) if (kb + u < k1) each(kb + u, c[u] - l[u]);
        }
    };
    // block-wide exclusive scan of one value per thread (and the total); two barriers
    auto block_scan = [&](unsigned long long local, unsigned long long& total) -> unsigned long long {
        unsigned long long v = local;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned long long t = (unsigned long long)__shfl_up((long long)v, o);
            if ((int)lane >= o) v += t;
        }
        __syncthreads();                                   // (wsum may still be read from the previous scan)
        if (lane == 63u) wsum[wave] = v;
        __syncthreads();
        unsigned long long run = v - local;
        total = 0;
#pragma unroll
        for (unsigned int w = 0; w < NT / 64u; ++w) { const unsigned long long x = wsum[w]; total += x; if (w < wave) run += x; }
        return run;
    };
    // (two sums in one scan: what the regions ask for, and what the lists themselves take -- a list rounded up to 64 keys --
    // in units of 64 keys, 32 bits each: 2^38 keys)
    auto exact_for = [](unsigned int len) -> unsigned int { return (len + 63u) & ~63u; };
    unsigned long long local = 0, both = 0;
    each_len([&](unsigned int k, unsigned int len) { local += ((unsigned long long)(region_for(sized_from(k, len)) >> 6) << 32) | (unsigned long long)(exact_for(len) >> 6); });
    (void)block_scan(local, both);
    const unsigned long long total = (both >> 32) << 6, total_exact = (both & 0xffffffffull) << 6;
    // The buffer is there: what the regions do not ask for is handed out in proportion (up to four times a region's
    // size), so that a camera that jumps finds room in the tiles its lists move to -- a frame that outgrows a region is
    // binned again (the redo launches), or skipped and redone, or lost if it was asynchronous.  C3: 24 M entries for
    // regions that ask for 16 M.  Regions that ask for MORE than the buffer holds, while the lists themselves fit, give up
    // their margins in proportion instead of being cut off: this frame has room for every key (the next one is more
    // likely to be binned again), and layout_total tells the host what a buffer with the full margins takes.
    const float spare = total ? fminf(spare_max, (float)key_entries / (float)total) : 1.0f;
    const bool squeeze = total > (unsigned long long)key_entries && total_exact <= (unsigned long long)key_entries;
    const float keep_frac = squeeze ? (float)((unsigned long long)key_entries - total_exact) / (float)(total - total_exact) * (1.0f - 1.0f / 1048576.0f) : 0.0f;
    auto grown = [&](unsigned int k, unsigned int len) -> unsigned int {
        const unsigned int r = region_for(sized_from(k, len));
        if (squeeze) { const unsigned int e = exact_for(len); return e + ((unsigned int)((float)(r - e) * keep_frac) & ~63u); }   // (rounded down: the sum stays within key_entries)
        return spare > 1.0f ? max(r, (unsigned int)((float)r * spare) & ~63u) : r;    // (the sum stays within key_entries: every term is rounded down)
    };
    local = 0;
    each_len([&](unsigned int k, unsigned int len) { local += grown(k, len); });
    unsigned long long dummy;
    unsigned long long run = block_scan(local, dummy);
    each_len([&](unsigned int k, unsigned int len) {
        const unsigned int off = (unsigned int)min(run, (unsigned long long)key_entries);
        next_layout[k] = off; next_counts[k] = off;
        run += grown(k, len);
    });
    if (tid == NT - 1u) next_layout[m] = (unsigned int)min(run, (unsigned long long)key_entries);      // (the last thread's run ends the last region)
    // What the host grows the key buffer to is what the regions sized from each tile's OWN list ask for: the motion filter's
    // larger margins live on the buffer's spare room and give way when there is none (squeeze), they are not a reason to make
    // the buffer larger (the first build reported the filtered total: 48 M -> 110 M entries per slot on the surface scene).
    unsigned long long reported = total;
    if (moving) {
        unsigned long long own = 0, all = 0;
        each_len([&](unsigned int, unsigned int len) { own += (unsigned long long)(region_for(len) >> 6); });
        (void)block_scan(own, all);
        reported = all << 6;
    }
    if (tid == 0u) {
        if (status) status->layout_total = reported;
        if (host_status) host_status->layout_total = reported;
    }
}
// (on its own: the bootstrap of a slot without a layout; otherwise the second workgroup of the scan's launch)
// (NT = 256 for the launch in a frame's overflow-redo chain: nine times in ten it reads one word and leaves, and a 16-wave
// workgroup first waits for a CU with sixteen free wave slots on a chip the compositor fills -- 150 us median on the binning
// chain of a moving camera's frame, rocprofv3 trace of tools/motion_probe.py; four waves find room at once)
template <int NT>
__global__ __launch_bounds__(NT) void layout_kernel(unsigned int m, const unsigned int* __restrict__ counts,
                                                      const unsigned int* __restrict__ layout, unsigned int* __restrict__ next_layout,
                                                      unsigned int* __restrict__ next_counts, unsigned int key_entries,
                                                      FrameStatus* __restrict__ status, FrameStatus* __restrict__ host_status, float spare_max,
                                                      const FrameStatus* __restrict__ redo_gate, unsigned int* __restrict__ large_count) {
    __shared__ unsigned long long wsum[NT / 64];
    if (redo_gate != nullptr && redo_gate->overflow != 2u) return;        // (a redo launch of a frame that needs none)
    if (large_count != nullptr && threadIdx.x == 0u) { large_count[0] = 0u; large_count[1] = 0u; }   // (a count-first frame: its count pass's list has been counted; empty for its K1)
    build_layout<NT>(m, counts, layout, next_layout, next_counts, key_entries, status, host_status, wsum, spare_max);
}

// The scan of one-pass binning has no prefix sum to do (a list starts at its bucket): lengths, the
// frame totals, and the longest-first tile order.  The order is a counting sort by length class whose
// counters are per-wave rows of LDS (8160 atomics on a handful of shared class counters serialised:
// most of the old kernel's 16 us); thread t keeps the same tiles in both passes, so its wave's row
// offsets are its own.
// (256 threads, not 1024: the kernel runs while the previous frame's compositor still fills the chip,
// and a 16-wave workgroup waits for a CU with 16 free wave slots -- 48 us on average instead of 11)
template <int SCAN_NT>
__global__ __launch_bounds__(SCAN_NT) void scan_bucket_kernel(unsigned int m, unsigned int* __restrict__ counts,
                                                           unsigned int* __restrict__ offsets,
                                                           unsigned int* __restrict__ order, unsigned int* __restrict__ lens,
                                                           FrameStatus* __restrict__ status, const unsigned int* __restrict__ layout,
                                                           unsigned int grid_big, unsigned int grid_mid, unsigned int grid_long,
                                                           unsigned int cls_in_lds, FrameStatus* __restrict__ host_status,
                                                           unsigned int* __restrict__ next_layout, unsigned int* __restrict__ next_counts,
                                                           unsigned int key_entries, float spare_max, unsigned int redo_only,
                                                           unsigned int* __restrict__ off2, unsigned int cap2, unsigned int* __restrict__ large_count,
                                                           unsigned int tiles_x, unsigned int motion_radius) {
    constexpr int NCLS = 64;
    if (redo_only && status->overflow != 2u) return;       // (the second scan of a frame that was binned again: see enqueue_frame)
    // (the frame's large-splat list has been binned -- bin_large_kernel, in front of this launch: empty for the slot's next K1;
    // how many there were goes into the status: the host decides from it whether the next frames keep a list at all)
    unsigned int n_large_seen = 0u, n_window_seen = 0u;
    if (large_count != nullptr && blockIdx.x == 0u && threadIdx.x == 0u) { n_large_seen = large_count[0]; n_window_seen = large_count[1]; large_count[0] = 0u; large_count[1] = 0u; }
    // Workgroup 1 of the launch (when there is one) builds the regions of the next frame on this binning stream from
    // the same cursors, beside the scan: no launch of its own, nothing added to the chain K1 -> scan -> sort.
    if (blockIdx.x == 1u) {
        __shared__ unsigned long long lsum[SCAN_NT / 64];
        extern __shared__ unsigned char dyn_lds[];            // (the launch gives 4 m bytes when motion_radius != 0)
        build_layout<SCAN_NT>(m, counts, layout, next_layout, next_counts, key_entries, status, host_status, lsum, spare_max,
                              tiles_x, motion_radius, reinterpret_cast<unsigned short*>(dyn_lds));
        return;
    }
    __shared__ unsigned int row[SCAN_NT / 64][NCLS];
    __shared__ unsigned int start[NCLS];
    __shared__ unsigned long long wsum[SCAN_NT / 64];
    __shared__ unsigned int wmax[SCAN_NT / 64];
    // The SECOND key buffer holds room only for the lists that use it -- the lists of more than 2048 keys: their sorted near
    // selection, or the scatter space of a full sort -- handed out here, list by list, from a counter (any order will do).
    // (A mirror of the first buffer, regions and all, used to stand behind every frame slot: 768 MB each on C3 for the 43 MB
    // its long lists hold.)  pool2 beyond cap2: the frame is skipped like any other that outgrows its storage, the host
    // grows the buffer (overflow = 4).
    __shared__ unsigned int pool2;
    const unsigned int tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    auto cls_of = [](unsigned int c) -> unsigned int {
        if (c == 0) return 0u;
        unsigned int msb = 31u - (unsigned int)__clz((int)c);
        unsigned int half = msb ? ((c >> (msb - 1)) & 1u) : 0u;
        return min(1u + 2u * msb + half, (unsigned int)NCLS - 1u);
    };
    row[wave][lane] = 0;
    if (tid == 0u) pool2 = 0u;
    __syncthreads();
    unsigned long long sum = 0;
    unsigned int mx = 0, over = 0;
    // The kernel is a latency chain on the frame's critical path (K1 -> scan -> sort): a thread's counts are
    // fetched eight at a time (eight loads in flight instead of one dependent round trip per tile), and the
    // length class of every tile is parked in LDS for the second pass instead of being read back from memory.
    extern __shared__ unsigned char cls_lds[];            // m bytes when the launch provides them (cls_in_lds)
    constexpr unsigned int U = 8;
    for (unsigned int base = 0; base < m; base += U * SCAN_NT) {
        // counts[k] is the CURSOR of tile k's region [layout[k], layout[k + 1]): the list's length is how far it moved
        // (it may have moved past the region's end: those keys were not stored and the frame is skipped).  The cursors
        // are left alone: layout_kernel re-initialises them together with the regions of the slot's next frame.
        unsigned int c[U], l0[U], l1[U];
#pragma unroll
        for (unsigned int u = 0; u < U; ++u) {
            const unsigned int k = base + u * SCAN_NT + tid;
            c[u] = (k < m) ? counts[k] : 0u; l0[u] = (k < m) ? layout[k] : 0u; l1[u] = (k < m) ? layout[k + 1u] : 0u;
        }
#pragma unroll
        for (unsigned int u = 0; u < U; ++u) {
            const unsigned int k = base + u * SCAN_NT + tid;
            if (k < m) {
                const unsigned int raw = c[u] - l0[u], cap = l1[u] - l0[u];
                const unsigned int len = min(raw, cap);
                over |= raw > cap ? 1u : 0u;
                c[u] = raw;
                lens[k] = len;
                offsets[k] = l0[u];
                sum += c[u]; mx = max(mx, c[u]);
                const unsigned int cls = cls_of(len);
                if (cls_in_lds) cls_lds[k] = (unsigned char)cls;
                atomicAdd(&row[wave][cls], 1u);
            }
            // the long lists' room in the second key buffer, a wave's tiles at a time: one atomic on the counter per wave and
            // step (every lane on it, one after the other, was 9 us of this 28-us kernel)
            if (off2 != nullptr) {
                const unsigned int len = (k < m) ? min(c[u], l1[u] - l0[u]) : 0u;
                const unsigned int want = len > 2048u ? (len + 63u) & ~63u : 0u;
                const unsigned int inc = wave_inclusive_sum(want);
                const unsigned int total = (unsigned int)wave_last((int)inc);
                unsigned int room = 0u;
                if (total != 0u) {
                    if (lane == 63u) room = atomicAdd(&pool2, total);
                    room = (unsigned int)wave_last((int)room);
                }
                if (k < m) off2[k] = want ? room + inc - want : 0u;
            }
        }
    }
    mx |= over ? 0x80000000u : 0u;           // (a list is far shorter than 2^31: the flag rides on the maximum)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        mx = max(mx, (unsigned int)__shfl_xor((int)mx, o));
        sum += (unsigned long long)__shfl_xor((long long)sum, o);
    }
    if (lane == 0) { wsum[wave] = sum; wmax[wave] = mx; }
    __syncthreads();
    if (tid < NCLS) {                      // class tid: its total, and every wave's offset inside the class
        unsigned int acc = 0;
#pragma unroll
        for (int w = 0; w < SCAN_NT / 64; ++w) { const unsigned int t = row[w][tid]; row[w][tid] = acc; acc += t; }
        start[tid] = acc;                  // (total for now)
    }
    __syncthreads();
    if (tid == 0) {
        unsigned long long tot = 0;
        for (int w = 0; w < SCAN_NT / 64; ++w) { tot += wsum[w]; mx = max(mx, wmax[w]); }
        unsigned int run = 0, ge[3] = {0, 0, 0};
        const unsigned int c8 = cls_of(8192u), c2 = cls_of(2048u), c16 = cls_of(16384u);
        for (int cidx = NCLS - 1; cidx >= 0; --cidx) {
            const unsigned int t = start[cidx];
            start[cidx] = run; run += t;
            if ((unsigned int)cidx == c16) ge[0] = run;
            if ((unsigned int)cidx == c8) ge[1] = run;
            if ((unsigned int)cidx == c2) ge[2] = run;
        }
        offsets[m] = (unsigned int)tot;
        status->n_pairs = tot;
        status->max_tile_len = mx & 0x7fffffffu;
        status->n_ge16384 = ge[0]; status->n_ge8192 = ge[1]; status->n_ge2048 = ge[2];
        status->overflow = (mx & 0x80000000u) ? 2u : ((ge[1] > grid_big || ge[2] > grid_mid || ge[0] > grid_long) ? 3u : 0u);
        status->n_long_keys = pool2;            // (complete: the barriers above)
        status->arrived = 1u;
        if (off2 != nullptr && pool2 > cap2 && status->overflow == 0u) status->overflow = 4u;
        // this kernel initialises the frame's status (nothing before it in a one-pass frame touches it) ...
        status->n_visible = 0; status->n_singular = 0;
        status->n_fallback = 0; status->n_sort_fallback = 0; status->n_near_tiles = 0; status->n_near_fallback = 0; status->n_large = n_large_seen; status->n_window = n_window_seen;
        status->redone = redo_only ? 1u : 0u;  // (1: this frame outgrew its regions and was binned again on the device)
        status->n_blocks_culled = 0;
        // ... and delivers it to the host: everything an asynchronous frame reports is decided here (layout_total, the
        // last word, belongs to the workgroup that builds the layout: both copies are its to write)
        if (host_status) {
            static_assert(offsetof(FrameStatus, layout_total) + 8 == sizeof(FrameStatus), "layout_total is the last word");
            const unsigned long long* src = reinterpret_cast<const unsigned long long*>(status);
            unsigned long long* dst = reinterpret_cast<unsigned long long*>(host_status);
            // (system scope: the host peeks at frames in flight -- enqueue_frame, the redo's arming -- without waiting for anything)
            for (unsigned int q = 0; q < offsetof(FrameStatus, layout_total) / 8u; ++q)
                __hip_atomic_store(&dst[q], src[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __threadfence_system();
        }
    }
    __syncthreads();
    // second pass: thread t keeps the same tiles, so its wave's row offsets are its own
    for (unsigned int base = 0; base < m; base += U * SCAN_NT) {
#pragma unroll
        for (unsigned int u = 0; u < U; ++u) {
            const unsigned int k = base + u * SCAN_NT + tid;
            if (k < m) {
                const unsigned int cls = cls_in_lds ? (unsigned int)cls_lds[k] : cls_of(lens[k]);    // (written by this very thread)
                order[start[cls] + atomicAdd(&row[wave][cls], 1u)] = k;
            }
        }
    }
}

// K2 -- one thread per Gaussian slot: claim a slot in every overlapped tile's bucket and write the
// 64-bit key (depth_key << 32 | ORIGINAL index).  Bucket order is arbitrary; K3 fixes it.
constexpr int EMIT_G = 1;              // Gaussians per thread in K2 (4 measured slower: 0.10 -> 0.17 ms)
__global__ __launch_bounds__(256) void emit_kernel(FrameConst fc, const float* __restrict__ depth,
                                                   const ushort4* __restrict__ rect, const unsigned int* __restrict__ orig,
                                                   const unsigned int* __restrict__ vislist,
                                                   unsigned int* __restrict__ cursor, unsigned long long* __restrict__ keys,
                                                   const FrameStatus* __restrict__ status) {
    __shared__ BinShared sh;
    if (status->overflow) return;
    const unsigned long long nvis = status->n_visible;        // K1 compacted the slab's slots into vislist
    const unsigned long long base = (unsigned long long)blockIdx.x * (256ull * EMIT_G);
    if (base >= nvis) return;
    bool vis[EMIT_G];
    int tx0[EMIT_G], tx1[EMIT_G], ty0[EMIT_G], ty1[EMIT_G];
    unsigned long long key[EMIT_G];
#pragma unroll
    for (int g = 0; g < EMIT_G; ++g) {
        const unsigned long long j = base + (unsigned long long)g * 256ull + threadIdx.x;
        vis[g] = j < nvis;
        tx0[g] = 0; tx1[g] = -1; ty0[g] = 0; ty1[g] = -1; key[g] = 0;
        if (vis[g]) {
            const unsigned int i = vislist[j];
            ushort4 rc = rect[i];
            int y0 = max((int)rc.z, fc.row_px0), y1 = min((int)rc.w, fc.row_px1 - 1);
            tx0[g] = rc.x >> 4; tx1[g] = rc.y >> 4; ty0[g] = (y0 >> 4) - fc.tile_row0; ty1[g] = (y1 >> 4) - fc.tile_row0;
            key[g] = ((unsigned long long)depth_key(depth[i]) << 32) | (unsigned long long)i;
        }
    }
    bin_block<true, EMIT_G>(sh, vis, tx0, tx1, ty0, ty1, fc.tiles_x, cursor, keys, key);
}

// ---------------------------------------------------------------------------
// K3 -- per-tile sort of the 64-bit keys (ascending depth key, ties by index == the
// reference's stable ascending-z order).  Bitonic network in its all-ascending "flip" form, so
// that a list of any length n sorts in place: slots >= n behave as +inf and never move.
// ---------------------------------------------------------------------------
// Order of two keys = (depth_key << 32 | SLOT): ascending depth, and among equal depths ascending ORIGINAL index
// (the reference's stable sort, src/gaussians.rs:302-303).  The key carries the Morton slot -- records and scene
// planes live in slot order -- so a tie (exact f32 depth equality: a few per ten thousand keys) looks the original
// indices up; the branch is taken by the lanes that hold a tie only.
__device__ __forceinline__ bool key_gt(unsigned long long a, unsigned long long b, const unsigned int* __restrict__ orig) {
    const unsigned int da = (unsigned int)(a >> 32), db = (unsigned int)(b >> 32);
    if (da != db) return da > db;
    return orig[(unsigned int)a] > orig[(unsigned int)b];
}
__device__ __forceinline__ void cmp_swap(unsigned long long* s, unsigned int i, unsigned int p, const unsigned int* __restrict__ orig) {
    unsigned long long a = s[i], b = s[p];
    if (key_gt(a, b, orig)) { s[i] = b; s[p] = a; }
}

__device__ __forceinline__ void bitonic_sort(unsigned long long* s, unsigned int n, unsigned int tid, unsigned int nt,
                                             const unsigned int* __restrict__ orig) {
    unsigned int P = 1;
    while (P < n) P <<= 1;
    const unsigned int half = P >> 1;
    for (unsigned int k = 2; k <= P; k <<= 1) {
        const unsigned int hk = k >> 1;
        for (unsigned int t = tid; t < half; t += nt) {
            unsigned int off = t & (hk - 1), base = (t - off) << 1;   // block start = (t / hk) * k
            unsigned int i = base + off, p = base + (k - 1 - off);
            if (p < n) cmp_swap(s, i, p, orig);
        }
        __syncthreads();
        for (unsigned int j = k >> 2; j > 0; j >>= 1) {
            for (unsigned int t = tid; t < half; t += nt) {
                unsigned int lowbits = t & (j - 1);
                unsigned int i = ((t - lowbits) << 1) | lowbits, p = i + j;
                if (p < n) cmp_swap(s, i, p, orig);
            }
            __syncthreads();
        }
    }
}


// Digit plan of a list whose depth keys span [dmin, dmax]: the keys are sorted on (depth - dmin),
// which has V = bits(dmax - dmin) significant bits, in P = ceil(V/wmax) passes of w = ceil(V/P) bits
// each (wmax = 9 when the list leaves room in LDS for 512-bin histograms -- float depths over a few
// octaves have V = 24..27, three passes instead of four -- else 8).  Subtracting the minimum matters beyond saving passes: the raw top byte (sign + exponent)
// takes two or three values per tile, and a pass whose 64 lanes all hit the same two histogram
// words serialises in the LDS atomic unit -- the degenerate top passes, not the evenly spread low
// ones, made a 10 000-key list take 60 us.
struct DigitPlan { unsigned int dmin; int P; int w; int lb; };   // lb: log2 of the histogram size (>= w)
__device__ __forceinline__ DigitPlan plan_digits(unsigned int dmin, unsigned int dmax, bool even_passes, int wmax) {
    const unsigned int range = dmax - dmin;
    const int V = range ? 32 - __clz((int)range) : 0;
    int P = (V + wmax - 1) / wmax;
    if (even_passes && (P & 1)) ++P;                       // ping-pong buffers: finish where we started
    DigitPlan pl;
    pl.dmin = dmin; pl.P = P; pl.w = P ? (V + P - 1) / P : 0; pl.lb = wmax;
    return pl;
}
__device__ __forceinline__ unsigned int digit_of(unsigned long long key, const DigitPlan& pl, int pass) {
    return (((unsigned int)(key >> 32) - pl.dmin) >> (pl.w * pass)) & ((1u << pl.w) - 1u);
}
// Block-wide min / max of the depth halves; every thread passes the min / max of the keys it has seen.
// `scratch` = two LDS words.  Ends with a barrier.
__device__ __forceinline__ DigitPlan block_digit_plan(unsigned int mn, unsigned int mx, unsigned int* scratch,
                                                      unsigned int tid, bool even_passes, int wmax) {
    if (tid == 0) { scratch[0] = 0xffffffffu; scratch[1] = 0u; }
    __syncthreads();
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        mn = min(mn, (unsigned int)__shfl_xor((int)mn, o));
        mx = max(mx, (unsigned int)__shfl_xor((int)mx, o));
    }
    if ((tid & 63u) == 0) { atomicMin(&scratch[0], mn); atomicMax(&scratch[1], mx); }
    __syncthreads();
    const DigitPlan pl = plan_digits(scratch[0], scratch[1], even_passes, wmax);
    __syncthreads();                                       // scratch is reused by the passes
    return pl;
}

// In-LDS LSD radix sort of the keys by their DEPTH half (upper 32 bits): pl.P passes of pl.w bits.
// Wave w owns the contiguous chunk [w*C, (w+1)*C) of the array, striped over its lanes, so the
// (wave, round, lane) processing order is the memory order and the sort is stable:
//   count    per-wave digit histogram; lanes with the same digit find each other with 8 ballots
//   scan     per-digit exclusive offsets over waves, then over digits
//   scatter  keys (held in registers since the load) go to offset + rank-within-round
// Depth ties are left in bucket order, which is arbitrary; the caller checks the full 64-bit
// order afterwards and falls back to the exact bitonic network if anything is out of place.
template <int NT, int EMAX>
__device__ __forceinline__ void radix_sort_depth(unsigned long long* s, unsigned int* hist /*[NT/64][nb]*/,
                                                 unsigned int* tot /*[nb]*/, unsigned int* dbase /*[nb]*/,
                                                 unsigned int n, unsigned int tid, const DigitPlan pl) {
    constexpr unsigned int NW = NT / 64;
    const unsigned int wave = tid >> 6, lane = tid & 63u;
    const unsigned int C = (((n + NW - 1) / NW) + 63u) & ~63u;       // chunk per wave, multiple of 64
    const unsigned int w0 = wave * C, w1 = min(w0 + C, n);
    const unsigned int E = C >> 6;                                   // rounds per wave (<= EMAX)
    const unsigned int nb = 1u << pl.lb;                              // 256 or 512 bins
    unsigned int* myhist = hist + wave * nb;
    for (int pass = 0; pass < pl.P; ++pass) {
        for (unsigned int q = lane; q < nb; q += 64u) myhist[q] = 0;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // every wave pulls its keys into registers ...
        unsigned long long k[EMAX];
        unsigned int rank[EMAX];
#pragma unroll
        for (int e = 0; e < EMAX; ++e) {
            const unsigned int i = w0 + (unsigned int)e * 64u + lane;
            k[e] = ((unsigned int)e < E && i < w1) ? s[i] : ~0ull;
        }
        // ... and counts them with ONE returning LDS atomic per key on its histogram row: the value
        // returned is the key's rank among the wave's keys of the same digit.  The rounds are issued
        // in order and the LDS serialises the lanes of one instruction in lane order on this
        // hardware, so ranks follow the memory order and the pass is stable; that is NOT an
        // architectural guarantee, so the caller's 64-bit order check (+ exact fallback) stays the
        // safety net.
#pragma unroll
        for (int e = 0; e < EMAX; ++e) {
            const unsigned int i = w0 + (unsigned int)e * 64u + lane;
            rank[e] = 0;
            if ((unsigned int)e < E && i < w1) rank[e] = atomicAdd(&myhist[digit_of(k[e], pl, pass)], 1u);
        }
        __syncthreads();
        // scan over waves (one thread per digit), then over digits (wave 0, nb/64 digits per lane)
        for (unsigned int d = tid; d < nb; d += NT) {
            unsigned int acc = 0;
#pragma unroll
            for (unsigned int w = 0; w < NW; ++w) {
                const unsigned int t = hist[w * nb + d];
                hist[w * nb + d] = acc;
                acc += t;
            }
            tot[d] = acc;
        }
        __syncthreads();
        if (tid < 64) {
            const unsigned int per = nb >> 6;                        // 4 or 8
            unsigned int t[8], sum = 0;
#pragma unroll
            for (unsigned int j = 0; j < 8; ++j) { t[j] = (j < per) ? tot[per * tid + j] : 0u; sum += t[j]; }
            unsigned int v = sum;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const unsigned int u = (unsigned int)__shfl_up((int)v, o);
                if ((int)tid >= o) v += u;
            }
            unsigned int ex = v - sum;
#pragma unroll
            for (unsigned int j = 0; j < 8; ++j)
                if (j < per) { dbase[per * tid + j] = ex; ex += t[j]; }
        }
        __syncthreads();
        // scatter in place: digit base + this wave's offset within the digit + rank within the wave
#pragma unroll
        for (int e = 0; e < EMAX; ++e) {
            const unsigned int i = w0 + (unsigned int)e * 64u + lane;
            if ((unsigned int)e < E && i < w1) {
                const unsigned int d = digit_of(k[e], pl, pass);
                s[dbase[d] + myhist[d] + rank[e]] = k[e];
            }
        }
        __syncthreads();
    }
}

// Radix by depth, then put depth ties into index order.  With thousands of keys per tile exact f32
// depth ties are routine (birthday effect) but the tied runs are 2-3 keys long: a few odd-even
// transposition passes over adjacent keys fix them (any inversion left after the radix passes is
// between equal depths, so swapping inverted neighbours is exactly a bubble sort of those runs).
// Long runs (many Gaussians at one depth) fall back to the exact bitonic network.
template <int NT, int EMAX>
__device__ __forceinline__ void sort_keys_lds(unsigned long long* s, unsigned int* hist, unsigned int* tot,
                                              unsigned int* dbase, unsigned int n, unsigned int tid,
                                              FrameStatus* status, const DigitPlan pl, const unsigned int* __restrict__ orig) {
    radix_sort_depth<NT, EMAX>(s, hist, tot, dbase, n, tid, pl);
    for (int it = 0; it < 6; ++it) {
        bool swapped = false;
#pragma unroll
        for (unsigned int parity = 0; parity < 2; ++parity) {
            for (unsigned int i = parity + 2u * tid; i + 1 < n; i += 2u * NT) {
                const unsigned long long x = s[i], y = s[i + 1];
                if (key_gt(x, y, orig)) { s[i] = y; s[i + 1] = x; swapped = true; }
            }
            __syncthreads();
        }
        if (!__syncthreads_or(swapped ? 1 : 0)) return;      // a full pass without swaps: sorted
    }
    if (tid == 0) atomicAdd(&status->n_sort_fallback, 1ull);
    bitonic_sort(s, n, tid, NT, orig);
}

// The same radix sort for lists that do not fit in LDS: keys stay in global memory (L2-resident),
// each pass scatters from `src` to `dst` (a second key buffer at the same offsets); four passes
// leave the result in `src`.  One workgroup; __syncthreads() orders its own global accesses.
template <int NT>
__device__ __forceinline__ void radix_sort_depth_global(unsigned long long* src, unsigned long long* dst,
                                                        unsigned int* hist, unsigned int* tot, unsigned int* dbase,
                                                        unsigned int n, unsigned int tid, const DigitPlan pl) {
    constexpr unsigned int NW = NT / 64;
    const unsigned int wave = tid >> 6, lane = tid & 63u;
    const unsigned int C = (((n + NW - 1) / NW) + 63u) & ~63u;
    const unsigned int w0 = wave * C, w1 = min(w0 + C, n);
    const unsigned int E = C >> 6;
    unsigned int* myhist = hist + wave * 256u;
    for (int pass = 0; pass < pl.P; ++pass) {              // pl.P is even: the result ends in `src`
#pragma unroll
        for (int q = 0; q < 4; ++q) myhist[lane * 4u + q] = 0;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        for (int phase = 0; phase < 2; ++phase) {          // 0: count, 1: scatter
            for (unsigned int e = 0; e < E; ++e) {
                const unsigned int i = w0 + e * 64u + lane;
                if (i < w1) {
                    const unsigned long long key = src[i];
                    const unsigned int d = digit_of(key, pl, pass);
                    const unsigned int rank = atomicAdd(&myhist[d], 1u);     // see radix_sort_depth on stability
                    if (phase == 1) dst[dbase[d] + rank] = key;
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
            if (phase == 0) {
                __syncthreads();
                if (tid < 256) {
                    unsigned int acc = 0;
#pragma unroll
                    for (unsigned int w = 0; w < NW; ++w) {
                        const unsigned int t = hist[w * 256u + tid];
                        hist[w * 256u + tid] = acc;
                        acc += t;
                    }
                    tot[tid] = acc;
                }
                __syncthreads();
                if (tid < 64) {
                    const unsigned int t0 = tot[4 * tid], t1 = tot[4 * tid + 1], t2 = tot[4 * tid + 2], t3 = tot[4 * tid + 3];
                    unsigned int v = t0 + t1 + t2 + t3;
#pragma unroll
                    for (int o = 1; o < 64; o <<= 1) {
                        const unsigned int t = (unsigned int)__shfl_up((int)v, o);
                        if ((int)tid >= o) v += t;
                    }
                    const unsigned int ex = v - (t0 + t1 + t2 + t3);
                    dbase[4 * tid] = ex; dbase[4 * tid + 1] = ex + t0; dbase[4 * tid + 2] = ex + t0 + t1; dbase[4 * tid + 3] = ex + t0 + t1 + t2;
                }
                __syncthreads();
            }
        }
        __syncthreads();
        unsigned long long* t = src; src = dst; dst = t;
    }
}

// keys + one 256-bin histogram row per wave + one row of per-digit totals, which the digit scan turns into the
// per-digit bases IN PLACE (every lane reads its totals before it writes its bases).  At <256, 2048> that is
// 21 504 bytes: seven compositor workgroups leave 9 KB of a CU's LDS free, room for one K1 workgroup (6.2 KB)
// of the next frame beside them.
template <int NT, int CAP>
constexpr unsigned int sort_lds_bytes() { return CAP * 8 + ((NT / 64) * 256 + 256) * 4; }

// Sort one list of n <= CAP keys (global, at gin) through this workgroup's LDS (smem: sort_lds_bytes<NT, CAP>(), of
// which the last `reserve` bytes are left alone), into gout (which may be gin).  Used by sort_tiles_kernel and, for
// the lists of its own tile, by the compositor's workgroup.
// idx_out != nullptr (the compositor's own short list): the sorted ORDER -- the keys' index halves -- is left in LDS
// at idx_out (which may overlap the workspace) for the workgroup that is about to walk the list; the keys go back
// to global memory only if write_back is set (statistics / debug frames read the lists from there).
// gin == nullptr: the keys are in LDS already (s[0 .. n), complete behind a barrier the CALLER has passed -- the nearest
// keys of a long list, select_near); pre_mn / pre_mx then bound their depth halves (any bounds do: they only size the digits).
template <int NT, int CAP>
__device__ __forceinline__ void sort_list_in_lds(unsigned char* smem, const unsigned long long* gin, unsigned long long* gout,
                                                 unsigned int n, unsigned int radix_min, FrameStatus* __restrict__ status,
                                                 const unsigned int* __restrict__ orig,
                                                 unsigned int* idx_out = nullptr, bool write_back = true, unsigned int reserve = 0u,
                                                 unsigned int pre_mn = 0xffffffffu, unsigned int pre_mx = 0u) {
    unsigned long long* s = reinterpret_cast<unsigned long long*>(smem);
    // histograms live right behind the keys in use: a list that leaves room gets 512 bins
    constexpr unsigned int NW = NT / 64;
    const unsigned int keys_bytes = ((n * 8u) + 15u) & ~15u;
    const int lb = (keys_bytes + (NW + 1u) * 512u * 4u + reserve <= sort_lds_bytes<NT, CAP>()) ? 9 : 8;
    unsigned int* hist = reinterpret_cast<unsigned int*>(smem + keys_bytes);
    unsigned int* tot = hist + (NW << lb);
    unsigned int* dbase = tot;             // in place (see sort_lds_bytes)
    unsigned int mn = pre_mn, mx = pre_mx;
    if (gin != nullptr)
    for (unsigned int t0 = threadIdx.x; t0 < n; t0 += 8u * NT) {     // eight loads in flight per thread
        unsigned long long k[8];
#pragma unroll
        for (unsigned int u = 0; u < 8; ++u) { const unsigned int t = t0 + u * NT; k[u] = (t < n) ? gin[t] : 0ull; }
#pragma unroll
        for (unsigned int u = 0; u < 8; ++u) {
            const unsigned int t = t0 + u * NT;
            if (t < n) { s[t] = k[u]; mn = min(mn, (unsigned int)(k[u] >> 32)); mx = max(mx, (unsigned int)(k[u] >> 32)); }
        }
    }
    if (n <= radix_min) {
        __syncthreads();
        bitonic_sort(s, n, threadIdx.x, NT, orig);   // few steps: cheaper than the radix passes
    } else {
        const DigitPlan pl = block_digit_plan(mn, mx, tot, threadIdx.x, false, lb);   // (its barriers publish s[])
        sort_keys_lds<NT, CAP / NT>(s, hist, tot, dbase, n, threadIdx.x, status, pl, orig);
    }
    if (idx_out == nullptr) {
        for (unsigned int t = threadIdx.x; t < n; t += NT) gout[t] = s[t];
        return;
    }
    constexpr unsigned int E = CAP / NT;
    unsigned long long k[E];
#pragma unroll
    for (unsigned int u = 0; u < E; ++u) { const unsigned int t = threadIdx.x + u * NT; k[u] = (t < n) ? s[t] : 0ull; }
    if (write_back) {
#pragma unroll
        for (unsigned int u = 0; u < E; ++u) { const unsigned int t = threadIdx.x + u * NT; if (t < n) gout[t] = k[u]; }
    }
    __syncthreads();                       // every key has been read: the index area may overlap them
#pragma unroll
    for (unsigned int u = 0; u < E; ++u) { const unsigned int t = threadIdx.x + u * NT; if (t < n) idx_out[t] = (unsigned int)k[u]; }
}

// A list of any length through global memory (L2-resident): LSD radix passes by depth from g to h and back (an even
// number of them), then the tie fix-up and, for long runs of one depth, the exact network -- both on global memory.
// Slow (one workgroup, every pass a round trip): the route of last resort.  smem: (NT / 64 + 1) * 256 words.
template <int NT>
__device__ __forceinline__ void sort_list_global(unsigned char* smem, unsigned long long* g, unsigned long long* h, unsigned int n,
                                                 FrameStatus* __restrict__ status, const unsigned int* __restrict__ orig) {
    unsigned int* hist = reinterpret_cast<unsigned int*>(smem);
    unsigned int* tot = hist + (NT / 64) * 256;
    unsigned int* dbase = tot;         // in place
    unsigned int mn = 0xffffffffu, mx = 0u;
    for (unsigned int t = threadIdx.x; t < n; t += NT) {
        const unsigned int d = (unsigned int)(g[t] >> 32);
        mn = min(mn, d); mx = max(mx, d);
    }
    const DigitPlan pl = block_digit_plan(mn, mx, tot, threadIdx.x, true, 8);
    radix_sort_depth_global<NT>(g, h, hist, tot, dbase, n, threadIdx.x, pl);
    bool sorted = false;
    for (int it = 0; it < 6 && !sorted; ++it) {
        bool swapped = false;
#pragma unroll
        for (unsigned int parity = 0; parity < 2; ++parity) {
            for (unsigned int i = parity + 2u * threadIdx.x; i + 1 < n; i += 2u * NT) {
                const unsigned long long x = g[i], y = g[i + 1];
                if (key_gt(x, y, orig)) { g[i] = y; g[i + 1] = x; swapped = true; }
            }
            __syncthreads();
        }
        sorted = !__syncthreads_or(swapped ? 1 : 0);
    }
    if (!sorted) {
        if (threadIdx.x == 0) atomicAdd(&status->n_sort_fallback, 1ull);
        bitonic_sort(g, n, threadIdx.x, NT, orig);       // exact network, slow: only for long runs of equal depth
    }
}

// A list LONGER than the 2048 keys a compositor workgroup sorts in its LDS, put in order by that same workgroup (NT = 256
// threads, the compositor's sort_lds_bytes<NT, 2048>() of LDS) before it composites the tile -- so that no frame
// waits for sort launches whose big workgroups (74 / 147 KB of LDS) only find room on a chip full of compositor
// workgroups once those have drained -- in full, or only its nearest `cap` keys (cap < n: near selection, see select_near):
//   1  a depth range from a 256-key sample of the list; 1024 bins over it (keys outside fall into the end bins)
//   2  histogram (LDS atomics); the largest SUFFIX of bins -- the near end -- that holds at most `cap` keys is the
//      selection (all of them if cap >= n); exclusive scan of the selected bins -> every bin's place in the output
//   3  scatter g -> h (the second key buffer): the selected keys are now grouped by bin, bins in depth order; g is untouched
//   4  consecutive bins are grouped into PARTS of at most 1792 keys (a bin of more than 512 keys is a part of its own, up
//      to 1536 keys) and every part is sorted through LDS by the code that
//      sorts the short lists: into g for a full sort (the list in its region, as a sort launch leaves it), in place in h
//      for a selection (the region keeps the whole unordered list: a repair can still sort it all).
// A selected bin of more than 1536 keys (thousands of Gaussians at one depth), or more than 62 parts, fails: the caller
// takes the whole list the global route.
// partition_long_list does 1-3 and returns the number of parts (their bounds in `pstart`, LDS, the last
// LONG_SORT_RESERVE bytes of the workspace) and the number of selected keys in *m_out, or 0 on failure; the caller runs
// the parts through sort_list_in_lds (the compositor has ONE inlined copy of that code for its short and its long
// lists: a second one cost 24 VGPRs, two workgroups per CU).
constexpr unsigned int LONG_SORT_RESERVE = 256u;
template <int NT>
__device__ __forceinline__ unsigned int partition_long_list(unsigned char* smem, const unsigned long long* __restrict__ g, unsigned long long* __restrict__ h,
                                                            unsigned int n, unsigned int cap, unsigned int* m_out) {
    constexpr unsigned int NB = 1024u, PART_T = 1280u, PART_MAX = 1792u, BIN_MAX = 512u, BIG_PART_MAX = 1536u, PMAX = 63u, RESERVE = LONG_SORT_RESERVE;
    static_assert(NT == 256, "four bins per thread");
    unsigned int* bins = reinterpret_cast<unsigned int*>(smem);             // counts, then exclusive starts
    unsigned int* cur = bins + NB;                                          // scatter cursors
    unsigned int* misc = cur + NB;                                          // 0 min, 1 max, 2 largest selected bin, 3 last part, 4 first bin, 5 selected keys, 8.. wave sums
    unsigned int* pstart = reinterpret_cast<unsigned int*>(smem + sort_lds_bytes<NT, 2048>() - RESERVE);   // [64]
    const unsigned int tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    for (unsigned int q = tid; q < NB; q += NT) { bins[q] = 0u; cur[q] = 0u; }
    if (tid < 64u) pstart[tid] = 0xffffffffu;
    if (tid == 0u) { misc[0] = 0xffffffffu; misc[1] = 0u; misc[2] = 0u; misc[3] = 0u; misc[4] = NB; misc[5] = 0u; }
    unsigned int mn, mx;
    {   // the sample (see select_near)
        const unsigned int d = (unsigned int)(g[(unsigned int)(((unsigned long long)tid * n) / NT)] >> 32);
        mn = d; mx = d;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        mn = min(mn, (unsigned int)__shfl_xor((int)mn, o));
        mx = max(mx, (unsigned int)__shfl_xor((int)mx, o));
    }
    __syncthreads();
    if (lane == 0u) { atomicMin(&misc[0], mn); atomicMax(&misc[1], mx); }
    __syncthreads();
    const unsigned int smin = misc[0], smax = misc[1], quarter = (smax - smin) >> 2;
    const unsigned int lo = smin > quarter ? smin - quarter : 0u;
    const unsigned int hi = smax < 0xffffffffu - quarter ? smax + quarter : 0xffffffffu;
    const unsigned int range = hi - lo;
    const unsigned int sh = range >= NB ? (unsigned int)(32 - __clz((int)range)) - 10u : 0u;      // (range >> sh) < 1024
    auto bin_of = [&](unsigned int d) -> unsigned int { return d < lo ? 0u : min((d - lo) >> sh, NB - 1u); };
    for (unsigned int t0 = tid; t0 < n; t0 += 8u * NT) {              // eight loads in flight per thread
        unsigned int d[8];
#pragma unroll
        for (unsigned int u = 0; u < 8; ++u) { const unsigned int t = t0 + u * NT; d[u] = (t < n) ? (unsigned int)(g[t] >> 32) : 0u; }
#pragma unroll
        for (unsigned int u = 0; u < 8; ++u) if (t0 + u * NT < n) atomicAdd(&bins[bin_of(d[u])], 1u);
    }
    __syncthreads();
    unsigned int c[4], before;                     // this thread's four bins and the keys in the bins in front of them
    {
        unsigned int sum = 0u;
#pragma unroll
        for (unsigned int j = 0; j < 4; ++j) { c[j] = bins[4u * tid + j]; sum += c[j]; }
        unsigned int v = sum;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned int u = (unsigned int)__shfl_up((int)v, o);
            if ((int)lane >= o) v += u;
        }
        if (lane == 63u) misc[8u + wave] = v;
        __syncthreads();
        before = v - sum;
        for (unsigned int w = 0; w < wave; ++w) before += misc[8u + w];
        // the first selected bin: the smallest b with S(b) = n - (keys in bins < b) <= cap
        unsigned int S = n - before;
#pragma unroll
        for (unsigned int j = 0; j < 4; ++j) {
            const unsigned int b = 4u * tid + j;
            const unsigned int cnt_prev = (j == 0u) ? ((tid == 0u) ? 0u : bins[b - 1u]) : c[j - 1u];
            if (S <= cap && (b == 0u || S + cnt_prev > cap)) { misc[4] = b; misc[5] = S; }
            S -= c[j];
        }
    }
    __syncthreads();
    const unsigned int first = misc[4], m = misc[5];
    *m_out = m;
    if (first >= NB || m == 0u) { __syncthreads(); return 0u; }          // the nearest bin alone overflows the cap
    {
        const unsigned int skipped = n - m;        // keys in the bins in front of the selection
        unsigned int ex = before, big = 0u;
#pragma unroll
        for (unsigned int j = 0; j < 4; ++j) {
            const unsigned int b = 4u * tid + j;
            if (b >= first) {
                const unsigned int at = ex - skipped;
                bins[b] = at;                               // the bin's place in the output
                if (c[j]) {                                 // (ordinary lists) a part begins at the first bin placed in its interval of PART_T places
                    const unsigned int p = at / PART_T;
                    if (p < PMAX) atomicMin(&pstart[p], at);
                    atomicMax(&misc[3], p);
                    big = max(big, c[j]);
                }
            }
            ex += c[j];
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) big = max(big, (unsigned int)__shfl_xor((int)big, o));
        if (lane == 0u) atomicMax(&misc[2], big);
    }
    __syncthreads();
    // The parts.  No selected bin above BIN_MAX keys (the rule): every interval of PART_T = 1280 output places holds a bin's
    // start, the first of them begins a part, parts stay under 1280 + 512 keys.  Otherwise -- many Gaussians at one depth:
    // a wall seen face on -- one thread walks the selected bins: consecutive bins are grouped while the group stays within
    // PART_MAX keys, a bin above BIN_MAX is a part of its own (up to BIG_PART_MAX keys, else the whole list goes the global
    // route).
    if (tid == 0u) {
        unsigned int np = 0u;
        bool ok = true;
        if (misc[2] <= BIN_MAX) {
            np = misc[3] + 1u;
            ok = np < PMAX;
        } else {
            unsigned int part_beg = 0u, size = 0u;
            bool prev_big = false;
            for (unsigned int b = first; b < NB && ok; ++b) {
                const unsigned int at = bins[b], cnt = ((b + 1u < NB) ? bins[b + 1u] : m) - at;
                if (cnt == 0u) continue;
                const bool bigbin = cnt > BIN_MAX;
                // (a part of its own up to the size the parts of ordinary lists reach.  Larger ones -- up to the 2016 keys the
                // workspace would hold beside the part table -- were blamed for wrong tiles on hostile scenes once; on the final
                // build they pass the same 2500 scenes, and the failures look like the allocation-time fill race found later
                // (DESIGN.md section 1).  The limit costs nothing measurable and stays.)
                if (cnt > BIG_PART_MAX) { ok = false; break; }       // (2016 keys + the sort's histograms + the part table: the workspace)
                if (size != 0u && (bigbin || prev_big || size + cnt > PART_MAX)) {
                    if (np >= PMAX - 1u) { ok = false; break; }
                    pstart[np++] = part_beg;
                    part_beg = at; size = 0u;
                }
                size += cnt; prev_big = bigbin;
            }
            if (ok && size != 0u) pstart[np++] = part_beg;
        }
        if (ok) pstart[np] = m;
        misc[3] = ok ? np : 0u;
    }
    __syncthreads();
    const unsigned int P = misc[3];
    if (P == 0u) { __syncthreads(); return 0u; }
    for (unsigned int t0 = tid; t0 < n; t0 += 8u * NT) {
        unsigned long long k[8];
#pragma unroll
        for (unsigned int u = 0; u < 8; ++u) { const unsigned int t = t0 + u * NT; k[u] = (t < n) ? g[t] : 0ull; }
#pragma unroll
        for (unsigned int u = 0; u < 8; ++u) {
            const unsigned int b = bin_of((unsigned int)(k[u] >> 32));
            if (t0 + u * NT < n && b >= first) h[bins[b] + atomicAdd(&cur[b], 1u)] = k[u];
        }
    }
    __syncthreads();                                   // h is complete (this workgroup's own global writes) and bins / cur are free
    return P;
}

// The NEAREST keys of a long list, selected without sorting it.  The compositor's exact walk starts where the
// transmittance of every pixel of a block has fallen below ~1e-6 and needs nothing behind that point (the skipped layers
// are bracketed, composite_tile): of the lists of more than 2048 keys -- two thirds of all keys on C3, five sixths on C5 --
// the deepest of a tile's four walks reaches 2-9 % (SPLAT_DBG_STARTS, tools/starts_probe.py).  Sorting the other 90 % is
// the single largest piece of wasted work in the frame.  So, by the tile's own workgroup (NT = 256 threads, its
// sort_lds_bytes<NT, 2048>() of LDS), in two streaming passes over the n keys at g (global, unordered):
//   1  a depth range for the bins, from a 256-key sample of the list
//   2  a histogram of the keys' depths over 1024 bins of that range (LDS atomics)
//   3  the largest SUFFIX of bins -- ascending depth key = far first, so the suffix is the near end -- that holds at most
//      `cap` <= 2048 keys; those keys are compacted into LDS (smem as the key array s[0 .. m)), in arrival order
// and returns m (with the min / max of the selected depth halves for the digit plan of the sort that follows).  A bin
// holds every key of its depths, so the selection is EXACTLY the last m entries of the fully sorted list, ties in depth
// included.  Returns 0 if the nearest bin alone holds more than `cap` keys (hundreds of Gaussians at one depth): the
// caller sorts the whole list instead.  Ends with a barrier: s[0 .. m) is complete.
template <int NT>
__device__ __forceinline__ unsigned int select_near(unsigned char* smem, const unsigned long long* __restrict__ g, unsigned int n,
                                                    unsigned int cap, unsigned int* sel_mn, unsigned int* sel_mx, unsigned int* sel_thr) {
    constexpr unsigned int NB = 1024u;
    static_assert(NT == 256, "four bins per thread");
    static_assert(2048u * 8u + NB * 4u + 64u <= sort_lds_bytes<NT, 2048>(), "keys + bins + a few words fit the workspace");
    unsigned long long* s = reinterpret_cast<unsigned long long*>(smem);
    unsigned int* bins = reinterpret_cast<unsigned int*>(smem + 2048u * 8u);
    unsigned int* misc = bins + NB;                 // 0 min, 1 max, 2 first selected bin, 3 selected keys, 4 compaction cursor, 8.. wave sums
    const unsigned int tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    for (unsigned int q = tid; q < NB; q += NT) bins[q] = 0u;
    if (tid == 0u) { misc[0] = 0xffffffffu; misc[1] = 0u; misc[2] = NB; misc[3] = 0u; misc[4] = 0u; }
    // The bins' depth range comes from a SAMPLE of the list -- one key per thread, strided over the list (its order is
    // K1's arrival order: arbitrary) -- widened by a quarter on either side; keys outside fall into the end bins.  Any
    // monotone binning selects correctly; a range that misses only costs resolution there (the true extremes would take a
    // pass over the whole list: a third of this function's memory traffic).
    unsigned int mn, mx;
    {
        const unsigned int d = (unsigned int)(g[(unsigned int)(((unsigned long long)tid * n) / NT)] >> 32);
        mn = d; mx = d;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        mn = min(mn, (unsigned int)__shfl_xor((int)mn, o));
        mx = max(mx, (unsigned int)__shfl_xor((int)mx, o));
    }
    __syncthreads();
    if (lane == 0u) { atomicMin(&misc[0], mn); atomicMax(&misc[1], mx); }
    __syncthreads();
    const unsigned int smin = misc[0], smax = misc[1], quarter = (smax - smin) >> 2;
    const unsigned int lo = smin > quarter ? smin - quarter : 0u;
    const unsigned int hi = smax < 0xffffffffu - quarter ? smax + quarter : 0xffffffffu;
    const unsigned int range = hi - lo;
    const unsigned int sh = range >= NB ? (unsigned int)(32 - __clz((int)range)) - 10u : 0u;      // (range >> sh) < 1024
    auto bin_of = [&](unsigned int d) -> unsigned int { return d < lo ? 0u : min((d - lo) >> sh, NB - 1u); };
    for (unsigned int t0 = tid; t0 < n; t0 += 8u * NT) {              // eight loads in flight per thread
        unsigned int d[8];
#pragma unroll
        for (unsigned int u = 0; u < 8; ++u) { const unsigned int t = t0 + u * NT; d[u] = (t < n) ? (unsigned int)(g[t] >> 32) : 0u; }
#pragma unroll
        for (unsigned int u = 0; u < 8; ++u) if (t0 + u * NT < n) atomicAdd(&bins[bin_of(d[u])], 1u);
    }
    __syncthreads();
    {   // suffix sums, four consecutive bins per thread: thread t owns bins 4t .. 4t+3; S(b) = keys in bins >= b
        unsigned int c[4], sum = 0u;
#pragma unroll
        for (unsigned int j = 0; j < 4; ++j) { c[j] = bins[4u * tid + j]; sum += c[j]; }
        unsigned int v = sum;                       // inclusive scan over the wave's lanes, then the waves in front of this one
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned int u = (unsigned int)__shfl_up((int)v, o);
            if ((int)lane >= o) v += u;
        }
        if (lane == 63u) misc[8u + wave] = v;
        __syncthreads();
        unsigned int total = 0u, before = v - sum;  // keys in the bins in front of this thread's
        for (unsigned int w = 0; w < NT / 64u; ++w) { const unsigned int x = misc[8u + w]; total += x; if (w < wave) before += x; }
        // the first selected bin is the smallest b with S(b) <= cap: S(b) <= cap and (b == 0 or S(b - 1) = S(b) + count(b - 1) > cap)
        unsigned int S = total - before;
#pragma unroll
        for (unsigned int j = 0; j < 4; ++j) {
            const unsigned int b = 4u * tid + j;
            const unsigned int cnt_prev = (j == 0u) ? ((tid == 0u) ? 0u : bins[b - 1u]) : c[j - 1u];
            if (S <= cap && (b == 0u || S + cnt_prev > cap)) { misc[2] = b; misc[3] = S; }
            S -= c[j];
        }
    }
    __syncthreads();
    const unsigned int first = misc[2], m = misc[3];
    if (first >= NB || m == 0u) { __syncthreads(); return 0u; }          // the nearest bin alone overflows the cap
    {   // the selection is exactly the keys of depth >= this (0: no such statement -- bin 0 also holds what lies in front of the range)
        const unsigned long long t = (unsigned long long)lo + ((unsigned long long)first << sh);
        *sel_thr = (first == 0u || t > 0xffffffffull) ? 0u : (unsigned int)t;
    }
    __syncthreads();                                                     // (misc[0..1] are reused for the selection's extremes)
    if (tid == 0u) { misc[0] = 0xffffffffu; misc[1] = 0u; }
    mn = 0xffffffffu; mx = 0u;
    for (unsigned int t0 = tid; t0 < n; t0 += 8u * NT) {
        unsigned long long k[8];
#pragma unroll
        for (unsigned int u = 0; u < 8; ++u) { const unsigned int t = t0 + u * NT; k[u] = (t < n) ? g[t] : 0ull; }
#pragma unroll
        for (unsigned int u = 0; u < 8; ++u) {
            const unsigned int d = (unsigned int)(k[u] >> 32);
            const bool take = (t0 + u * NT < n) && (bin_of(d) >= first);
            const unsigned long long tm = __builtin_amdgcn_ballot_w64(take);
            if (tm) {                               // one cursor atomic per wave and round
                unsigned int base = 0u;
                if (lane == 0u) base = atomicAdd(&misc[4], (unsigned int)__builtin_popcountll(tm));
                base = (unsigned int)__builtin_amdgcn_readfirstlane((int)base);
                if (take) {
                    s[base + __builtin_amdgcn_mbcnt_hi((unsigned int)(tm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)tm, 0u))] = k[u];
                    mn = min(mn, d); mx = max(mx, d);
                }
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        mn = min(mn, (unsigned int)__shfl_xor((int)mn, o));
        mx = max(mx, (unsigned int)__shfl_xor((int)mx, o));
    }
    __syncthreads();                                // (tid 0's reset of misc[0..1] is in place)
    if (lane == 0u) { atomicMin(&misc[0], mn); atomicMax(&misc[1], mx); }
    __syncthreads();                                // s[0 .. m) and the extremes are complete
    *sel_mn = misc[0];                              // the depth range of the selection: sizes the digits of the sort that follows
    *sel_mx = misc[1];
    return m;
}

// The same selection in ONE pass over the list, when the depth it begins at is known beforehand (the tile's selection of
// the previous frame began there: select_near's *sel_thr): every key of depth >= thr is compacted into LDS, s[0 .. count).
// Returns the count -- which may be anything: more than the 2048 keys the workspace holds (only the first 2048 arrivals
// were stored: the caller takes the two passes after all), or too few to be worth walking.  Exactly the nearest `count`
// keys of the list whatever the threshold (every key at or beyond a depth is a suffix of the sorted list, ties included).
template <int NT>
__device__ __forceinline__ unsigned int select_by_depth(unsigned char* smem, const unsigned long long* __restrict__ g, unsigned int n,
                                                        unsigned int thr, unsigned int* sel_mn, unsigned int* sel_mx) {
    unsigned long long* s = reinterpret_cast<unsigned long long*>(smem);
    unsigned int* misc = reinterpret_cast<unsigned int*>(smem + 2048u * 8u);      // 0 min, 1 max, 2 cursor
    const unsigned int tid = threadIdx.x, lane = tid & 63u;
    if (tid == 0u) { misc[0] = 0xffffffffu; misc[1] = 0u; misc[2] = 0u; }
    __syncthreads();
    unsigned int mn = 0xffffffffu, mx = 0u;
    for (unsigned int t0 = tid; t0 < n; t0 += 8u * NT) {              // eight loads in flight per thread
        unsigned long long k[8];
#pragma unroll
        for (unsigned int u = 0; u < 8; ++u) { const unsigned int t = t0 + u * NT; k[u] = (t < n) ? g[t] : 0ull; }
#pragma unroll
        for (unsigned int u = 0; u < 8; ++u) {
            const unsigned int d = (unsigned int)(k[u] >> 32);
            const bool take = (t0 + u * NT < n) && d >= thr;
            const unsigned long long tm = __builtin_amdgcn_ballot_w64(take);
            if (tm) {                               // one cursor atomic per wave and round
                unsigned int base = 0u;
                if (lane == 0u) base = atomicAdd(&misc[2], (unsigned int)__builtin_popcountll(tm));
                base = (unsigned int)__builtin_amdgcn_readfirstlane((int)base);
                if (take) {
                    const unsigned int at = base + __builtin_amdgcn_mbcnt_hi((unsigned int)(tm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)tm, 0u));
                    if (at < 2048u) s[at] = k[u];
                    mn = min(mn, d); mx = max(mx, d);
                }
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        mn = min(mn, (unsigned int)__shfl_xor((int)mn, o));
        mx = max(mx, (unsigned int)__shfl_xor((int)mx, o));
    }
    if (lane == 0u) { atomicMin(&misc[0], mn); atomicMax(&misc[1], mx); }
    __syncthreads();                                // s[0 .. count) and the extremes are complete
    *sel_mn = misc[0];
    *sel_mx = misc[1];
    const unsigned int count = misc[2];
    __syncthreads();                                // (misc is the sort's from here on)
    return count;
}

// The whole of it, for the compositor's workgroup: the nearest min(cap, n) keys of the list at g (region of the key buffer,
// n keys, unordered) put in painter's order.  Returns m, how many that are: m == n -- the whole list, sorted in g (the
// global route of last resort included); m < n -- the selection, sorted in h[0 .. m) (h = the second key buffer's region),
// g untouched.  A selection that cannot be partitioned (a bin too full, too many parts) becomes the whole list.
// need_min: a selection of fewer keys than this is not worth having (the bins are coarse where many keys share a depth): the
// whole list instead.
__device__ __forceinline__ unsigned int sort_long_list(unsigned char* smem, unsigned long long* g, unsigned long long* h, unsigned int n,
                                                       unsigned int cap, unsigned int radix_min, FrameStatus* status, const unsigned int* orig,
                                                       unsigned int need_min = 0u) {
    unsigned int m = 0u, parts = 0u;
    for (int attempt = 0; attempt < 2; ++attempt) {        // (one inlined copy of the partition)
        parts = partition_long_list<256>(smem, g, h, n, cap, &m);
        if ((parts != 0u && (m >= need_min || m == n)) || cap >= n) break;
        cap = n; parts = 0u;
    }
    if (parts == 0u) {
        sort_list_global<256>(smem, g, h, n, status, orig);
        return n;
    }
    const unsigned int* const pstart = reinterpret_cast<const unsigned int*>(smem + sort_lds_bytes<256, 2048>() - LONG_SORT_RESERVE);
    unsigned long long* const dst = (m == n) ? g : h;
    for (unsigned int p = 0; p < parts; ++p) {
        const unsigned int a = (unsigned int)__builtin_amdgcn_readfirstlane((int)pstart[p]);
        const unsigned int b = (unsigned int)__builtin_amdgcn_readfirstlane((int)pstart[p + 1u]);
        sort_list_in_lds<256, 2048>(smem, h + a, dst + a, b - a, radix_min, status, orig, nullptr, true, LONG_SORT_RESERVE);
        __syncthreads();                               // the workspace is the next part's
    }
    return m;
}

// One workgroup per tile; a launch handles the lists with lo < n <= CAP (three size classes, so
// that mid-sized lists get two workgroups per CU instead of one LDS-filling one).
// chunks > 0 (the 1024-thread class): a list of up to chunks * CAP keys is sorted as that many
// CAP-sized runs, one workgroup each, which merge_runs_kernel then merges; only a list longer than
// that takes the slow route (radix passes through global memory by its first workgroup).
template <int NT, int CAP>
__global__ __launch_bounds__(NT) void sort_tiles_kernel(const unsigned int* __restrict__ offsets,
                                                         const unsigned int* __restrict__ order,
                                                         const unsigned int* __restrict__ lens,
                                                         unsigned long long* __restrict__ keys,
                                                         unsigned long long* __restrict__ keys2,
                                                         FrameStatus* __restrict__ status, unsigned int lo,
                                                         unsigned int radix_min, int chunks, unsigned int grid0,
                                                         unsigned int grid_long, const unsigned int* __restrict__ orig,
                                                         const unsigned int* __restrict__ off2) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    if (status->overflow) return;
    // workgroups [0, grid0) take run 0 of the first grid0 lists; the next 3 * grid_long ones take runs
    // 1..3 of the first grid_long lists (those >= 16384 keys are a prefix of the longest-first order)
    unsigned int oi = blockIdx.x, chunk = 0u;
    if (chunks && blockIdx.x >= grid0) { const unsigned int r = blockIdx.x - grid0; chunk = 1u + r / grid_long; oi = r % grid_long; }
    const unsigned int tile = order[oi];
    unsigned int b = offsets[tile];
    unsigned int n = lens[tile];
    if (n < 2 || n <= lo) return;
    const bool too_long = n > (unsigned int)CAP * (unsigned int)max(chunks, 1);
    if (chunks && n > (unsigned int)CAP && !too_long) {          // this workgroup's run of a long list
        if (chunk * (unsigned int)CAP >= n) return;
        b += chunk * (unsigned int)CAP;
        n = min((unsigned int)CAP, n - chunk * (unsigned int)CAP);
        if (n < 2) return;
    } else if (chunk != 0) {
        return;
    }
    if (n <= (unsigned int)CAP) {
        sort_list_in_lds<NT, CAP>(smem, keys + b, keys + b, n, radix_min, status, orig);
    } else if (chunks) {
        // longer than chunks * CAP: radix passes over the L2-resident bucket, then the same tie fix-up
        sort_list_global<NT>(smem + (size_t)CAP * 8, keys + b, keys2 + off2[tile], n, status, orig);
    }
}

// Merge path: how many of the first d outputs of merge(A[0..la), B[0..lb)) come from A.  Keys are
// unique (the slot half) and key_gt is a strict total order, so there are no ties.
template <typename PA, typename PB>
__device__ __forceinline__ unsigned int merge_path(unsigned int d, unsigned int la, unsigned int lb, PA A, PB B,
                                                   const unsigned int* __restrict__ orig) {
    unsigned int lo = d > lb ? d - lb : 0u, hi = min(d, la);
    while (lo < hi) {
        const unsigned int mid = (lo + hi) >> 1;
        if (key_gt(B[d - 1u - mid], A[mid], orig)) lo = mid + 1u; else hi = mid;
    }
    return lo;
}

// K3b -- lists of CAP < n <= 4 CAP keys, which sort_tiles_kernel left as up to four sorted runs of
// CAP keys: two levels of pairwise merging, keys -> keys2 -> keys (an absent run is an empty one,
// so a two-run list simply gets copied back by the second level).  One 1024-thread workgroup per
// list; each level walks its output in tiles of MERGE_T keys: the tile's inputs (found with one
// merge-path search per tile boundary, all boundaries searched at once) are staged in LDS, every
// thread merge-path-searches its eight outputs there and writes them.
constexpr unsigned int MERGE_T = 8192;
template <int NT, int CAP>
__global__ __launch_bounds__(NT) void merge_runs_kernel(const unsigned int* __restrict__ offsets,
                                                         const unsigned int* __restrict__ order,
                                                         const unsigned int* __restrict__ lens,
                                                         unsigned long long* __restrict__ keys,
                                                         unsigned long long* __restrict__ keys2,
                                                         const FrameStatus* __restrict__ status,
                                                         const unsigned int* __restrict__ orig, const unsigned int* __restrict__ off2) {
    __shared__ unsigned long long sm[MERGE_T];
    __shared__ unsigned int split[4 * CAP / MERGE_T + 2];
    if (status->overflow) return;
    const unsigned int tile = order[blockIdx.x];
    const unsigned int n = lens[tile];
    if (n <= (unsigned int)CAP || n > 4u * (unsigned int)CAP) return;
    const unsigned int tid = threadIdx.x;
    const size_t base = offsets[tile];
    auto merge = [&](const unsigned long long* A, unsigned int la, const unsigned long long* B, unsigned int lb,
                     unsigned long long* dst) {
        const unsigned int total = la + lb, ntile = (total + MERGE_T - 1u) / MERGE_T;
        if (tid <= ntile) split[tid] = merge_path(min(tid * MERGE_T, total), la, lb, A, B, orig);
        __syncthreads();
        for (unsigned int k = 0; k < ntile; ++k) {
            const unsigned int k0 = k * MERGE_T, k1 = min(k0 + MERGE_T, total);
            const unsigned int i0 = split[k], i1 = split[k + 1], j0 = k0 - i0, na = i1 - i0, nb = (k1 - k0) - na;
            for (unsigned int t = tid; t < na; t += NT) sm[t] = A[i0 + t];
            for (unsigned int t = tid; t < nb; t += NT) sm[na + t] = B[j0 + t];
            __syncthreads();
            constexpr unsigned int E = MERGE_T / NT;
            const unsigned int q0 = tid * E;
            if (q0 < k1 - k0) {
                const unsigned long long* sA = sm;
                const unsigned long long* sB = sm + na;
                unsigned int ia = merge_path(q0, na, nb, sA, sB, orig), ib = q0 - ia;
#pragma unroll
                for (unsigned int e = 0; e < E; ++e) {
                    if (q0 + e < k1 - k0) {
                        const bool take_a = (ib >= nb) || (ia < na && key_gt(sB[ib], sA[ia], orig));
                        dst[k0 + q0 + e] = take_a ? sA[ia] : sB[ib];
                        ia += take_a ? 1u : 0u; ib += take_a ? 0u : 1u;
                    }
                }
            }
            __syncthreads();
        }
    };
    const unsigned int r1 = min(n, (unsigned int)CAP), r2 = min(n, 2u * CAP), r3 = min(n, 3u * CAP);
    unsigned long long* g = keys + base;
    unsigned long long* h = keys2 + off2[tile];
    merge(g, r1, g + r1, r2 - r1, h);                 // runs 0, 1 -> h[0 .. r2)
    merge(g + r2, r3 - r2, g + r3, n - r3, h + r2);   // runs 2, 3 -> h[r2 .. n)
    __syncthreads();                                  // (orders this workgroup's global writes and reads)
    merge(h, r2, h + r2, n - r2, g);                  // -> the list, sorted, back in place
}

// Does ANY sample s = lo + k (k = 0..count-1, all exactly representable) satisfy |s - c| <= h ?
// |s - c| grows monotonically (also after f32 rounding) away from c, so testing the one or two
// samples nearest to c is exact.
__device__ __forceinline__ bool any_sample_covered(float c, float h, float lo, float hi, float off) {
    float s1 = fminf(fmaxf(floorf(c - off) + off, lo), hi);
    float s2 = fminf(s1 + 1.0f, hi);
    return (int)(fabsf(s1 - c) <= h) | (int)(fabsf(s2 - c) <= h);
}

__device__ __forceinline__ unsigned long long uniform_u64(unsigned long long v) {
    // (readfirstlane returns int: go through unsigned, or the low word sign-extends)
    return ((unsigned long long)(unsigned int)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) |
           (unsigned long long)(unsigned int)__builtin_amdgcn_readfirstlane((int)(unsigned int)v);
}

// exp(x) for the compositor: the same reduction ocml's expf performs (2^(x*log2e) with a
// compensated product, v_exp_f32 on the fractional part, ldexp) without its overflow/underflow
// selects -- x is a Gaussian exponent, <= 0 and far above -100 wherever the result is used.
__device__ __forceinline__ float exp_neg(float x) {
    // x * log2(e) as an unevaluated sum ph + pl (compensated product); v_exp_f32 takes the rounded part
    // whole -- it does its own range reduction and x is a Gaussian exponent (<= 0, far above -100
    // wherever the result is used), so ocml's integer / fraction split, its ldexp and its range
    // selects are not needed -- and the residual enters to first order: 2^(ph+pl) = 2^ph (1 + pl ln 2).
    const float L2E_HI = __uint_as_float(0x3fb8aa3bu), L2E_LO = __uint_as_float(0x32a5705fu);
    const float LN2 = 0.6931471805599453f;
    float ph = x * L2E_HI;
    float pl = fmaf(x, L2E_HI, -ph);
    pl = fmaf(x, L2E_LO, pl);
    const float e = __builtin_amdgcn_exp2f(ph);
    return fmaf(e * pl, LN2, e);
}

// expf as glibc computes it (sysdeps/ieee754/flt-32/e_expf.c since 2.27: the ARM optimized-routines algorithm,
// restated from its published description): x N/ln2 = k + r with N = 32, 2^(k/N) from a 32-entry table of doubles,
// 2^(r/N) as a cubic, all in double, one rounding to float at the end.  Bit-identical to the host libm's expf --
// which is what the oracle (and, through Rust's f32::exp, the reference on a glibc host) calls -- on every input
// that can reach it here (checked against libm on random arguments by tests/test_host.py through the same
// constants).  SPLAT_MODE_LIBM_EXP selects it: the frame is then the oracle's frame BIT FOR BIT, which shows that
// the exponential's last place is the only thing the default build rounds differently.  It costs ~18 double
// instructions per fragment, so it is a verification mode, not the default.
__constant__ const unsigned long long EXP2F_TAB[32] = {      // tab[i] = bits(2^(i/32)) - (i << 47)
    0x3ff0000000000000ull, 0x3fefd9b0d3158574ull, 0x3fefb5586cf9890full, 0x3fef9301d0125b51ull, 0x3fef72b83c7d517bull,
    0x3fef54873168b9aaull, 0x3fef387a6e756238ull, 0x3fef1e9df51fdee1ull, 0x3fef06fe0a31b715ull, 0x3feef1a7373aa9cbull,
    0x3feedea64c123422ull, 0x3feece086061892dull, 0x3feebfdad5362a27ull, 0x3feeb42b569d4f82ull, 0x3feeab07dd485429ull,
    0x3feea47eb03a5585ull, 0x3feea09e667f3bcdull, 0x3fee9f75e8ec5f74ull, 0x3feea11473eb0187ull, 0x3feea589994cce13ull,
    0x3feeace5422aa0dbull, 0x3feeb737b0cdc5e5ull, 0x3feec49182a3f090ull, 0x3feed503b23e255dull, 0x3feee89f995ad3adull,
    0x3feeff76f2fb5e47ull, 0x3fef199bdd85529cull, 0x3fef3720dcef9069ull, 0x3fef5818dcfba487ull, 0x3fef7c97337b9b5full,
    0x3fefa4afa2a490daull, 0x3fefd0765b6e4540ull};
__device__ __forceinline__ float exp_libm(float x, const unsigned long long* __restrict__ tab /* LDS copy of EXP2F_TAB */) {
    const double InvLn2N = 0x1.71547652b82fep+0 * 32.0, SHIFT = 0x1.8p+52;
    const double C0 = 0x1.c6af84b912394p-5 / 32.0 / 32.0 / 32.0, C1 = 0x1.ebfce50fac4f3p-3 / 32.0 / 32.0, C2 = 0x1.62e42ff0c52d6p-1 / 32.0;
    // below -87 expf is < 2e-38 and the fragment is rejected whatever its opacity; the clamp keeps k in the table's
    // range without libm's underflow branches.  NaN stays NaN.
    const double xd = (double)((x != x) ? x : fmaxf(x, -87.0f));
    const double z = InvLn2N * xd;
    double kd = z + SHIFT;                                     // round to nearest integer, in the low mantissa bits
    const unsigned long long ki = (unsigned long long)__double_as_longlong(kd);
    kd -= SHIFT;
    const double r = z - kd;
    const double sc = __longlong_as_double((long long)(tab[ki & 31ull] + (ki << 47)));
    const double zz = C0 * r + C1;
    const double r2 = r * r;
    double y = C2 * r + 1.0;
    y = zz * r2 + y;
    y = y * sc;
    return (float)y;
}

// k / 255.0f (IEEE) for every integer k in [0,255] in two instructions: 1/255 split into
// hi + lo floats, fma(k, hi, k*lo) rounds once (checked exhaustively in tests/test_host.py).
__device__ __forceinline__ float div255(float k) {
    const float RH = 0x1.010102p-8f, RL = -0x1.fdfdfep-33f;
    return fmaf(k, RH, k * RL);
}
// One channel of blend(): src/pipelines.rs:157-161.  Monotone non-decreasing in the state k for
// fixed alpha/colour (every step -- /255, *ia, +const, *255, clamp, trunc -- is monotone under
// round-to-nearest), which is what makes the [lo,hi] bracket of the early-out exact.
// The u8 cast saturates (NaN/negative -> 0, >= 255 -> 255).  Clamping the blended value to [0,1]
// BEFORE the *255 gives the same byte for every input (x in [0,1] is untouched; x > 1 -> 255; x < 0 or
// NaN -> 0) and folds into the add as its clamp modifier: one VALU less per channel than med3.
__device__ __forceinline__ float blend_channel(float k, float ia, float ac) {
    const float x = __builtin_amdgcn_fmed3f(ia * div255(k) + ac, 0.0f, 1.0f);
    return truncf(x * 255.0f);
}

struct WaveLds { float4 a[64]; float4 b[64]; float4 c[64]; };   // one batch of 64 records, private to a wave

// ---- the exact walk two records at a time (composite_exact_kernel<true>) -----------------------------------
// fragment() of two consecutive records is evaluated with packed f32 instructions (v_pk_add/mul/fma_f32: the two
// records side by side in even-aligned register pairs), then blend() runs for the first and for the second, red
// and green side by side.  Every component goes through the same IEEE operations in the same order as the
// one-record loop, so the pixels are the same bits.  A packed instruction occupies the SIMD for two issue
// slots (tools/valu_probe.hip), so this buys little where the compositor is throughput-bound; it is for the
// waves that are alone on their SIMD -- the densest tile of a multi-GPU slab, the tail of every launch, small
// frames -- which issue one instruction per ~5 cycles whatever its kind: a third fewer issue slots per record.
// Measured (same box, frames byte-identical): C2 4291 -> 5126 frames/s, C3 2243 -> 2116 (its compositor is
// throughput-bound, and the paired loop costs a workgroup of occupancy: 74 VGPRs).  The host picks the kernel
// per frame from the previous frame's statistics (pairs per key of the longest list; SPLAT_PAIR_BLEND=0/1 forces).
// The batch is staged pair-interleaved, so that the operands are born in aligned register pairs (no moves):
//   pair p (records 2p, 2p+1) = 24 floats:  cx0 cx1 cy0 cy1 | hx0 hx1 hy0 hy1 | A0 A1 C0 C1 | B0 B1 op0 op1 |
//                                            r0 g0 r1 g1 | b0 - b1 -
typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f2 pk_add_clamp(f2 a, f2 b) {      // clamp(a + b, 0, 1) per component: the add's clamp modifier
    f2 r;                                                      // (NaN -> 0 like the one-record loop's v_add_f32 ... clamp)
    asm("v_pk_add_f32 %0, %1, %2 clamp" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// blend() for two channels side by side: see blend_channel
__device__ __forceinline__ f2 blend_channel2(f2 k, float ia, f2 ac) {
    const float RH = 0x1.010102p-8f, RL = -0x1.fdfdfep-33f;
    const f2 f = __builtin_elementwise_fma(k, (f2)(RH), k * RL);       // div255, both channels
    const f2 y = pk_add_clamp(ia * f, ac) * 255.0f;
    return (f2){truncf(y.x), truncf(y.y)};
}

// K4 -- compositor.  One wave = one 8x8 pixel block of a 16x16 tile (4 waves per workgroup, but
// they never synchronise: each wave streams the tile's list through its own 3 KB of LDS, 64
// records at a time, fetching the next batch into registers while it walks the current one).
// Staging compacts a batch to the records that cover a sample of the block (ballot + mbcnt); the
// wave walks them in a counted loop, one uniform LDS broadcast per record.
//
// EARLY-OUT, exactness preserved:
//   phase A  walks the list near -> far with a cheap approximate alpha until every pixel of the
//            block has transmittance < eps; the farthest such layer is the wave's start.
//   phase B  composites far -> near from that start.  Skipped layers are bracketed: the state
//            is carried twice, from 0 and from 255; blend() is monotone in the state, so once
//            lo == hi the result provably does not depend on anything skipped and the walk
//            continues single-state.  A wave that ends with lo != hi redoes the whole list.
#ifndef SPLAT_COMP_WAVES
#define SPLAT_COMP_WAVES 1
#endif
// The compositor kernel's arguments, as ONE struct: the kernarg segment then is this struct, and the rarely taken repair path
// (repair_tile, a real function call) re-reads them from there instead of having them handed over in registers.
struct CompArgs {
    FrameConst fc;
    const unsigned int* offsets; const unsigned int* order; const unsigned int* lens;
    unsigned long long* keys; const Rec* recs; uint32_t* argb; FrameStatus* status;
    unsigned int fused_sort_max, radix_min;
    uint2* iters;
    unsigned int keep_keys, clear_first;
    const unsigned int* orig;
    unsigned long long* keys2; const unsigned int* near_m;
    unsigned int* need_hint; unsigned int* start_hint; const unsigned int* off2;
};
typedef const __attribute__((address_space(4))) CompArgs* KernArgs;       // (constant address space: uniform scalar loads)
template <bool PAIR, bool LIBM>
static __device__ void repair_tile(KernArgs ka, unsigned int smem_lds, unsigned int exptab_lds, unsigned int item, bool again);

// One tile (slot `item` of the longest-first tile order) by one workgroup; `smem` = sort_lds_bytes<256, 2048>() bytes.
// PAIR: the exact walk takes two records per step with packed math (see the note above WaveLds).
// LONG: lists of more than 2048 keys are sorted by this workgroup as well (sort_long_list), no sort launch needed.
// LONGM: who orders a list of more than 2048 keys when no sort launch did --
//   0  nobody here (the sort launches ran)
//   1  this workgroup, in full (sort_long_list)
//   2  NEAR SELECTION: this workgroup selects and sorts the nearest <= near_cap keys only; a wave whose walk needs more
//      makes its tile repair itself: the workgroup sorts the whole list (flavour 1's code) and the waves that asked walk again
template <bool PAIR, bool LIBM, int LONGM>
__device__ __forceinline__ void composite_tile(unsigned char* smem, const unsigned long long* exptab, const unsigned int item, const FrameConst& fc,
                                               const unsigned int* __restrict__ offsets, const unsigned int* __restrict__ order,
                                               const unsigned int* __restrict__ lens, unsigned long long* __restrict__ keys,
                                               const Rec* __restrict__ recs, uint32_t* __restrict__ argb,
                                               FrameStatus* __restrict__ status, unsigned int fused_sort_max,
                                               unsigned int radix_min, uint2* __restrict__ iters, unsigned int keep_keys,
                                               const unsigned int* __restrict__ orig, const unsigned int clear_first,
                                               unsigned long long* __restrict__ keys2, const unsigned int* __restrict__ near_m,
                                               unsigned int* __restrict__ need_hint, unsigned int* __restrict__ start_hint,
                                               const unsigned int* __restrict__ off2, KernArgs ka /* the kernel's arguments where they lie: see late() */,
                                               const bool walk_this_wave = true, const bool second_walk = false /* flavour 1 inside repair_tile */) {
    // LATE ARGUMENTS.  What a wave needs once, at the end of its walk (the image, the hint tables, the statistics) or on a path
    // few tiles take (the empty tile's clear, the retry counter) is read from the kernarg segment THERE, through a pointer the
    // compiler cannot see through -- otherwise every such argument is loaded at the kernel's entry and carried through the
    // walks in scalar registers the hot loops need: the near-selection flavour then wants 106 of the 96 a seven-workgroup CU
    // allows, spills twenty into the lanes of a vector register, and that register into scratch
    // (round 5's build: one scratch_store at entry, six scratch_loads in front of the walks).  tests/test_codeobj.py holds
    // the kernel to no spill and no scratch instruction of its own.
    auto late = [&]() -> KernArgs {
        unsigned long long p = (unsigned long long)(size_t)ka;
        asm volatile("" : "+s"(p));
        return (KernArgs)(size_t)p;
    };
    // One LDS block, two lives: the workspace of the workgroup's own list sort (lists of up to
    // fused_sort_max <= 2048 keys are sorted here, by all four waves, instead of in a sort launch of
    // their own -- the short lists are most of the tiles, and their sort then runs beside the next
    // frame's K1 like the rest of this kernel instead of in the phase where the chip idles), then the
    // four waves' private record batches.
    static_assert(sizeof(WaveLds) * 4 <= sort_lds_bytes<256, 2048>(), "staging fits the sort workspace");
    WaveLds* slds = reinterpret_cast<WaveLds*>(smem);
    const unsigned int tile = (unsigned int)__builtin_amdgcn_readfirstlane((int)order[item]);
    const unsigned int tid = threadIdx.x;
    const unsigned int beg = __builtin_amdgcn_readfirstlane(offsets[tile]);
    const unsigned int end = beg + __builtin_amdgcn_readfirstlane(lens[tile]);
    if (beg == end) {
        // nothing covers this tile.  A frame that starts from a cleared image (clear_first: the viewer loop's
        // color.clear(0), src/main.rs:73, fused into this kernel) still owes the tile its zeros.
        if (clear_first) {
            const KernArgs k = late();
            const int tiles_x = k->fc.tiles_x, W = k->fc.W, H = k->fc.H, r0 = k->fc.row_px0, r1 = k->fc.row_px1;
            const int txx0 = (int)(tile % (unsigned int)tiles_x), tyy0 = (int)(tile / (unsigned int)tiles_x) + k->fc.tile_row0;
            const int px0 = txx0 * TILE + (int)(tid & 15u), py0 = tyy0 * TILE + (int)(tid >> 4);
            if (px0 < W && py0 < H && py0 >= r0 && py0 < r1) k->argb[(size_t)py0 * W + px0] = 0u;
        }
        return;
    }
    // A list this workgroup sorts itself never travels back to memory: the order (2048 x 4 B) stays in LDS behind
    // the waves' record batches, and the walks below read their indices from there instead of from the bucket.
    unsigned int* const lds_idx = reinterpret_cast<unsigned int*>(smem + sizeof(WaveLds) * 4);
    static_assert(sizeof(WaveLds) * 4 + 2048 * 4 <= sort_lds_bytes<256, 2048>(), "the order fits behind the batches");
    // ... and behind the order, 32 x 8 B per wave: which records of a 64-record batch the transmittance scan staged
    // (see `stage`), so that the exact walk over the same batches does not test them again.
    static_assert(sizeof(WaveLds) * 4 + 2048 * 4 + 4 * 32 * 8 <= sort_lds_bytes<256, 2048>(), "the batch masks fit too");
    unsigned long long* const wmask = reinterpret_cast<unsigned long long*>(smem + sizeof(WaveLds) * 4 + 2048 * 4) + (tid >> 6) * 32u;
    bool own_order = end - beg >= 2u && end - beg <= fused_sort_max;
    // A list of more than 2048 keys, when no sort launch ran in front of this kernel, is this workgroup's to order.
    // LONGM == 1: in full -- partitioned by depth into parts of fewer than 1792 keys through the second key buffer
    // (partition_long_list), every part sorted through LDS back into the region, then read from memory like a list a sort
    // launch had left.
    // LONGM == 2, NEAR SELECTION: a kernel in front of this one (select_near_kernel, on the frame's binning stream) has put
    // only the NEAREST keys of the list in order; the walks then see the list [lb, end) with `has_far` set -- farther keys
    // exist in front of `lb`, NOT in order.  The early-out makes that enough: a walk that starts inside the selection
    // and closes its bracket there is exact as always.  A wave whose walk would have to start at or before `lb` (its
    // pixels did not saturate within the selection, the bracket did not close, a pixel met no record) writes no pixel: the
    // workgroup then sorts the list in full and that wave walks again (the end of this function).
    // On C3 / C5 the selection serves every tile of the bench pose (tools/near_probe.py): 93-98 % of the long lists' keys
    // are never sorted.
    unsigned int lb = beg;
    bool has_far = false;
    const unsigned long long* wk = keys;          // the walks' key of list position p (lb <= p < end) is wk[p] (unless the order is in LDS)
    {
        const unsigned long long* gin = keys + beg;
        unsigned int n_sort = end - beg;
        if constexpr (LONGM == 1) {
            if (end - beg > 2048u) {
                (void)sort_long_list(smem, keys + beg, keys2 + off2[tile], end - beg, end - beg, radix_min, status, orig);
                __syncthreads();
            }
        }
        if constexpr (LONGM == 2) {
            if (end - beg > 2048u) {
                // select_near_kernel has been here: near_m[tile] of the list's nearest keys are in order -- all of them (in
                // the region itself), or a selection (at the start of the second key buffer's region)
                const unsigned int m = (unsigned int)__builtin_amdgcn_readfirstlane((int)near_m[tile]);
                if (m < end - beg) {
                    has_far = true; lb = end - m;
                    wk = keys2 + off2[tile] - lb;      // (the selection, in order, at the start of the tile's room in the second key buffer)
                }
            }
        }
        if (own_order) {
#if SPLAT_EXP_SORT2
            // timing experiment: the short lists' sort run twice (same frames): what it costs the compositor
            sort_list_in_lds<256, 2048>(smem, gin, keys + beg, n_sort, radix_min, status, orig, lds_idx, false);
            __syncthreads();
#endif
            sort_list_in_lds<256, 2048>(smem, gin, keys + beg, n_sort, radix_min, status, orig, lds_idx, (keep_keys & 1u) != 0u);
            __syncthreads();          // the order is in LDS, the rest of the workspace is free for the batches
        }
    }
    const unsigned int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63u;
    // The longest lists are the critical path of the launch: let their waves win VALU arbitration
    // against the short-list waves that share their SIMD (priority 0..3 by list length).
    {
        const unsigned int len = end - beg;
        if (len >= (unsigned int)fc.prio_len * 4u) __builtin_amdgcn_s_setprio(3);
        else if (len >= (unsigned int)fc.prio_len * 2u) __builtin_amdgcn_s_setprio(2);
        else if (len >= (unsigned int)fc.prio_len) __builtin_amdgcn_s_setprio(1);
    }
    // Everything this wave does with the list as it stands -- alpha pass, scan, exact walk, its pixels, its hints.  Returns
    // true when the walk needs keys in FRONT of a near selection (has_far): nothing has been written then.  (The wave's
    // own constants are derived inside: a tile that repairs itself sorts its list between two calls, and nothing of the
    // walk should have to live in registers across that sort.)
    if (!walk_this_wave) return;                 // (repair_tile: this wave's pixels are final; it was here for the sort)
    auto walk_list = [&]() -> bool {
    WaveLds& L = slds[wave];
    const int txx = (int)(tile % (unsigned int)fc.tiles_x), tyy = (int)(tile / (unsigned int)fc.tiles_x) + fc.tile_row0;
    // wave w owns the 8x8 pixel block (w&1, w>>1) of the tile: a square block meets fewer of the
    // tile's records than a 16x4 strip (measured: -8 % compositor time)
    const int bx0 = txx * TILE + 8 * (int)(wave & 1u), by0 = tyy * TILE + 8 * (int)(wave >> 1);
    const int px = bx0 + (int)(lane & 7u), py = by0 + (int)(lane >> 3);
    const bool inside = px < fc.W && py < fc.H && py >= fc.row_px0 && py < fc.row_px1;
    const float off = fc.sample_half ? 0.5f : 0.0f;
    // sample extents of this wave's strip, clipped to the target / slab
    const int x1w = min(bx0 + 7, fc.W - 1);
    const int y0w = by0, y1w = min(y0w + 7, min(fc.H, fc.row_px1) - 1);
    if (bx0 > x1w || y0w > y1w) return false;     // block entirely off the target: nothing to do
    const float xlo = (float)bx0 + off, xhi = (float)x1w + off;
    const float ylo = (float)y0w + off, yhi = (float)y1w + off;
    const float sx = (float)px + off, sy = (float)py + off;
    const uint32_t old = (inside && !clear_first) ? argb[(size_t)py * fc.W + px] : 0u;

    // Pixels outside the target never pass the coverage test: NaN sample coordinates.
    const float sxm = inside ? sx : __uint_as_float(0x7fc00000u), sym = inside ? sy : __uint_as_float(0x7fc00000u);

    // fetch: lane l pulls record (base + l) into registers.
    // stage: registers -> this wave's LDS, COMPACTED to the records that cover a sample of the strip
    //        (order kept); returns how many.  The walkers are then plain counted loops -- a lone wave
    //        issues one instruction per ~4 cycles whatever its type, so scalar bookkeeping per record
    //        is as expensive as vector work on the launch's critical path.
    auto fetch = [&](unsigned int base, unsigned int cnt, Rec& r) {
        if (lane < cnt) r = recs[own_order ? lds_idx[base - lb + lane] : (unsigned int)wk[base + lane]];
    };
    // Can ANY sample of the block be accepted?  Upper bound of alpha over the block: the minimum of
    // the conic's quadratic form q = a dx^2 + 2 b dx dy + c dy^2 over the block's sample rectangle
    // (convex: 0 if the centre is inside, else the best of the four edges) gives the largest
    // power = -q/2; below the record's certain-reject threshold (c.w, margin 1e-3) every fragment
    // of the record in this block is (0,0,0,0): RGB untouched.  A third of all (block, record)
    // overlaps on C3.  Their only effect, the alpha byte, is resolved by the alpha pass below.
    auto may_contribute = [&](const Rec& r) -> bool {
        const float a = r.b.x, b = r.b.y, c = r.b.z;
        // not positive definite: keep.  Nearly singular (a needle: b^2 within 0.1 % of a c): keep as well -- the minimum of
        // q is then the small difference of terms thousands of times its size, and its f32 rounding error would no
        // longer fit the margin below (found by tools/fuzz_parity.py: 3 of 12 000 hostile scenes lost a fragment of
        // alpha ~ 1/255 to it, one count in one or two pixels)
        const float ac = a * c;
        if (!(a > 0.0f && c > 0.0f && ac - b * b > 1e-3f * ac)) return true;
        const float x0 = xlo - r.a.x, x1 = xhi - r.a.x, y0 = r.a.y - yhi, y1 = r.a.y - ylo;
        // q is convex with its minimum (0) at the centre: over the rectangle the minimum is 0 if the centre
        // is inside, else it lies on an edge that FACES the centre (from any point of a far edge q
        // decreases along the segment towards the centre, which starts inside the rectangle) -- at most
        // one vertical and one horizontal edge, each minimised in closed form along its length.
        auto q = [&](float x, float y) { return a * x * x + 2.0f * b * x * y + c * y * y; };
        auto cl = [](float v, float lo, float hi) { return fminf(fmaxf(v, lo), hi); };
        // (v_rcp, 1 ulp: this is a bound with a 1e-3 margin, not the reference's arithmetic)
        const float nbc = -b * __builtin_amdgcn_rcpf(c), nba = -b * __builtin_amdgcn_rcpf(a);
        const bool xin = x0 <= 0.0f && x1 >= 0.0f, yin = y0 <= 0.0f && y1 >= 0.0f;
        const float ex = (x0 > 0.0f) ? x0 : x1, ey = (y0 > 0.0f) ? y0 : y1;      // the facing edges (when the interval excludes 0)
        const float qv = q(ex, cl(nbc * ex, y0, y1)), qh = q(cl(nba * ey, x0, x1), ey);
        const float big = 3.0e38f;
        float qmin = fminf(xin ? big : qv, yin ? big : qh);
        if (xin && yin) qmin = 0.0f;
        const float pmax = -0.5f * qmin;
        return !(pmax + 1e-3f * (1.0f + fabsf(pmax)) < r.c.w);
    };
    // scan_layout: the phase-A walk only estimates transmittance, so its records are staged in a
    // form that makes the estimate cheap -- the conic pre-multiplied by -log2(e)/2 and log2(opacity),
    // so that alpha ~ exp2(a' dx^2 + b' dx dy + c' dy^2 + l2o) is six VALU and one v_exp.
    // Batches are aligned to the END of the list (batch k = records [end - 64 (k + 1), end - 64 k), clipped at beg) in
    // every pass, so the verdicts of one pass serve the next: the scan saves its ballot (save_k), the exact walk
    // takes it (reuse_k; reuse_shift = records of the batch below the walk's start) instead of running the two
    // coverage tests and the contribution bound again -- 60 of the 76 instructions of a staging.
    // (One near -> far walk for the alpha pass AND the scan -- every covering record staged once with its rectangle, the
    // scan's form of the conic and the conic itself; a step serves both until every pixel has met a covering record -- was
    // built and measured in round 6: SLOWER, orbit 2530 vs 2640 frames/s on C3 (profiles/r06_fused_scan_ab.txt).  The alpha
    // pass is cheap as it is: its staging runs no contribution bound and its loop leaves after the handful of records that
    // cover the block, where the fused batches carry the third of the covering records that contribute nothing through
    // every scan step.)
    auto stage = [&](const Rec& r, unsigned int cnt, unsigned int base, bool only_contributing, bool scan_layout = false,
                     bool pair_layout = false, int reuse_k = -1, unsigned int reuse_shift = 0u, int save_k = -1) -> unsigned int {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // earlier LDS reads of this wave are done
        __builtin_amdgcn_wave_barrier();
        bool ov = false;
        if (reuse_k >= 0) {
            ov = (((wmask[reuse_k] >> reuse_shift) >> lane) & 1ull) != 0ull;
        } else if (lane < cnt) {
            ov = any_sample_covered(r.a.x, r.a.z, xlo, xhi, off) && any_sample_covered(r.a.y, r.a.w, ylo, yhi, off);
            if (ov && only_contributing) ov = may_contribute(r);
#if SPLAT_EXP_STAGE2
            // timing experiment (tools/lab, VERDICT r5 item 1's gate): the staging verdict computed a second time behind a
            // scheduling fence -- same frames, the launch's slowdown is what the verdicts cost
            Rec r2 = r;
            asm volatile("" : "+v"(r2.a.x), "+v"(r2.a.y), "+v"(r2.b.x), "+v"(r2.b.y), "+v"(r2.c.w));
            bool ov2 = any_sample_covered(r2.a.x, r2.a.z, xlo, xhi, off) && any_sample_covered(r2.a.y, r2.a.w, ylo, yhi, off);
            if (ov2 && only_contributing) ov2 = may_contribute(r2);
            ov = ov & ov2;
#endif
        }
        const unsigned long long m = __builtin_amdgcn_ballot_w64(ov);
        if (save_k >= 0 && lane == 0u) wmask[save_k] = m;
        if (ov) {
            const unsigned int slot = __builtin_amdgcn_mbcnt_hi((unsigned int)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)m, 0u));
            if (pair_layout) {
                float* P = reinterpret_cast<float*>(&L) + (slot >> 1) * 24u + (slot & 1u);
                P[0] = r.a.x; P[2] = r.a.y; P[4] = r.a.z; P[6] = r.a.w;
                P[8] = r.b.x; P[10] = r.b.z; P[12] = r.b.y; P[14] = r.b.w;
                float* Q = reinterpret_cast<float*>(&L) + (slot >> 1) * 24u + 2u * (slot & 1u);
                Q[16] = r.c.x; Q[17] = r.c.y; Q[20] = r.c.z;
            } else if (scan_layout) {
                const float L2E = 1.4426950408889634f;
                L.a[slot] = make_float4(r.a.x, r.a.y, __uint_as_float(base + lane), 0.0f);
                L.b[slot] = make_float4(-0.5f * L2E * r.b.x, -L2E * r.b.y, -0.5f * L2E * r.b.z, __log2f(r.b.w));
            } else {
                L.a[slot] = r.a; L.b[slot] = r.b;
                L.c[slot] = make_float4(r.c.x, r.c.y, r.c.z, __uint_as_float(base + lane));   // .w: list position
            }
        }
        const unsigned int kk = (unsigned int)__builtin_popcountll(m);
        if (pair_layout && (kk & 1u) && lane == 0) {
            // an odd batch is completed by a record that covers nothing (half extents -1: alpha 0, blend is the identity)
            float* P = reinterpret_cast<float*>(&L) + (kk >> 1) * 24u + 1u;
            P[0] = 0.0f; P[2] = 0.0f; P[4] = -1.0f; P[6] = -1.0f; P[8] = 0.0f; P[10] = 0.0f; P[12] = 0.0f; P[14] = 0.0f;
            float* Q = reinterpret_cast<float*>(&L) + (kk >> 1) * 24u + 2u;
            Q[16] = 0.0f; Q[17] = 0.0f; Q[20] = 0.0f;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        return kk;
    };
    // fragment(): src/pipelines.rs:134-143, branch-free.  Returns alpha, forced to 0 where the
    // fragment is rejected or the sample is not covered; `cov` reports coverage.
    auto expv = [&](float x) -> float {
        if constexpr (LIBM) return exp_libm(x, exptab);
        else return exp_neg(x);
    };
    auto frag_alpha = [&](const float4& a, const float4& b, auto e_of_power, bool& cov) -> float {
        float dx = sxm - a.x, dy = a.y - sym;          // K1 folded the y-axis sign into b.y
        cov = (fabsf(dx) <= a.z) & (fabsf(dy) <= a.w);
        float power = -0.5f * (b.x * dx * dx + b.z * dy * dy) - b.y * dx * dy;
        float alpha = fminf(0.99f, b.w * e_of_power(power));
        bool accept = cov & !(power > 0.0f) & !(alpha < 1.0f / 255.0f);
        return accept ? alpha : 0.0f;
    };

    bool need_far = false;                       // near selection: this wave's walk needs keys in front of the selection
    unsigned int itA = 0, itB = 0;               // (wave, record) iterations per phase, for the stats
    // ---------------- alpha pass: the alpha byte ----------------
    // blend() stores the NEW fragment's alpha (src/pipelines.rs:162-167), rejected fragments store 0,
    // so the byte is decided by each pixel's nearest covering record -- including the records that
    // cannot contribute colour and are dropped from the walks below.  Nearest first, until every
    // pixel of the block has met one.
    float alast = -1.0f;                         // < 0: no record covers the pixel, the byte keeps its old value
    if (!need_far) {
        bool found = !inside;
        Rec r;
        unsigned int cntN = min(64u, end - lb), bsN = end - cntN;
        fetch(bsN, cntN, r);
        while (true) {
            const unsigned int bs = bsN, cnt = cntN;
            const unsigned int k = stage(r, cnt, bs, false);
            if (bs > lb) { cntN = min(64u, bs - lb); bsN = bs - cntN; fetch(bsN, cntN, r); }
            // the walk itself only tests coverage (six VALU per record) and remembers which record of
            // the batch each pixel met first; the exact alpha is evaluated once per batch, every lane
            // on its own record
            unsigned int jsel = 0xffffffffu;
            for (unsigned int j = k; j-- > 0;) {
                const float4 a = L.a[j];
                const bool cov = (fabsf(sxm - a.x) <= a.z) & (fabsf(a.y - sym) <= a.w);
                jsel = (cov & !found) ? j : jsel;
                found = found | cov;
                if (__builtin_amdgcn_ballot_w64(!found) == 0ull) break;
            }
            if (__builtin_amdgcn_ballot_w64(jsel != 0xffffffffu) != 0ull) {
                const unsigned int jr = (jsel != 0xffffffffu) ? jsel : 0u;
                const float4 a = L.a[jr], b = L.b[jr];
                bool cov;
                const float alpha = frag_alpha(a, b, expv, cov);      // exact, 0 when rejected
                alast = (jsel != 0xffffffffu) ? alpha : alast;
            }
            if (__builtin_amdgcn_ballot_w64(!found) == 0ull || bs == lb) break;
        }
        // (near selection: a pixel that met no record among the selected keys may meet one among the farther ones)
        if (has_far && __builtin_amdgcn_ballot_w64(!found) != 0ull) need_far = true;
    }
    // ---------------- phase A: where must the exact walk start? ----------------
    unsigned int ws = lb;                        // this wave's start position (uniform)
    unsigned int nA = 0;                         // batches (from the end of the list) whose staging verdicts the scan saved
    const bool early = fc.early_eps > 0.0f && (has_far || end - beg >= (unsigned int)fc.early_min);
    // A camera at rest: the lists are the previous frame's lists, and so is the start its walk took -- start_hint, one word
    // per wave, left by the last frame that scanned (or had to retry deeper).  The walk from there closes its bracket as it
    // did then; if it ever did not (a hint from another camera: frames overlap on the device), the retry below takes over
    // as after any scan that stopped short.  Exactness never rests on the hint, only the scan's time does.
    bool scanned = false, hinted = false;
    if (early && !need_far && fc.start_hints != 0 && start_hint != nullptr) {
        unsigned int h = (unsigned int)__builtin_amdgcn_readfirstlane((int)start_hint[tile * 4u + wave]);
        if (fc.start_hints >= 2 && h != 0u) {
            // a camera in slow motion: the previous frames' start with a margin, and every fourth frame (tiles take turns) the scan;
            // in very slow motion (under ~0.15 degrees a frame: start_light) half the margin, every eighth frame
            if (fc.start_light) h = (((unsigned int)fc.start_hints + tile) & 7u) == 0u ? 0u : h + (h >> 4) + 16u;
            else h = (((unsigned int)fc.start_hints + tile) & 3u) == 0u ? 0u : h + (h >> 3) + 32u;
        }
        if (h != 0u) {
            if (h < end - lb) { ws = end - h; hinted = true; }
            else if (!has_far) { ws = lb; hinted = true; }       // (the whole list, as last time: no scan needed to find that out again)
        }
    }
    if (early && !need_far && !hinted) {
        scanned = true;
        float T = 1.0f;
        unsigned int sp = inside ? lb : 0xffffffffu;     // per lane: first layer the lane needs
        bool done = !inside;
        // a strip that has not saturated after half of its list will not save enough to pay for the scan
        // (a selection is scanned to its end: what lies behind it costs a sort of the whole list)
        const unsigned int giveup = has_far ? lb : end - (unsigned int)(((unsigned long long)(end - beg) * (unsigned int)fc.early_scan8) >> 3);
        Rec r;
        unsigned int cntN = min(64u, end - lb), bsN = end - cntN;
        fetch(bsN, cntN, r);
        while (true) {
            const unsigned int bs = bsN, cnt = cntN;
            const unsigned int k = stage(r, cnt, bs, true, true, false, -1, 0u, nA < 32u ? (int)nA : -1);
            nA += nA < 32u ? 1u : 0u;
            if (bs > lb) { cntN = min(64u, bs - lb); bsN = bs - cntN; fetch(bsN, cntN, r); }   // prefetch farther batch
            for (unsigned int j = k; j-- > 0;) {                  // nearest first
                const float4 a = L.a[j], b = L.b[j];
                // an estimate is enough here (too shallow a start costs a retry, never exactness):
                // no 3-sigma rectangle, no power > 0 case, v_exp instead of expf
                const float dx = sxm - a.x, dy = a.y - sym;
                const float p2 = fmaf(dx, fmaf(b.x, dx, b.y * dy), (b.z * dy) * dy) + b.w;
                float alpha = fminf(0.99f, __builtin_amdgcn_exp2f(p2));
                alpha = (p2 < -7.994353f) ? 0.0f : alpha;          // log2(1/255)
                T = fmaf(-alpha, T, T);
                const bool now = !done & (T < fc.early_eps);
                sp = now ? __float_as_uint(a.z) : sp;
                done = done | now;
            }
            itA += k;
            if (__builtin_amdgcn_ballot_w64(!done) == 0ull || bs == lb || bs <= giveup) break;
        }
        unsigned int need = done ? sp : lb;               // lanes that never saturated need the whole list
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) need = min(need, (unsigned int)__shfl_xor((int)need, o));
        ws = (unsigned int)__builtin_amdgcn_readfirstlane((int)need);
        if (ws == 0xffffffffu) ws = end;          // no pixel of this strip is on the target
    }

    // ---------------- phase B: exact compositing from the start layer ----------------
    // The bracket is closed when hi - lo <= cw on every channel of every pixel.  cw = 0: lo == hi, the exact frame.
    // cw = 2 (SPLAT_MODE_FAST): blend() is monotone in the state and never expands a difference of integer states
    // (|blend(x) - blend(y)| <= |x - y|: the real difference is (1 - alpha) |x - y|, not an integer unless alpha = 0,
    // where blend is the identity), so a state within 1 of the exact one stays within 1 to the nearest record.  The
    // middle of a bracket of width <= 2 is such a state: the walk continues with it alone, every channel ends within
    // 1 of the exact frame, and the walk may start where the transmittance is ~1e-3 instead of ~1e-6.
    const float cw = fc.close_width;
    float R = (float)((old >> 16) & 0xffu), G = (float)((old >> 8) & 0xffu), B = (float)(old & 0xffu);
    float R2 = 255.0f, G2 = 255.0f, B2 = 255.0f;  // upper end of the bracket
    // blend(): src/pipelines.rs:147-167.  With alpha == 0 it is the identity on the 8-bit state
    // (k/255*255 truncates back to k for every byte), so it runs unconditionally.
    // BR = true: carry the state twice (lo from 0, hi from 255).  Compile-time flag.
    auto shade = [&](auto BRt, const float4& a, const float4& b, const float4& c) {
        constexpr bool BR = decltype(BRt)::value;
        bool cov;
        const float alpha = frag_alpha(a, b, expv, cov);
        const float ia = 1.0f - alpha;
        const float ar = alpha * c.x, ag = alpha * c.y, ab = alpha * c.z;
        R = blend_channel(R, ia, ar);
        G = blend_channel(G, ia, ag);
        B = blend_channel(B, ia, ab);
        if (BR) {
            R2 = blend_channel(R2, ia, ar);
            G2 = blend_channel(G2, ia, ag);
            B2 = blend_channel(B2, ia, ab);
        }
    };
    // two records per step (see the note above WaveLds); q0..q5 = the pair's 24 staged floats
    const f2 sxm2 = {sxm, sxm}, sym2 = {sym, sym};
    auto shade_pair = [&](auto BRt, const float4& q0, const float4& q1, const float4& q2, const float4& q3, const float4& q4,
                          const float4& q5) {
        constexpr bool BR = decltype(BRt)::value;
        const f2 cx = {q0.x, q0.y}, cy = {q0.z, q0.w}, A = {q2.x, q2.y}, C = {q2.z, q2.w}, Bc = {q3.x, q3.y}, op = {q3.z, q3.w};
        const f2 dx = sxm2 - cx, dy = cy - sym2;                     // K1 folded the y-axis sign into the cross term
        const bool cov0 = (fabsf(dx.x) <= q1.x) & (fabsf(dy.x) <= q1.z), cov1 = (fabsf(dx.y) <= q1.y) & (fabsf(dy.y) <= q1.w);
        const f2 power = -0.5f * (A * dx * dx + C * dy * dy) - Bc * dx * dy;
        // exp_neg, both records
        const float L2E_HI = __uint_as_float(0x3fb8aa3bu), L2E_LO = __uint_as_float(0x32a5705fu), LN2 = 0.6931471805599453f;
        const f2 ph = power * L2E_HI;
        f2 pl = __builtin_elementwise_fma(power, (f2)(L2E_HI), -ph);
        pl = __builtin_elementwise_fma(power, (f2)(L2E_LO), pl);
        const f2 e = {__builtin_amdgcn_exp2f(ph.x), __builtin_amdgcn_exp2f(ph.y)};
        const f2 ex = __builtin_elementwise_fma(e * pl, (f2)(LN2), e);
        const f2 al = op * ex;
        const float a0 = fminf(0.99f, al.x), a1 = fminf(0.99f, al.y);
        const float alpha0 = (cov0 & !(power.x > 0.0f) & !(a0 < 1.0f / 255.0f)) ? a0 : 0.0f;
        const float alpha1 = (cov1 & !(power.y > 0.0f) & !(a1 < 1.0f / 255.0f)) ? a1 : 0.0f;
        auto blend_one = [&](float alpha, f2 rg, float b) {
            const float ia = 1.0f - alpha;
            const f2 arg = alpha * rg;
            const float ab = alpha * b;
            f2 s = blend_channel2((f2){R, G}, ia, arg);
            R = s.x; G = s.y;
            B = blend_channel(B, ia, ab);
            if (BR) {
                s = blend_channel2((f2){R2, G2}, ia, arg);
                R2 = s.x; G2 = s.y;
                B2 = blend_channel(B2, ia, ab);
            }
        };
        blend_one(alpha0, (f2){q4.x, q4.y}, q5.x);
        blend_one(alpha1, (f2){q4.z, q4.w}, q5.z);
    };
    // Walk batches [start, end).  In bracket mode stop at the first batch boundary where every
    // pixel has lo == hi and return that position; otherwise return `end`.
    auto run = [&](auto BRt, unsigned int start) -> unsigned int {
        constexpr bool BR = decltype(BRt)::value;
        Rec r;
        // the first batch ends where the scan's batch that contains `start` ends; the others are the scan's batches
        unsigned int kb = start < end ? (end - 1u - start) >> 6 : 0u;
        unsigned int bsN = start, cntN = start < end ? (end - (kb << 6)) - start : 0u;
        if (cntN) fetch(bsN, cntN, r);
        while (cntN) {
            const unsigned int bs = bsN, cnt = cntN;
            const bool reuse = kb < nA;
            const unsigned int abase = (end - lb >= ((kb + 1u) << 6)) ? end - ((kb + 1u) << 6) : lb;   // where the scan's batch began
            const unsigned int k = stage(r, cnt, bs, true, false, PAIR, reuse ? (int)kb : -1, bs - abase);
            bsN = bs + cnt; cntN = min(64u, end - bsN); kb -= cntN ? 1u : 0u;
            if (cntN) fetch(bsN, cntN, r);                  // prefetch the next (nearer) batch
            if constexpr (PAIR) {
                const float4* P4 = reinterpret_cast<const float4*>(&L);
                for (unsigned int j = 0; j < (k + 1u) / 2u; ++j)
                    shade_pair(BRt, P4[6 * j], P4[6 * j + 1], P4[6 * j + 2], P4[6 * j + 3], P4[6 * j + 4], P4[6 * j + 5]);
            } else {
                for (unsigned int j = 0; j < k; ++j) shade(BRt, L.a[j], L.b[j], L.c[j]);
            }
            itB += k;
            if (BR && __builtin_amdgcn_ballot_w64(inside & ((R2 - R > cw) | (G2 - G > cw) | (B2 - B > cw))) == 0ull) return bsN;
        }
        return end;
    };
    unsigned int start = ws;
    while (!need_far) {
        if (start <= lb) {                            // nothing skipped: exact from the real pixel ...
            if (has_far) { need_far = true; break; }  // ... unless unsorted keys lie in front of the selection: the whole list, then
            R = (float)((old >> 16) & 0xffu); G = (float)((old >> 8) & 0xffu); B = (float)(old & 0xffu);
            run(std::false_type{}, lb);
            break;
        }
        // skipped layers [beg, start): bracket them
        R = G = B = 0.0f; R2 = G2 = B2 = 255.0f;
        const unsigned int pos = run(std::true_type{}, start);
        const bool open = __builtin_amdgcn_ballot_w64(inside & ((R2 - R > cw) | (G2 - G > cw) | (B2 - B > cw))) != 0ull;
        if (!open) {
            // closed: continue single-state from the middle of the bracket (lo itself when lo == hi, the exact mode)
            R = truncf((R + R2) * 0.5f); G = truncf((G + G2) * 0.5f); B = truncf((B + B2) * 0.5f);
            if (pos < end) run(std::false_type{}, pos);
            break;
        }
        // lo != hi somewhere at the end of the list: not proven.  Retry from twice the depth
        // (geometric, so a long list is not redone in full for one stubborn LSB).
        if (lane == 0) atomicAdd(&late()->status->n_fallback, 1ull);
        const unsigned int depth = end - start;
        start = (start - lb > depth) ? start - depth : lb;
    }
    const float A = (alast < 0.0f) ? (float)(old >> 24) : truncf(alast * 255.0f);   // alpha in {0} U [1/255, .99]
    // statistics frames only (iters != nullptr): this wave's (scan, blend) iteration counts as one plain store
    // (atomics on two frame-wide counters cost ~0.2 ms per frame)
    // (keep_keys bit 1, SPLAT_DBG_STARTS: the list's length and how many of its nearest keys this wave's walk needed instead)
    // (a second walk ADDS: the wave walked this tile once already, with the selection, and left its counts here)
    const KernArgs kl = late();                  // (the arguments of the walk's end: see late())
    uint2* const iters_l = kl->iters;
    if (iters_l != nullptr && lane == 0) {
        const unsigned int kk = kl->keep_keys;
        uint2 v = (kk & 2u) ? make_uint2(end - beg, end - max(start, beg)) : make_uint2(itA, itB);
        if (second_walk && !(kk & 2u)) { const uint2 o = iters_l[item * 4u + wave]; v.x += o.x; v.y += o.y; }
        iters_l[item * 4u + wave] = v;
    }
    if constexpr (LONGM != 0) {
        // what this wave's walk needed of the list's near end, for the next frame's selection (see the prologue)
        unsigned int* const need_l = kl->need_hint;
        if (need_l != nullptr && end - beg > 2048u && lane == 0u) need_l[tile * 4u + wave] = need_far ? 0xffffffffu : end - max(start, lb);
    }
    // ... and where it started, for the next frame of a camera at rest (phase A): after a scan, or when the hinted start
    // did not do and the retry went deeper
    unsigned int* const start_l = kl->start_hint;
    if (start_l != nullptr && early && !need_far && (scanned || start != ws) && lane == 0u) {
        unsigned int used = max(end - max(start, lb), 1u);
        if (!has_far && used > (end - beg) - ((end - beg) >> 2)) used = end - beg;     // (most of the list anyway: all of it, without a bracket)
        start_l[tile * 4u + wave] = used;
    }
    if (need_far) return true;
    if (inside)
        kl->argb[(size_t)py * kl->fc.W + px] = ((uint32_t)A << 24) | ((uint32_t)R << 16) | ((uint32_t)G << 8) | (uint32_t)B;
    return false;
    };      // walk_list
    if constexpr (LONGM != 2) {
        (void)walk_list();
    } else {
        // NEAR SELECTION, when the selection did not do for every wave (its pixels did not saturate within the selected keys, the
        // bracket did not close there, a pixel met no record): the TILE REPAIRS ITSELF.  The workgroup is resident and holds the
        // workspace -- its four waves meet, sort the whole list (what the sort launches would have left: flavour 1's code), and
        // the waves that asked walk again, now over a list that is in order from its first key.  (Round 5 put such tiles on a
        // list for a repair LAUNCH behind the frame: a serial launch of a few workgroups on a chip the next frames' binning
        // keeps busy -- 0.1 ms behind every frame of a camera in motion on the surface scene, profiles/r05_motion_probe.txt.)
        // The repair is a FUNCTION CALL, not inlined code: with the long-list sort and a second copy of the walks inlined the
        // kernel needed scratch and half its uniform values spilled -- the hot path paid 4.4 % for a path a frame at rest never
        // takes (profiles/r06_repair_ab.txt).
        const bool again = walk_list();
        if (has_far && __syncthreads_or(again ? 1 : 0) != 0) {                // (has_far is uniform in the workgroup: the tile's near_m)
            if (tid == 0u) atomicAdd(&late()->status->n_near_fallback, 1u);
            repair_tile<PAIR, LIBM>(ka, (unsigned int)(size_t)(__attribute__((address_space(3))) unsigned char*)smem,
                                    (unsigned int)(size_t)(__attribute__((address_space(3))) const unsigned long long*)exptab, item, again);
        }
    }
}

// The rarely taken path of a near-selection frame: the tile's whole list sorted by its workgroup, the waves in `again` walked
// over it (flavour 1 of composite_tile: sort_long_list, then the walks).  Not inlined: see composite_tile's last lines.
// Everything but the workgroup's own state comes from the kernarg segment again; LDS addresses travel as offsets.
// (static: with every caller in sight the compiler hands the kernel's seven-waves-per-SIMD register budget down to this
// function -- 72 VGPRs, the rest on the stack -- instead of raising the kernel's to the 150 it would like.)
template <bool PAIR, bool LIBM>
static __device__ __attribute__((noinline)) void repair_tile(KernArgs ka, unsigned int smem_lds, unsigned int exptab_lds, unsigned int item, bool again) {
    // (function arguments arrive in vector registers: make the uniform ones uniform again)
    const unsigned long long kp = uniform_u64((unsigned long long)(size_t)ka);
    const KernArgs k = (KernArgs)(size_t)kp;
    unsigned char* const smem = (unsigned char*)(__attribute__((address_space(3))) unsigned char*)(size_t)__builtin_amdgcn_readfirstlane((int)smem_lds);
    const unsigned long long* const exptab = (const unsigned long long*)(__attribute__((address_space(3))) const unsigned long long*)(size_t)__builtin_amdgcn_readfirstlane((int)exptab_lds);
    const unsigned int it = (unsigned int)__builtin_amdgcn_readfirstlane((int)item);
    FrameConst fc;                                  // (word by word out of the constant address space: scalar loads)
    {
        const __attribute__((address_space(4))) unsigned int* src = (const __attribute__((address_space(4))) unsigned int*)&k->fc;
        unsigned int* dst = reinterpret_cast<unsigned int*>(&fc);
#pragma unroll
        for (unsigned int q = 0; q < sizeof(FrameConst) / 4u; ++q) dst[q] = src[q];
    }
    composite_tile<PAIR, LIBM, 1>(smem, exptab, it, fc, k->offsets, k->order, k->lens, k->keys, k->recs, k->argb, k->status, k->fused_sort_max, k->radix_min,
                                  k->iters, k->keep_keys, k->orig, k->clear_first, k->keys2, nullptr, k->need_hint, k->start_hint, k->off2, k, again, true);
}

// NEAR SELECTION, the kernel (one workgroup per slot of the longest-first tile order; 256 threads and the compositor's
// 21.5 KB of LDS, so that its workgroups find room beside a compositor in flight -- the sort launches it replaces need
// 74 / 147 KB each and starve there).  A list of more than 2048 keys is NOT sorted: how many of its nearest keys the tile's
// walks will need is known from the previous frame -- every wave leaves it in need_hint, four words per tile, plain
// stores: 0 = nothing known, ~0 = more than it was given -- and the deepest of the four, with a margin, is
// selected by depth and put in order:
//   through LDS (select_near + the short lists' sort) while that fits the 2048-key workspace: the sorted selection goes
//   to the START of the second key buffer's region, the region itself keeps the whole unordered list;
//   as sorted parts (sort_long_list with a cap) beyond that; and a tile that ran out of its selection last time, or whose
//   selection would be most of the list anyway, is sorted in full, in its region, as a sort launch would leave it.
// near_m[tile] = how many of the nearest keys are in order (== the list's length: all of them).  A wrong guess costs time
// only -- a larger sort than necessary, or a tile that sorts its whole list inside the compositor after all -- never a pixel.
__global__ __launch_bounds__(256) void select_near_kernel(const unsigned int* __restrict__ offsets, const unsigned int* __restrict__ order,
                                                          const unsigned int* __restrict__ lens, unsigned long long* __restrict__ keys,
                                                          unsigned long long* __restrict__ keys2, FrameStatus* __restrict__ status,
                                                          const unsigned int* __restrict__ orig, unsigned int radix_min, unsigned int near_cap,
                                                          const unsigned int* __restrict__ need_hint, unsigned int* __restrict__ near_m,
                                                          unsigned int tiles_x, unsigned int tile_rows, unsigned int* __restrict__ near_thr,
                                                          unsigned int n_slots, unsigned int at_rest, const unsigned int* __restrict__ off2,
                                                          int hint_radius) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[sort_lds_bytes<256, 2048>()];
    if (status->overflow) return;
    // (an eighth as many workgroups as tiles, each taking the slots blockIdx.x, + gridDim.x, ... of the longest-first order until it
    // meets a list the compositor sorts itself: seven of eight tiles have one, and a workgroup launched only to find that
    // out still has to wait for 21.5 KB of LDS on a chip the compositor fills)
  for (unsigned int slot = blockIdx.x; slot < n_slots; slot += gridDim.x) {
    const unsigned int tile = (unsigned int)__builtin_amdgcn_readfirstlane((int)order[slot]);
    const unsigned int n = (unsigned int)__builtin_amdgcn_readfirstlane((int)lens[tile]);
    // (the compositor's workgroup sorts those itself -- and every slot behind this one: the order is by length CLASS, half
    // octaves, longest class first, so a list of exactly 2048 keys may stand in front of longer ones of its class)
    if (n < 2048u) return;
    if (n == 2048u) continue;
    const unsigned int beg = (unsigned int)__builtin_amdgcn_readfirstlane((int)offsets[tile]);
    unsigned long long* const k2 = keys2 + (unsigned int)__builtin_amdgcn_readfirstlane((int)off2[tile]);     // the tile's room in the second key buffer
    unsigned int want = near_cap, deepest, thr = 0u;
    {
        // ONE wave reads the hints and the workgroup takes its word for them: the previous frames' compositors are storing
        // into these words while this kernel runs (frames overlap on the device), and four waves that each read for
        // themselves can disagree -- on which path to take through the barriers below.
        unsigned int* const word = reinterpret_cast<unsigned int*>(smem);
        if (threadIdx.x < 64u) {
            const uint4 h4 = reinterpret_cast<const uint4*>(need_hint)[tile];
            unsigned int d0 = max(max(h4.x, h4.y), max(h4.z, h4.w));
            // ... and the tiles around it (a camera in motion carries a deep spot of the image from tile to tile: what the
            // neighbours' walks needed last frame is what this tile's may need now), as far around as the camera moved since:
            // hint_radius tiles, 2 for a camera at rest or creeping, up to 7 for a 10-degree step.  A neighbour that ran out (~0)
            // says nothing.
            if (hint_radius > 0) {
                const int D = 2 * hint_radius + 1;
                unsigned int nb = 0u;
                for (int idx = (int)threadIdx.x; idx < D * D; idx += 64) {
                    const int tx = (int)(tile % tiles_x) + idx % D - hint_radius, ty = (int)(tile / tiles_x) + idx / D - hint_radius;
                    if (tx >= 0 && ty >= 0 && tx < (int)tiles_x && ty < (int)tile_rows) {
                        const uint4 q = reinterpret_cast<const uint4*>(need_hint)[(unsigned int)ty * tiles_x + (unsigned int)tx];
                        const unsigned int a = q.x == 0xffffffffu ? 0u : q.x, b = q.y == 0xffffffffu ? 0u : q.y;
                        const unsigned int cc = q.z == 0xffffffffu ? 0u : q.z, d = q.w == 0xffffffffu ? 0u : q.w;
                        nb = max(nb, max(max(a, b), max(cc, d)));
                    }
                }
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) nb = max(nb, (unsigned int)__shfl_xor((int)nb, o));
                d0 = max(d0, nb);           // (~0 stays ~0)
            }
            if (threadIdx.x == 0u) { word[0] = d0; word[1] = near_thr != nullptr ? near_thr[tile] : 0u; }
        }
        __syncthreads();
        deepest = (unsigned int)__builtin_amdgcn_readfirstlane((int)word[0]);
        thr = (unsigned int)__builtin_amdgcn_readfirstlane((int)word[1]);
        __syncthreads();                    // (the workspace is the selection's from here on)
        // (While the hint leaves room, the selection is the full workspace: this kernel runs beside the previous frame's
        // compositor, its time is hidden, and a moving camera puts other Gaussians under the tile than the hint saw -- a
        // 36-pose orbit, 10 degrees a frame, lost a third of its rate to repairs with selections sized tightly.)
        // (A camera at rest needs no room for surprises: half as much again as the walks took, in steps of 256 keys -- what is not
        // selected is not sorted.)
        if (deepest == 0xffffffffu) want = n;
        else if (deepest != 0u) {
            const unsigned int room = deepest + (deepest >> 1) + 128u;
            want = room <= near_cap ? (at_rest ? min(near_cap, max(512u, (room + 255u) & ~255u)) : near_cap) : 2u * deepest + 256u;
        }
        if (want > n - (n >> 2)) want = n;      // (three quarters of the list: the whole list, then, and no repair to fear)
    }
    unsigned int m = 0u;
    if (want <= near_cap) {
        unsigned int smn, smx;
        // ONE pass while the camera is coherent: every key at or beyond the depth the tile's last selection began at -- if
        // those are still a workspace-full at most and three quarters of one at least (a moving camera finds its margin in
        // the size of the selection); else the two passes
        // (histogram over a sampled depth range, compaction), which leave the depth for the next frame
        if (thr != 0u && deepest != 0u) {
            m = select_by_depth<256>(smem, keys + beg, n, thr, &smn, &smx);
            const unsigned int least = at_rest ? deepest + (deepest >> 3) : max(deepest + (deepest >> 1) + 128u, near_cap - (near_cap >> 2));
            if (m > near_cap || m < least || m >= n) m = 0u;
        }
        if (m == 0u) {
            unsigned int thr_new = 0u;
            m = select_near<256>(smem, keys + beg, n, want, &smn, &smx, &thr_new);
            if (near_thr != nullptr && threadIdx.x == 0u) near_thr[tile] = m != 0u ? thr_new : 0u;
        }
        // (the bins are coarse where many keys share a depth: a selection that came out shorter than the tile needed last
        // time is not worth walking)
        if (m != 0u && (deepest == 0u || m >= deepest + (deepest >> 3))) {
            sort_list_in_lds<256, 2048>(smem, nullptr, k2, m, radix_min, status, orig, nullptr, true, 0u, smn, smx);
        } else {
            m = 0u; want = min(n, 2u * near_cap);
            __syncthreads();
        }
    }
#if SPLAT_EXP_SKIPFULL
    // timing experiment (frames invalid): lists that would be sorted in full, of SPLAT_EXP_SKIPFULL keys or more, are left alone
    if (m == 0u && want >= n && n >= (unsigned int)SPLAT_EXP_SKIPFULL) { m = n; if (threadIdx.x == 0u) atomicAdd(&status->n_fallback, 1u); }
#endif
    if (m == 0u)
        m = sort_long_list(smem, keys + beg, k2, n, want, radix_min, status, orig,
                           (deepest != 0u && deepest != 0xffffffffu) ? min(n, deepest + (deepest >> 3)) : 0u);
    if (threadIdx.x == 0u) {
        near_m[tile] = m;
        if (m < n) atomicAdd(&status->n_near_tiles, 1u);
    }
    __syncthreads();                            // the workspace is the next slot's
  }
}

// The launch: one workgroup per tile, slot blockIdx.x of the longest-first order.  (A persistent grid pulling
// slots from a ticket counter, and the order composited as consecutive chunk launches, were both measured as
// ways to cap the compositor's residency beside the next frame's K1: both slower -- DESIGN.md section 3.)
// Flavours that carry the long-list sort (1: always; 2: a tile of a near-selection frame that repairs itself): seven
// waves per SIMD = seven workgroups per CU, which is also what the 21.5 KB of LDS allow: at most 72 VGPRs.  The walks
// need 64-69; the long-list sort would take 88, and under this bound spills nine registers around its loop over the
// parts instead -- outside every hot loop (checked in the ISA).
// (At most 96 SGPRs: with 97-112 a CU admits six 256-thread workgroups instead of seven -- MI355X_MICROARCH.md,
// "Residency" -- and the compositor hides its LDS and dependency latency with residency.  The near-selection flavour
// carries a few more uniform values than the others and would take 106.)
template <bool PAIR, bool LIBM, int LONGM>
__global__ __launch_bounds__(256, (LONGM != 0 && !PAIR) ? 7 : SPLAT_COMP_WAVES) __attribute__((amdgpu_num_sgpr(96))) void composite_exact_kernel(CompArgs a) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[sort_lds_bytes<256, 2048>()];
    __shared__ unsigned long long exptab[LIBM ? 32 : 1];
    if (a.status->overflow) return;
    if constexpr (LIBM) {
        static_assert(!PAIR, "the libm exponential is built for the one-record walk only");
        if (threadIdx.x < 32) exptab[threadIdx.x] = EXP2F_TAB[threadIdx.x];
        __syncthreads();
    }
    composite_tile<PAIR, LIBM, LONGM>(smem, exptab, blockIdx.x, a.fc, a.offsets, a.order, a.lens, a.keys, a.recs, a.argb, a.status, a.fused_sort_max, a.radix_min, a.iters, a.keep_keys, a.orig,
                                      a.clear_first, a.keys2, a.near_m, a.need_hint, a.start_hint, a.off2, (KernArgs)__builtin_amdgcn_kernarg_segment_ptr());
}

// ---------------------------------------------------------------------------
// launch wrappers
// ---------------------------------------------------------------------------
// Experiment switches of the launches below: the CONTEXT's (read from the environment at splat_create), handed over by
// the API layer for the calling thread before it enqueues a frame -- two contexts of one process may differ.
static const LaunchKnobs g_default_knobs{};
static thread_local const LaunchKnobs* g_knobs = &g_default_knobs;
void use_launch_knobs(const LaunchKnobs* k) { g_knobs = k ? k : &g_default_knobs; }
static unsigned int sort_radix_min() { return g_knobs->sort_radix_min; }
static inline unsigned int blocks_for(uint64_t n, unsigned int bs) { return (unsigned int)((n + bs - 1) / bs); }

// Per-DEVICE kernel attributes (the large sort classes need more dynamic LDS than the default limit): called by
// splat_create with its device current, so every GPU a process opens a context on is covered.
hipError_t init_device_kernels() {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(sort_tiles_kernel<1024, 16384>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, sort_lds_bytes<1024, 16384>());
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute(reinterpret_cast<const void*>(sort_tiles_kernel<512, 8192>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, sort_lds_bytes<512, 8192>());
}

void launch_pack_scene(hipStream_t s, uint64_t n, const float* pos4, const float* cov3d, const float* opacity,
                       const float* sh, const unsigned int* perm, float4* planes) {
    if (!n) return;
    hipLaunchKernelGGL(pack_scene_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, s, n, pos4, cov3d, opacity, sh, perm, planes);
}
void launch_cov3d(hipStream_t s, uint64_t n, const float* scales3, const float* rot4, float* cov3d) {
    if (!n) return;
    hipLaunchKernelGGL(cov3d_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, s, n, scales3, rot4, cov3d);
}
void launch_preprocess(hipStream_t s, uint64_t n, const float4* planes, const unsigned int* orig, FrameConst fc, Rec* recs,
                       float* depth, ushort4* rect, unsigned int* counts, unsigned int* vislist, unsigned long long* keys,
                       const BlockBounds* bounds, unsigned int* blockinfo, FrameStatus* status, const unsigned int* layout, bool count_only,
                       uint4* large_list, unsigned int* large_count) {
    if (!n) return;
    if (!bounds || !blockinfo) fc.cull_blocks = 0;
    if (!blockinfo || !layout) fc.bucket_cap = 0;
    if (!large_count || !fc.bucket_cap) { large_list = nullptr; large_count = nullptr; }      // (a counter without a list: large splats are only counted)
    const dim3 grid(blocks_for(n, 256)), block(256);
    auto go = [&](auto kern) { hipLaunchKernelGGL(kern, grid, block, g_knobs->k1_lds_pad, s, n, planes, orig, fc, recs, depth, rect, counts, vislist, keys, bounds, blockinfo, status, large_list, large_count); };
    if (fc.bucket_cap && count_only) { if (fc.corrected) go(preprocess_kernel<true, true, true>); else go(preprocess_kernel<true, false, true>); }
    else if (fc.bucket_cap && fc.corrected) go(preprocess_kernel<true, true>);
    else if (fc.bucket_cap) go(preprocess_kernel<true, false>);
    else if (fc.corrected) go(preprocess_kernel<false, true>);
    else go(preprocess_kernel<false, false>);
}
void launch_bin_large(hipStream_t s, const FrameConst& fc, const uint4* large_list, const unsigned int* large_count, unsigned int large_cap, unsigned int* cursors,
                      unsigned long long* keys, const FrameStatus* status, bool count_only) {
    if (!large_list || !large_count || !fc.bucket_cap || fc.tiles_x <= 0 || fc.n_tile_rows <= 0) return;
    const unsigned int groups = (unsigned int)((fc.tiles_x + LARGE_G - 1) / LARGE_G) * (unsigned int)((fc.n_tile_rows + LARGE_G - 1) / LARGE_G);
    if (count_only)
        hipLaunchKernelGGL(bin_large_kernel<true>, dim3(groups), dim3(256), 0, s, large_list, large_count, large_cap, cursors, keys, fc.bucket_cap, fc.tiles_x, fc.n_tile_rows,
                           status, fc.redo_only ? 1u : 0u);
    else
        hipLaunchKernelGGL(bin_large_kernel<false>, dim3(groups), dim3(256), 0, s, large_list, large_count, large_cap, cursors, keys, fc.bucket_cap, fc.tiles_x, fc.n_tile_rows,
                           status, fc.redo_only ? 1u : 0u);
}
void launch_scan(hipStream_t s, unsigned int m, unsigned int* counts, unsigned int* offsets, unsigned int* cursor,
                 unsigned int* order, unsigned int* lens, FrameStatus* status, unsigned long long capacity,
                 unsigned int bucket_cap, unsigned int grid_big, unsigned int grid_mid, unsigned int grid_long,
                 FrameStatus* host_status, const unsigned int* layout, unsigned int* next_layout, unsigned int* next_counts, float spare_max,
                 bool redo_only, unsigned int* off2, unsigned int cap2, unsigned int* large_count, unsigned int tiles_x, unsigned int motion_radius) {
    if (bucket_cap && layout)
    {
        const unsigned int nwg = (next_layout && next_counts) ? 2u : 1u;
        // small grids: 256 threads start at once beside a busy compositor; 4K-sized ones need the width
        const int nt = g_knobs->scan_threads ? g_knobs->scan_threads : (m > 12000u ? 1024 : 256);
        // one byte of LDS per tile for the length classes (up to 48 KB: a 6-megapixel target), else they are re-read
        const unsigned int cls_bytes = (m + 15u) & ~15u, in_lds = cls_bytes <= 49152u ? 1u : 0u;
        // (the layout workgroup's motion filter: two u16 per tile -- up to 12 288 tiles, a 1440p target; larger ones keep the
        // regions sized from each tile's own list)
        const unsigned int mv_bytes = (4u * m + 15u) & ~15u;
        if (nwg < 2u || mv_bytes > 49152u || tiles_x == 0u) motion_radius = 0u;
        motion_radius = std::min(motion_radius, 12u);          // (build_layout's window: RMAX)
        const unsigned int dyn = std::max(in_lds ? cls_bytes : 0u, motion_radius ? mv_bytes : 0u);
        if (nt == 256)
            hipLaunchKernelGGL(scan_bucket_kernel<256>, dim3(nwg), dim3(256), dyn, s, m, counts, offsets, order, lens, status, layout,
                               grid_big, grid_mid, grid_long, in_lds, host_status, next_layout, next_counts, bucket_cap, spare_max, redo_only ? 1u : 0u, off2, cap2, large_count, tiles_x, motion_radius);
        else if (nt == 512)
            hipLaunchKernelGGL(scan_bucket_kernel<512>, dim3(nwg), dim3(512), dyn, s, m, counts, offsets, order, lens, status, layout,
                               grid_big, grid_mid, grid_long, in_lds, host_status, next_layout, next_counts, bucket_cap, spare_max, redo_only ? 1u : 0u, off2, cap2, large_count, tiles_x, motion_radius);
        else
            hipLaunchKernelGGL(scan_bucket_kernel<1024>, dim3(nwg), dim3(1024), dyn, s, m, counts, offsets, order, lens, status, layout,
                               grid_big, grid_mid, grid_long, in_lds, host_status, next_layout, next_counts, bucket_cap, spare_max, redo_only ? 1u : 0u, off2, cap2, large_count, tiles_x, motion_radius);
    }
    else
        hipLaunchKernelGGL(scan_kernel, dim3(1), dim3(1024), 0, s, m, counts, offsets, cursor, order, lens, status, capacity,
                           bucket_cap, grid_big, grid_mid, grid_long, host_status);
}
void launch_layout(hipStream_t s, unsigned int m, const unsigned int* counts, const unsigned int* layout, unsigned int* next_layout,
                   unsigned int* next_counts, unsigned int key_entries, FrameStatus* status, FrameStatus* host_status, float spare_max,
                   const FrameStatus* redo_gate, unsigned int* large_count) {
    if (redo_gate != nullptr && m <= 12000u)        // (above that -- a 4K target -- the launch keeps its sixteen waves, as the scan does: when it does run, it runs long)
        hipLaunchKernelGGL(layout_kernel<256>, dim3(1), dim3(256), 0, s, m, counts, layout, next_layout, next_counts, key_entries, status, host_status, spare_max, redo_gate, large_count);
    else
        hipLaunchKernelGGL(layout_kernel<1024>, dim3(1), dim3(1024), 0, s, m, counts, layout, next_layout, next_counts, key_entries, status, host_status, spare_max, redo_gate, large_count);
}
void launch_emit(hipStream_t s, uint64_t n, FrameConst fc, const float* depth, const ushort4* rect, const unsigned int* orig,
                 const unsigned int* vislist, unsigned int* cursor, unsigned long long* keys, const FrameStatus* status) {
    if (!n) return;
    hipLaunchKernelGGL(emit_kernel, dim3(blocks_for(n, 256 * EMIT_G)), dim3(256), 0, s, fc, depth, rect, orig, vislist, cursor, keys, status);
}
void launch_sort(hipStream_t s, unsigned int n_tiles, unsigned int grid_big, unsigned int grid_mid, unsigned int grid_long, const unsigned int* offsets,
                 const unsigned int* order, const unsigned int* lens, unsigned long long* keys, unsigned long long* keys2,
                 FrameStatus* status, const unsigned int* orig, unsigned int fused_sort_max, const unsigned int* off2) {
    if (!n_tiles) return;
    if (!off2) off2 = offsets;          // (two-pass binning: the second buffer mirrors the first, list for list)
    const unsigned int radix_min = sort_radix_min();
    // longest class first (the tiles are ordered longest-first too)
    grid_big = std::min(grid_big, n_tiles); grid_mid = std::min(grid_mid, n_tiles);
    if (grid_big) {
        // four workgroups per list: lists of 16385..65536 keys are sorted as runs of 16384 and merged
        // (two-pass binning only: one-pass buckets hold at most 16384 keys and there is no keys2)
        grid_long = keys2 ? std::min(grid_long, grid_big) : 0u;
        hipLaunchKernelGGL((sort_tiles_kernel<1024, 16384>), dim3(grid_big + 3u * grid_long), dim3(1024), (sort_lds_bytes<1024, 16384>()), s,
                           offsets, order, lens, keys, keys2, status, 8192u, radix_min, keys2 ? 4 : 1, grid_big, std::max(grid_long, 1u), orig, off2);
        if (grid_long)
            hipLaunchKernelGGL((merge_runs_kernel<1024, 16384>), dim3(grid_long), dim3(1024), 0, s, offsets, order, lens, keys, keys2,
                               status, orig, off2);
    }
    if (grid_mid)
    hipLaunchKernelGGL((sort_tiles_kernel<512, 8192>), dim3(grid_mid), dim3(512), (sort_lds_bytes<512, 8192>()), s, offsets, order,
                       lens, keys, keys2, status, 2048u, radix_min, 0, 0u, 1u, orig, off2);
    if (fused_sort_max < 2048u)      // (lists up to fused_sort_max are sorted by the compositor's own workgroups)
        hipLaunchKernelGGL((sort_tiles_kernel<256, 2048>), dim3(n_tiles), dim3(256), (sort_lds_bytes<256, 2048>()), s, offsets, order,
                           lens, keys, keys2, status, fused_sort_max, radix_min, 0, 0u, 1u, orig, off2);
}
void launch_select(hipStream_t s, unsigned int n_tiles, const unsigned int* offsets, const unsigned int* order, const unsigned int* lens,
                   unsigned long long* keys, unsigned long long* keys2, FrameStatus* status, const unsigned int* orig, unsigned int near_cap,
                   const unsigned int* need_hint, unsigned int* near_m, unsigned int tiles_x, unsigned int tile_rows, unsigned int* near_thr, unsigned int grid, bool at_rest,
                   const unsigned int* off2, int hint_radius) {
    if (!n_tiles) return;
    if (!off2) off2 = offsets;
    if (g_knobs->dbg_hint_radius >= 0) hint_radius = g_knobs->dbg_hint_radius;
    if (g_knobs->dbg_select_stride) grid = (n_tiles + g_knobs->dbg_select_stride - 1u) / g_knobs->dbg_select_stride;
    if (!grid) grid = (n_tiles + 7u) / 8u;
    hipLaunchKernelGGL(select_near_kernel, dim3(std::min(grid, n_tiles)), dim3(256), 0, s, offsets, order, lens, keys, keys2, status, orig, sort_radix_min(),
                       std::min(std::max(near_cap, 64u), 2048u), need_hint, near_m, tiles_x, tile_rows, near_thr, n_tiles, at_rest ? 1u : 0u, off2, std::min(std::max(hint_radius, 0), 7));
}
void launch_composite(hipStream_t s, unsigned int n_tiles, FrameConst fc, const unsigned int* offsets,
                      const unsigned int* order, const unsigned int* lens, unsigned long long* keys, const Rec* recs,
                      uint32_t* argb, FrameStatus* status, const unsigned int* orig, unsigned int fused_sort_max, uint2* iters,
                      bool keep_keys, bool pair_walk, bool libm_exp, bool clear_first, unsigned long long* keys2, const unsigned int* near_m,
                      unsigned int* need_hint, unsigned int* start_hint, const unsigned int* off2) {
    if (!n_tiles) return;
    if (!off2) off2 = offsets;
    if (g_knobs->dbg_ntiles) n_tiles = std::min(n_tiles, g_knobs->dbg_ntiles);   // debug: composite only the N longest tiles
    // SPLAT_COMP_LDS_PAD: extra dynamic LDS per workgroup, i.e. an occupancy cap (12 KB are in use:
    // 13 workgroups fit a CU's LDS, 8 its wave slots) -- for overlapping the next frame's K1
    const unsigned int pad = g_knobs->comp_lds_pad;
    const bool near = near_m != nullptr && keys2 != nullptr && need_hint != nullptr;
    const unsigned int flags = (keep_keys ? 1u : 0u) | (g_knobs->dbg_starts ? 2u : 0u);
    CompArgs a;
    a.fc = fc; a.offsets = offsets; a.order = order; a.lens = lens; a.keys = keys; a.recs = recs; a.argb = argb; a.status = status;
    a.fused_sort_max = fused_sort_max; a.radix_min = sort_radix_min(); a.iters = iters; a.keep_keys = flags; a.clear_first = clear_first ? 1u : 0u;
    a.orig = orig; a.keys2 = keys2; a.near_m = near_m; a.need_hint = near ? need_hint : nullptr; a.start_hint = start_hint; a.off2 = off2;
    auto go = [&](auto kern) { hipLaunchKernelGGL(kern, dim3(n_tiles), dim3(256), pad, s, a); };
    if (near) {         // (a tile the selection does not serve repairs itself: no launch behind this one)
        if (libm_exp) go(composite_exact_kernel<false, true, 2>);
        else if (pair_walk) go(composite_exact_kernel<true, false, 2>);
        else go(composite_exact_kernel<false, false, 2>);
    } else if (keys2 != nullptr) {
        if (libm_exp) go(composite_exact_kernel<false, true, 1>);
        else if (pair_walk) go(composite_exact_kernel<true, false, 1>);
        else go(composite_exact_kernel<false, false, 1>);
    } else {
        if (libm_exp) go(composite_exact_kernel<false, true, 0>);
        else if (pair_walk) go(composite_exact_kernel<true, false, 0>);
        else go(composite_exact_kernel<false, false, 0>);
    }
}

}  // namespace splat
