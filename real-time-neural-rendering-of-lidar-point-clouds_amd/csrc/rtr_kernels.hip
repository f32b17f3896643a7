// rtr_kernels.hip -- hand-written gfx950 (MI355X, wave64) kernels of the projector.
//
// Arithmetic contract (same as oracle/rtr_oracle.c): every fp32 op is
// one IEEE RNE operation, fused only where fmaf() is written.  This file MUST be
// compiled with -ffp-contract=off and hipcc's default correctly rounded fp32
// divide / sqrt.  Nothing here is GEMM-shaped: the path is an HBM-bound stream plus
// a scatter, so the work goes into coalescing, atomic traffic and launch count.
#include "rtr_kernels.h"

#include <hip/hip_ext.h>
#include <hip/hip_fp16.h>

#include <type_traits>

namespace rtr {

#define RTR_EMPTY 0x7F7FFFFFu
constexpr int kBlock = 256;   // 4 waves
// The grid-stride point kernels run Cloud::grid workgroups (kDefaultPointGrid = 1024, i.e. 4
// per CU, measured best: 1024 -> 205 us, 1536 -> 218, 2048 -> 239 for k_project_bin; 2048 would
// not even be co-resident: its 84 SGPRs admit 7 x 256 threads per CU, not 8).

// one rounding per operation: plain operators under -ffp-contract=off (hipcc's __fmul_rn &
// co. are the same plain operators; __fsqrt_rn is NOT correctly rounded, sqrtf is)
__device__ __forceinline__ float f_mul(float a, float b) { return a * b; }
__device__ __forceinline__ float f_add(float a, float b) { return a + b; }
__device__ __forceinline__ float f_sub(float a, float b) { return a - b; }

// ---------------------------------------------------------------------------------
// projection of one point: render.cu:33-40 (matmul rows 0..2), :63 (z cull),
// :65-66 (rintf of the quotient), :68 (frustum cull), :70 (pixel id).
// Returns pixel id or -1.
__device__ __forceinline__ int project_point(const Proj &P, float x, float y, float z, int W, int H, float fW,
                                             float fH, float &depth) {
    float rx = f_add(fmaf(P.m[2], z, fmaf(P.m[1], y, f_mul(P.m[0], x))), P.m[3]);
    float ry = f_add(fmaf(P.m[6], z, fmaf(P.m[5], y, f_mul(P.m[4], x))), P.m[7]);
    float rz = f_add(fmaf(P.m[10], z, fmaf(P.m[9], y, f_mul(P.m[8], x))), P.m[11]);
    float inv = 1.0f / rz;  // correctly rounded (v_div_scale / v_div_fmas / v_div_fixup)
    float fu = rintf(f_mul(rx, inv));
    float fv = rintf(f_mul(ry, inv));
    bool ok = (rz > 0.0f) && (fu >= 0.0f) && (fu < fW) && (fv >= 0.0f) && (fv < fH);
    depth = rz;
    return ok ? ((int)fv * W + (int)fu) : -1;
}

// ---------------------------------------------------------------------------------
// A1 + A2 fused: depth <- sentinel, accumulators <- 0, FULL coverage (the reference's
// truncated grid misses rows 1072..1079 at 1080p: SURVEY.md quirk Q1).
__global__ __launch_bounds__(kBlock) void k_clear(uint4 *__restrict__ depth4, uint4 *__restrict__ acc4,
                                                  size_t n_depth4, size_t n_acc4, uint32_t *depth, size_t npix) {
    size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    size_t stride = (size_t)gridDim.x * kBlock;
    const uint4 e = make_uint4(RTR_EMPTY, RTR_EMPTY, RTR_EMPTY, RTR_EMPTY);
    const uint4 z = make_uint4(0, 0, 0, 0);
    for (size_t j = i; j < n_acc4; j += stride) acc4[j] = z;
    for (size_t j = i; j < n_depth4; j += stride) depth4[j] = e;
    for (size_t j = n_depth4 * 4 + i; j < npix; j += stride) depth[j] = RTR_EMPTY;
}

void launch_clear(hipStream_t s, uint32_t *depth, uint32_t *acc, size_t npix) {
    size_t n_acc4 = npix, n_depth4 = npix / 4;
    int grid = (int)((n_acc4 + kBlock - 1) / kBlock);
    if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(k_clear, dim3(grid), dim3(kBlock), 0, s, (uint4 *)depth, (uint4 *)acc, n_depth4, n_acc4, depth,
                       npix);
}

// ---------------------------------------------------------------------------------
// Point passes.  All of them stream the SoA cloud with 16-byte non-temporal loads (1 KiB
// per wave-instruction per coordinate; `nt` keeps the 1.2 GB stream from evicting the
// frame buffers out of L2 / Infinity Cache) and project four points per lane.
struct Quad {
    int pix[4];
    float d[4];
};

typedef float v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ld_stream(const float4 *p) {
    v4f v = __builtin_nontemporal_load(reinterpret_cast<const v4f *>(p));
    return make_float4(v.x, v.y, v.z, v.w);
}

__device__ __forceinline__ Quad project_quad(const float4 *__restrict__ x4, const float4 *__restrict__ y4,
                                             const float4 *__restrict__ z4, uint64_t i, const Proj &P, int W, int H,
                                             float fW, float fH) {
    float4 X = ld_stream(x4 + i), Y = ld_stream(y4 + i), Z = ld_stream(z4 + i);
    Quad q;
    q.pix[0] = project_point(P, X.x, Y.x, Z.x, W, H, fW, fH, q.d[0]);
    q.pix[1] = project_point(P, X.y, Y.y, Z.y, W, H, fW, fH, q.d[1]);
    q.pix[2] = project_point(P, X.z, Y.z, Z.z, W, H, fW, fH, q.d[2]);
    q.pix[3] = project_point(P, X.w, Y.w, Z.w, W, H, fW, fH, q.d[3]);
    return q;
}

__device__ __forceinline__ uint32_t ld_fresh(const uint32_t *p) {  // sc1: bypass the CU's L1
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

static int point_grid(uint64_t n4, int grid) {
    uint64_t blocks = (n4 + kBlock - 1) / kBlock;
    uint64_t cap = grid < 1 ? 1 : (uint64_t)grid;
    return (int)(blocks < cap ? blocks : cap);
}

// Wave-level aggregation of same-pixel atomics: what the reference does with
// __match_any_sync / __reduce_*_sync on 32-lane warps (render.cu:74-82,113-128), rebuilt for
// wave64 without a match instruction: up to four "leader" rounds (stopped as soon as a group has
// fewer than 8 lanes) -- take the first pending
// lane's pixel, ballot the lanes that share it, reduce their values with xor-shuffles, let the
// leader issue ONE atomic -- then whatever is left (incoherent input) goes out lane by lane.
// Must be called by all lanes of the wave (wave-uniform control flow).
__device__ __forceinline__ void wave_min(uint32_t *__restrict__ depth, int pix, bool valid, uint32_t bits) {
    unsigned long long todo = __ballot(valid);
    if (todo == 0ull) return;
    const int lane = threadIdx.x & 63;
#pragma unroll 1
    for (int round = 0; round < 4 && __popcll(todo) >= 8; ++round) {  // few pending lanes: not worth a round
        const int first = __ffsll((long long)todo) - 1;
        const int lead = __builtin_amdgcn_readlane(pix, first);
        const bool grp = valid && pix == lead;
        uint32_t v = grp ? bits : 0xFFFFFFFFu;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            uint32_t o = __shfl_xor(v, off, 64);
            v = o < v ? o : v;
        }
        if (lane == first) atomicMin(&depth[lead], v);  // render.cu:81
        const unsigned long long gm = __ballot(grp);
        todo &= ~gm;
        valid = valid && !grp;
        if (__popcll(gm) < 8) break;  // small groups: the input is not pixel-coherent, rounds do not pay
    }
    if (valid) atomicMin(&depth[pix], bits);
}

// colour: the group's (sum c0, sum c1, sum c2, count) fit one 64-bit word (<= 64 lanes x 255)
__device__ __forceinline__ void acc_add(uint32_t *__restrict__ acc, int pix, uint32_t c);
__device__ __forceinline__ void wave_acc(uint32_t *__restrict__ acc, int pix, bool valid, uint32_t c) {
    unsigned long long todo = __ballot(valid);
    if (todo == 0ull) return;
    const int lane = threadIdx.x & 63;
#pragma unroll 1
    for (int round = 0; round < 4 && __popcll(todo) >= 8; ++round) {  // few pending lanes: not worth a round
        const int first = __ffsll((long long)todo) - 1;
        const int lead = __builtin_amdgcn_readlane(pix, first);
        const bool grp = valid && pix == lead;
        unsigned long long v = grp ? ((unsigned long long)(c & 0xFFu) | ((unsigned long long)((c >> 8) & 0xFFu) << 16) |
                                      ((unsigned long long)((c >> 16) & 0xFFu) << 32) | (1ull << 48))
                                   : 0ull;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
        if (lane == first) {  // render.cu:125-128 as two 64-bit adds on the 4 x u32 layout
            unsigned long long *a = reinterpret_cast<unsigned long long *>(acc) + 2 * (size_t)lead;
            atomicAdd(a + 0, (v & 0xFFFFull) | (((v >> 16) & 0xFFFFull) << 32));
            atomicAdd(a + 1, ((v >> 32) & 0xFFFFull) | ((v >> 48) << 32));
        }
        const unsigned long long gm = __ballot(grp);
        todo &= ~gm;
        valid = valid && !grp;
        if (__popcll(gm) < 8) break;  // small groups: the input is not pixel-coherent, rounds do not pay
    }
    if (valid) acc_add(acc, pix, c);
}

// ---- mode 0: the reference's structure (two full streams + global atomics) -------------
// A4 minDepthPass (render.cu:53-83).  Semantics = "atomicMin of every surviving point";
// the reference's __match_any_sync aggregation is only a contention trick.  Early-z: the
// atomic is skipped unless the point is strictly closer than what an L1-bypassing load
// sees.  A stale value is >= the true minimum, so staleness can only cause a redundant
// atomic, never a wrong result.  The four loads of a lane are issued together (culled
// points read a dummy pixel) so their latencies overlap.
__global__ __launch_bounds__(kBlock) void k_min_depth(const float4 *__restrict__ x4, const float4 *__restrict__ y4,
                                                      const float4 *__restrict__ z4, uint64_t n4, Proj P, int W, int H,
                                                      uint32_t *__restrict__ depth) {
    const float fW = (float)W, fH = (float)H;
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    const int lane = threadIdx.x & 63;
    // wave-uniform trip count: every lane runs every iteration (the wave-level aggregation below
    // shuffles across all 64 lanes); lanes past the end re-read the wave's first quad, masked out
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i - (uint64_t)lane < n4; i += stride) {
        const bool live = i < n4;
        Quad q = project_quad(x4, y4, z4, live ? i : i - (uint64_t)lane, P, W, H, fW, fH);
        if (!live) q.pix[0] = q.pix[1] = q.pix[2] = q.pix[3] = -1;
        bool any = (q.pix[0] & q.pix[1] & q.pix[2] & q.pix[3]) >= 0;  // some sign bit clear
        if (__ballot(any) == 0ull) continue;                          // wave-uniform skip
        uint32_t cur[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) cur[k] = ld_fresh(depth + (q.pix[k] >= 0 ? q.pix[k] : lane));
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            uint32_t b = __float_as_uint(q.d[k]);
            wave_min(depth, q.pix[k], q.pix[k] >= 0 && b < cur[k], b);
        }
    }
}

void launch_min_depth(hipStream_t s, const Cloud &c, const Proj &P, int W, int H, uint32_t *depth) {
    uint64_t n4 = (c.n + 3) / 4;
    if (n4 == 0) return;
    hipLaunchKernelGGL(k_min_depth, dim3(point_grid(n4, c.grid)), dim3(kBlock), 0, s, (const float4 *)c.x, (const float4 *)c.y,
                       (const float4 *)c.z, n4, P, W, H, depth);
}

// A5 accumulatePass (render.cu:85-130): re-project, depth-window test against the (global)
// minimum, integer colour sums.  The four u32 accumulators of a pixel (sum c0, sum c1,
// sum c2, count; project_cloud.h:33) are updated with TWO 64-bit atomic adds on the same
// memory layout: (c0 | c1 << 32) and (c2 | count << 32).  Identical to four u32 adds unless
// a single channel sum passes 2^32, where the reference itself wraps (> 16.8 M points in one
// pixel's window).
__device__ __forceinline__ void acc_add(uint32_t *__restrict__ acc, int pix, uint32_t c) {
    unsigned long long *a = reinterpret_cast<unsigned long long *>(acc) + 2 * (size_t)pix;
    atomicAdd(a + 0, (unsigned long long)(c & 0xFFu) | ((unsigned long long)((c >> 8) & 0xFFu) << 32));
    atomicAdd(a + 1, (unsigned long long)((c >> 16) & 0xFFu) | (1ull << 32));
}

__global__ __launch_bounds__(kBlock) void k_accumulate(const float4 *__restrict__ x4, const float4 *__restrict__ y4,
                                                       const float4 *__restrict__ z4,
                                                       const uint32_t *__restrict__ rgba, uint64_t n4, Proj P, int W,
                                                       int H, const uint32_t *__restrict__ depth,
                                                       uint32_t *__restrict__ acc, float window) {
    const float fW = (float)W, fH = (float)H;
    const int lane = threadIdx.x & 63;
    uint64_t stride = (uint64_t)gridDim.x * kBlock;
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i - (uint64_t)lane < n4; i += stride) {
        const bool live = i < n4;  // wave-uniform trip count, see k_min_depth
        Quad q = project_quad(x4, y4, z4, live ? i : i - (uint64_t)lane, P, W, H, fW, fH);
        if (!live) q.pix[0] = q.pix[1] = q.pix[2] = q.pix[3] = -1;
        bool any = (q.pix[0] & q.pix[1] & q.pix[2] & q.pix[3]) >= 0;
        if (__ballot(any) == 0ull) continue;
        uint32_t m[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) m[k] = depth[q.pix[k] >= 0 ? q.pix[k] : lane];
        bool hit[4];
        uint32_t col[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {  // the four colour loads in flight together
            hit[k] = q.pix[k] >= 0 && !(q.d[k] > f_add(__uint_as_float(m[k]), window));  // render.cu:106
            col[k] = hit[k] ? rgba[4 * i + k] : 0u;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) wave_acc(acc, q.pix[k], hit[k], col[k]);
    }
}

void launch_accumulate(hipStream_t s, const Cloud &c, const Proj &P, int W, int H, const uint32_t *depth,
                       uint32_t *acc, float window) {
    uint64_t n4 = (c.n + 3) / 4;
    if (n4 == 0) return;
    hipLaunchKernelGGL(k_accumulate, dim3(point_grid(n4, c.grid)), dim3(kBlock), 0, s, (const float4 *)c.x, (const float4 *)c.y,
                       (const float4 *)c.z, c.rgba, n4, P, W, H, depth, acc, window);
}

// read-only probe: the same loads and projection arithmetic as the point passes but no
// frame-buffer traffic -- the streaming ceiling of this access pattern.
// Variants (experiments): 0 as the passes, 1 approximate reciprocal (VALU sensitivity),
// 2 two quads in flight per lane, 3 plain (temporal) loads, 4 loads only (no projection).
template <int V>
__global__ __launch_bounds__(kBlock) void k_stream_probe(const float4 *__restrict__ x4, const float4 *__restrict__ y4,
                                                         const float4 *__restrict__ z4, uint64_t n4, Proj P, int W,
                                                         int H, uint32_t *__restrict__ sink) {
    const float fW = (float)W, fH = (float)H;
    uint64_t stride = (uint64_t)gridDim.x * kBlock;
    uint32_t h = 0;
    if (V == 2) {
        uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
        for (; i + stride < n4; i += 2 * stride) {
            float4 X0 = ld_stream(x4 + i), Y0 = ld_stream(y4 + i), Z0 = ld_stream(z4 + i);
            float4 X1 = ld_stream(x4 + i + stride), Y1 = ld_stream(y4 + i + stride), Z1 = ld_stream(z4 + i + stride);
            float d;
            h += project_point(P, X0.x, Y0.x, Z0.x, W, H, fW, fH, d) ^ project_point(P, X0.y, Y0.y, Z0.y, W, H, fW, fH, d) ^
                 project_point(P, X0.z, Y0.z, Z0.z, W, H, fW, fH, d) ^ project_point(P, X0.w, Y0.w, Z0.w, W, H, fW, fH, d);
            h += project_point(P, X1.x, Y1.x, Z1.x, W, H, fW, fH, d) ^ project_point(P, X1.y, Y1.y, Z1.y, W, H, fW, fH, d) ^
                 project_point(P, X1.z, Y1.z, Z1.z, W, H, fW, fH, d) ^ project_point(P, X1.w, Y1.w, Z1.w, W, H, fW, fH, d);
        }
        if (i < n4) {
            Quad q = project_quad(x4, y4, z4, i, P, W, H, fW, fH);
            h += (uint32_t)(q.pix[0] ^ q.pix[1] ^ q.pix[2] ^ q.pix[3]);
        }
    } else {
        for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n4; i += stride) {
            if (V == 0) {
                Quad q = project_quad(x4, y4, z4, i, P, W, H, fW, fH);
                h += (uint32_t)(q.pix[0] ^ q.pix[1] ^ q.pix[2] ^ q.pix[3]);
            } else if (V == 3) {
                float4 X = x4[i], Y = y4[i], Z = z4[i];
                float d;
                h += project_point(P, X.x, Y.x, Z.x, W, H, fW, fH, d) ^ project_point(P, X.y, Y.y, Z.y, W, H, fW, fH, d) ^
                     project_point(P, X.z, Y.z, Z.z, W, H, fW, fH, d) ^ project_point(P, X.w, Y.w, Z.w, W, H, fW, fH, d);
            } else if (V == 4) {
                float4 X = ld_stream(x4 + i), Y = ld_stream(y4 + i), Z = ld_stream(z4 + i);
                h += __float_as_uint(X.x) ^ __float_as_uint(Y.y) ^ __float_as_uint(Z.z) ^ __float_as_uint(X.w) ^
                     __float_as_uint(Y.x) ^ __float_as_uint(Z.y);
            } else {  // V == 1: v_rcp_f32 instead of the IEEE divide (NOT contract-conforming)
                float4 X = ld_stream(x4 + i), Y = ld_stream(y4 + i), Z = ld_stream(z4 + i);
                const float xs[4] = {X.x, X.y, X.z, X.w}, ys[4] = {Y.x, Y.y, Y.z, Y.w}, zs[4] = {Z.x, Z.y, Z.z, Z.w};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    float rx = f_add(fmaf(P.m[2], zs[k], fmaf(P.m[1], ys[k], f_mul(P.m[0], xs[k]))), P.m[3]);
                    float ry = f_add(fmaf(P.m[6], zs[k], fmaf(P.m[5], ys[k], f_mul(P.m[4], xs[k]))), P.m[7]);
                    float rz = f_add(fmaf(P.m[10], zs[k], fmaf(P.m[9], ys[k], f_mul(P.m[8], xs[k]))), P.m[11]);
                    float inv = __builtin_amdgcn_rcpf(rz);
                    float fu = rintf(f_mul(rx, inv)), fv = rintf(f_mul(ry, inv));
                    bool ok = (rz > 0.0f) && (fu >= 0.0f) && (fu < fW) && (fv >= 0.0f) && (fv < fH);
                    h += ok ? (uint32_t)((int)fv * W + (int)fu) : 0xFFFFFFFFu;
                }
            }
        }
    }
    if (h == 0x12345678u) sink[0] = h;  // practically never; keeps the work alive
}

void launch_stream_probe(hipStream_t s, const Cloud &c, const Proj &P, int W, int H, uint32_t *sink, int variant) {
    uint64_t n4 = (c.n + 3) / 4;
    if (n4 == 0) return;
    dim3 g(point_grid(n4, c.grid)), b(kBlock);
    const float4 *x = (const float4 *)c.x, *y = (const float4 *)c.y, *z = (const float4 *)c.z;
    switch (variant) {
        case 1: hipLaunchKernelGGL(k_stream_probe<1>, g, b, 0, s, x, y, z, n4, P, W, H, sink); break;
        case 2: hipLaunchKernelGGL(k_stream_probe<2>, g, b, 0, s, x, y, z, n4, P, W, H, sink); break;
        case 3: hipLaunchKernelGGL(k_stream_probe<3>, g, b, 0, s, x, y, z, n4, P, W, H, sink); break;
        case 4: hipLaunchKernelGGL(k_stream_probe<4>, g, b, 0, s, x, y, z, n4, P, W, H, sink); break;
        default: hipLaunchKernelGGL(k_stream_probe<0>, g, b, 0, s, x, y, z, n4, P, W, H, sink); break;
    }
}

// =================================================================================
// mode 1 (default): tile-binned pipeline.
//
// Scattered device-scope atomics run at a few 10^10 per second on MI355X (they execute
// at the memory side), which is what bounds the atomic form above once the stream itself
// runs at ~6 TB/s.  The binned form has NO global atomics on the frame buffers:
//   T1 k_project_bin : stream the cloud once (12 B/pt); every in-frustum point becomes ONE
//                      8-byte entry (depth bits, in-tile pixel, colour) appended straight to the
//                      stream of its 32x16 "storage tile" -- one returning atomic per wave and
//                      tile claims the positions, so the entries are written once, already in
//                      tile order (round 1 wrote wave lists, scanned and counting-sorted them:
//                      three launches and 0.39 GB for 0.08 GB of entries).  The last workgroup
//                      to finish turns the stream lengths into the tile kernel's work list;
//   T4 k_tile<>      : one workgroup per 32x32 (64x32) tile keeps the tile's depth and
//                      accumulators in LDS: ds_min (render.cu:81), barrier, window test + ds_add
//                      (render.cu:106,125-128), barrier, resolve (render.cu:147-162) and
//                      writes every pixel of the tile -- clear, both reference passes and
//                      the resolve of one tile in one launch.  A tile with more than
//                      TileStore::heavy entries is split over several workgroups (slices of its
//                      streams, private LDS z-buffers): their minima meet in the depth buffer
//                      (atomicMin), a second launch accumulates the slices against that global
//                      minimum into the accumulators, and its last workgroup per tile resolves.
// Results are identical to the atomic form because min and integer sums commute.
constexpr int kTileH = 32;
#ifndef RTR_TILE_THREADS
#define RTR_TILE_THREADS 512
#define RTR_TILE_BATCH 8
#define RTR_TILE_WAVES 6
#endif
constexpr int kTileThreads = RTR_TILE_THREADS;
#ifndef RTR_TILE0_THREADS
#define RTR_TILE0_THREADS 256
#endif
constexpr int kTileThreadsCompact = RTR_TILE0_THREADS;  // k_tile<0> (see tile_body)
#ifndef RTR_TILE0_WAVES
#define RTR_TILE0_WAVES 8
#define RTR_TILE0_SWEEP 4
#endif
#ifndef RTR_TILE0_HEAD
#define RTR_TILE0_HEAD 1  // k_tile<0>: tiles beyond one batch keep their first batch in registers too (tile_body)
#endif
#ifndef RTR_TILE0_AHEAD
#define RTR_TILE0_AHEAD 1
#endif
#ifndef RTR_TILE0_BATCH
#define RTR_TILE0_BATCH 8
#endif
constexpr int kTileBatch = RTR_TILE_BATCH;    // entries in flight per thread in k_tile
#ifndef RTR_T1_RING
#define RTR_T1_RING 3  // the packed point kernel's LDS ring: chunks of A streams in flight per wave (k_project_bin)
#endif
#ifndef RTR_T1_LEAD
#define RTR_T1_LEAD 4  // ... and how many iterations a chunk's header is requested ahead of its A streams
#endif
#ifndef RTR_T1_WAVES
#define RTR_T1_WAVES 5  // the packed point kernel: five waves per SIMD (96 registers, four of them spilled to scratch in the
#endif                  // long path only; round 4: 126-129 us against 127-133 at four, six -- 80 registers, 51 spilled -- 180);
                        // the fp32 and the culling forms stay at four (their grid is four workgroups per CU anyway).
                        // (Capping it at 80 registers so that a tile workgroup of the previous frame fits beside it --
                        // option "overlap" -- spills in the hot loop: +25 us alone, measured 0.333 vs 0.296 ms.)
constexpr int kSplitGrid = 256; // workgroups of the mode-3 launch (they stride over the split tiles' slices)
constexpr int kPer2 = kTileBatch / 2, kPer4 = kTileBatch / 4;  // registers per stream of a 2- / 4-stream tile
constexpr int kMaxGroups = RTR_MAX_GROUPS;    // tile groups of a quad that get a wave-level claim; the rest claim per lane
constexpr int kMaxSegs = 4 * kDirK;
// Timing experiments (tools/kbench.py): `make experiment` builds librtr_hip_xp.so with RTR_EXPERIMENT,
// where option "xp" switches parts of T1 off (frames become wrong).  The shipped library has none of it.
#ifdef RTR_EXPERIMENT
#define RTR_XP(bit) ((xp & (bit)) != 0)
#define RTR_STAMP(S, slot) do { if (threadIdx.x == 0) ts_dbg(S)[slot] = wall_clock64(); } while (0)
#else
#define RTR_XP(bit) false
#define RTR_STAMP(S, slot) do { } while (0)
#endif

struct TileGeom {
    int tw_shift;  // log2(processing tile width): 5 (32x32) or 6 (64x32)
    int tiles_x, tiles_y, ntiles;
    int stx, sty, nst;  // 32x16 storage tiles
};

__host__ __device__ inline TileGeom tile_geom(int W, int H) {
    TileGeom g;
    g.tw_shift = 5;
    g.tiles_x = (W + 31) >> 5;
    g.tiles_y = (H + kTileH - 1) / kTileH;
    if (g.tiles_x * g.tiles_y > 4096) {  // work-list items carry 12 tile bits and the peer-to-peer occupancy
                                         // bitmap 4096 bits (4K frames); the host falls back to the atomic
                                         // form when even 64-wide tiles exceed 4096
        g.tw_shift = 6;
        g.tiles_x = (W + 63) >> 6;
    }
    g.ntiles = g.tiles_x * g.tiles_y;
    g.stx = (W + 31) >> 5;
    g.sty = (H + 15) >> 4;
    g.nst = g.stx * g.sty;
    return g;
}

int tile_count(int W, int H) { return tile_geom(W, H).ntiles; }
int storage_tile_count(int W, int H) { return tile_geom(W, H).nst; }

// entry = depth bits (31: positive floats) << 33 | pixel inside the 32x16 storage tile (9) << 24 | colour (24)
__device__ __forceinline__ unsigned long long make_entry(uint32_t depth_bits, uint32_t pix9, uint32_t colour) {
    return ((unsigned long long)depth_bits << 33) | ((unsigned long long)pix9 << 24) | (unsigned long long)(colour & 0xFFFFFFu);
}

__device__ __forceinline__ void store_error(const TileStore &S, uint32_t code) { atomicOr(ts_hdr(S) + kHdrErrLive, code); }
// ... from the TILE kernels: T1's epilogue has already published this frame's word, so the code also goes straight to the
// mapped host word (a plain store, like the epilogue's: sticky until the host reads it) -- the synchronising call that
// returns this frame must not return RTR_OK
__device__ __forceinline__ void store_error_now(const TileStore &S, uint32_t code) {
    store_error(S, code);
    typedef uint32_t __attribute__((address_space(1))) *gu32_t;
    uint32_t *const host = ts_consts(S)->err_host;
    if (host) *(volatile gu32_t)host = code;
}

// Stream position v >= kS0 of storage tile st lies in extent k, which holds [kS0 << (k-1), kS0 << k).
// The lane that claimed an extent's FIRST position allocates it (one returning add on the pool
// cursor) and publishes base | stamp as one 8-byte word; every other lane polls that word.
// A wave discharges ALL its allocation duties (extent_alloc over its four points) before any of
// its lanes polls (extent_slot): an allocator then never waits for anything, so the polls of all
// waves terminate -- polling for a peer's extent on one point while still owing an allocation on
// a later one is a circular wait between two waves.  The polls are bounded all the same.
__device__ __forceinline__ unsigned long long extent_alloc(const TileStore &S, uint32_t st, uint32_t v) {
    const int k = 32 - __clz((int)(v >> kS0Shift));  // 1..20
    const uint32_t start = kS0 << (k - 1);           // first position = size of extent k
    if (v != start) return 0ull;
    const unsigned long long base = atomicAdd(ts_pool(S), (unsigned long long)start);
    const unsigned long long e = (base << 24) | (unsigned long long)S.seq;
    __hip_atomic_store(ts_dir(S) + (size_t)st * kDirK + k, e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return e;
}
__device__ __forceinline__ uint64_t *extent_slot(const TileStore &S, uint32_t st, uint32_t v, unsigned long long own) {
    const int k = 32 - __clz((int)(v >> kS0Shift));
    const uint32_t start = kS0 << (k - 1);
    unsigned long long e = own;
    if (e == 0ull) {
        const unsigned long long *d = ts_dir(S) + (size_t)st * kDirK + k;
        int polls = 0;
        for (;;) {
            e = __hip_atomic_load(d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((uint32_t)(e & 0xFFFFFFull) == S.seq) break;
            if (++polls > (1 << 22)) {
                store_error(S, 1u);
                return nullptr;
            }
            __builtin_amdgcn_s_sleep(4);
        }
    }
    const unsigned long long base = e >> 24;
    const StoreConsts *sc = ts_consts(S);
    if (base + start > sc->dyn_cap) {  // cannot happen: the extents of a frame sum to < 2 x its entries
        store_error(S, 2u);
        return nullptr;
    }
    return sc->dyn + base + (v - start);
}

// inclusive scan over the workgroup (<= 8 waves); returns the inclusive prefix, `total` = sum of all
__device__ __forceinline__ uint32_t block_scan(uint32_t v, uint32_t *s_w /*[8]*/, uint32_t &total) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t o = __shfl_up(v, off, 64);
        if (lane >= off) v += o;
    }
    __syncthreads();  // s_w may still be read from the previous scan
    if (lane == 63) s_w[wv] = v;
    __syncthreads();
    uint32_t base = 0;
    total = 0;
    for (int k = 0; k < nw; ++k) {
        const uint32_t w = s_w[k];
        base += k < wv ? w : 0u;
        total += w;
    }
    return v + base;
}
__device__ __forceinline__ uint32_t block_max(uint32_t v, uint32_t *s_w) {
    const int nw = blockDim.x >> 6;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const uint32_t o = __shfl_xor(v, off, 64);
        v = o > v ? o : v;
    }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = v;
    __syncthreads();
    uint32_t m = s_w[0];
    for (int k = 1; k < nw; ++k) m = s_w[k] > m ? s_w[k] : m;
    return m;
}

// storage tile of stream s (0 .. 2 or 4) of processing tile (tx, ty); -1 when outside the frame
__device__ __forceinline__ int stream_tile(const TileGeom &g, int tx, int ty, int s) {
    const int per_row = 1 << (g.tw_shift - 5);
    const int sx = tx * per_row + (s & (per_row - 1)), sy = ty * 2 + (s >> (g.tw_shift - 5));
    return (sx < g.stx && sy < g.sty) ? sy * g.stx + sx : -1;
}

// The bookkeeping of a binned frame, run by the LAST workgroup of T1 (every stream length is final:
// each wave waited for its own returning adds before its workgroup took a ticket).  It runs alone on
// the chip, so it is kept to ONE round of loads and no scan in the usual case:
//   fill[] <- 0, entries per processing tile, one work-list record (with the stream lengths) per tile at the
//   position perm[tile] (the launch order of the tile kernel: heavy tiles of the previous frame
//   first), frame statistics, the occupancy bitmap of the peer-to-peer exchange, pool / ticket reset.
// Only when some tile exceeds the split threshold are its slices laid out (scans) and, if asked for,
// its pixels reset.
// flags: bit 0 = reset the pixels of the tiles that will be split (clear_split), bit 1 = split NO tile in this frame:
// the host does not launch k_tile_split behind it -- it skips that launch while the frames it has seen complete had no
// tile above the threshold (sc.split_host tells it, without a sync) -- so a tile that is above it after all is
// processed by its one workgroup: slow, exact, and reported (next_frame_order), which brings the split launch back a
// few frames later.
__device__ void bin_epilogue(const TileStore &S, int W, int H, int flags, uint32_t colour_chunks) {
    const int clear_split = flags & 1;
    const bool no_split = (flags & 2) != 0;
    const TileGeom g = tile_geom(W, H);
    uint32_t *const fill = ts_fill(S), *const tile_cnt = ts_tile_cnt(S);
    uint32_t *const hdr = ts_hdr(S), *const hctr = ts_hctr(S);
    const uint32_t *const perm = ts_perm(S);
    uint4 *const records = reinterpret_cast<uint4 *>(ts_items(S));
    uint4 *const cnt4 = ts_cnt4(S);
    RTR_STAMP(S, 1);
    const StoreConsts sc = *ts_consts(S);
    typedef uint32_t __attribute__((address_space(1))) *gu32_t;  // (pointers out of memory: global, not flat)
    const gu32_t depth = (gu32_t)sc.depth, acc = (gu32_t)sc.acc, occ = (gu32_t)sc.occ;
    const uint32_t heavy = no_split ? 0xFFFFFFFFu : sc.heavy;
#ifdef RTR_EXPERIMENT
    if (threadIdx.x == 0 && heavy != 7u) ts_dbg(S)[5] = wall_clock64();  // consts have arrived
#endif
    __shared__ uint32_t s_w[8];
    __shared__ uint32_t s_occ[128];
    __shared__ uint32_t s_split[64];  // tiles to reset (more are reset by a second sweep)
    __shared__ uint32_t s_nsplit, s_total, s_max;
    const int t = threadIdx.x;
    if (t < 128) s_occ[t] = 0;
    if (t == 0) s_nsplit = s_total = s_max = 0;
    __syncthreads();
    const int wide = g.tw_shift - 5;  // 0: two streams per tile (32 x 32), 1: four (64 x 32)
    uint32_t sum = 0, mx = 0, heavy_n = 0;
    auto load_lengths = [&](auto ns_tag, auto batch_tag) {  // thread t owns tiles t, t + 256, ...: coalesced
        constexpr int NS = decltype(ns_tag)::value, BATCH = decltype(batch_tag)::value;
#pragma unroll 1
        for (int k0 = 0; k0 * kBlock < g.ntiles; k0 += BATCH) {
            uint32_t f[BATCH][NS], pos[BATCH];
            int stv[BATCH][NS];
#pragma unroll
            for (int k = 0; k < BATCH; ++k) {  // every load of the batch is issued before the first one is waited for
                const int tile = (k0 + k) * kBlock + t;
                const int tx = tile % g.tiles_x, ty = tile / g.tiles_x;
                // (unconditional loads from clamped indices: a load whose result merges with a constant at the
                // end of a branch is waited for right there, which serialised the batch: 10 us instead of 2)
                pos[k] = perm[tile < g.ntiles ? tile : 0];
#pragma unroll
                for (int s = 0; s < NS; ++s) {
                    stv[k][s] = tile < g.ntiles ? stream_tile(g, tx, ty, s) : -1;
                    const uint32_t got = __hip_atomic_load(fill + ((size_t)(stv[k][s] >= 0 ? stv[k][s] : 0) << S.fill_shift),
                                                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    f[k][s] = stv[k][s] >= 0 ? got : 0u;
                }
            }
#ifdef RTR_EXPERIMENT
            if (threadIdx.x == 0 && k0 == 0 && (f[0][0] | pos[0]) != 0xFFFFFFFFu) ts_dbg(S)[6] = wall_clock64();  // first loads back
#endif
#pragma unroll
            for (int k = 0; k < BATCH; ++k) {
                const int tile = (k0 + k) * kBlock + t;
                uint32_t c = 0;
#pragma unroll
                for (int s = 0; s < NS; ++s)
                    if (stv[k][s] >= 0) {
                        fill[(size_t)stv[k][s] << S.fill_shift] = 0;
                        c += f[k][s];
                    }
                if (tile < g.ntiles) {
                    tile_cnt[tile] = c;
                    records[2 * (size_t)pos[k]] = make_uint4(c > heavy ? kItemSkip : (uint32_t)tile, f[k][0], f[k][1], NS > 2 ? f[k][NS > 2 ? 2 : 0] : 0u);
                    records[2 * (size_t)pos[k] + 1] = make_uint4(NS > 2 ? f[k][NS > 2 ? 3 : 0] : 0u, 0u, 0u, 0u);
                    if (c) atomicOr(&s_occ[tile >> 5], 1u << (tile & 31));
                    if (occ) cnt4[tile] = make_uint4(f[k][0], f[k][1], NS > 2 ? f[k][NS > 2 ? 2 : 0] : 0u, NS > 2 ? f[k][NS > 2 ? 3 : 0] : 0u);  // (sharded frames)
                }
                sum += c;
                mx = c > mx ? c : mx;
                heavy_n += c > heavy ? 1u : 0u;
            }
        }
    };
    if (wide)
        load_lengths(std::integral_constant<int, 4>{}, std::integral_constant<int, 4>{});
    else
        load_lengths(std::integral_constant<int, 2>{}, std::integral_constant<int, 8>{});
    RTR_STAMP(S, 2);
    // statistics (and the split decision) through LDS atomics: one barrier instead of a scan each
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        sum += __shfl_xor(sum, off, 64);
        const uint32_t o = __shfl_xor(mx, off, 64);
        mx = o > mx ? o : mx;
    }
    if ((t & 63) == 0) {
        atomicAdd(&s_total, sum);
        atomicMax(&s_max, mx);
    }
    const int any_heavy = __syncthreads_or(heavy_n != 0u);
    RTR_STAMP(S, 3);
    const uint32_t total = s_total;
    mx = s_max;
    if (occ && t < 128) occ[t] = s_occ[t];
    uint32_t n_split_items = 0, n_heavy = 0, slice = sc.slice < 1u ? 1u : sc.slice;
    if (any_heavy) {  // workgroup-uniform, rare: lay out the slices of the tiles above the split threshold
        uint32_t heavy_sum = 0;
#pragma unroll 1
        for (int tile = t; tile < g.ntiles; tile += kBlock) {
            const uint32_t c = tile_cnt[tile];  // written above by this same thread
            heavy_sum += c > heavy ? c : 0u;    // (a frame has < 2^32 entries)
        }
        uint32_t heavy_total = 0;
        block_scan(heavy_sum, s_w, heavy_total);
        block_scan(heavy_n, s_w, n_heavy);
        // slice size: sum of ceil(cnt / slice) over the split tiles <= heavy_total / slice + n_heavy, and the
        // split region of the list has room for ntiles + kHeavyExtra records
        {
            const uint32_t need = (uint32_t)(((unsigned long long)heavy_total + kHeavyExtra - 1) / kHeavyExtra);
            slice = need > slice ? need : slice;
            const uint32_t need2 = (mx + 1023u) / 1024u;  // nsub - 1 has 10 bits
            slice = need2 > slice ? need2 : slice;
        }
        uint32_t my_sub = 0;
#pragma unroll 1
        for (int tile = t; tile < g.ntiles; tile += kBlock) {
            const uint32_t c = tile_cnt[tile];
            if (c > heavy) my_sub += (c + slice - 1) / slice;
        }
        uint32_t pos = block_scan(my_sub, s_w, n_split_items) - my_sub;
#pragma unroll 1
        for (int tile = t; tile < g.ntiles; tile += kBlock) {
            const uint32_t c = tile_cnt[tile];
            if (c > heavy) {
                const uint32_t nsub = (c + slice - 1) / slice;
                const uint4 own0 = records[2 * (size_t)perm[tile]], own1 = records[2 * (size_t)perm[tile] + 1];  // written above by this same thread
                for (uint32_t j = 0; j < nsub; ++j) {
                    uint4 *rec = records + 2 * (size_t)((uint32_t)g.ntiles + pos + j);
                    rec[0] = make_uint4((uint32_t)tile | (j << 12) | ((nsub - 1u) << 22), own0.y, own0.z, own0.w);
                    rec[1] = own1;
                }
                pos += nsub;
                hctr[tile] = 0;
                const uint32_t q = atomicAdd(&s_nsplit, 1u);
                if (q < 64u) s_split[q] = (uint32_t)tile;
            }
        }
        __syncthreads();
    }
    if (t == 0) {
        hdr[kHdrItems] = (uint32_t)g.ntiles - n_heavy + n_split_items;
        hdr[kHdrSplitItems] = n_split_items;
        hdr[kHdrEntries] = total;
        if (sc.entries_host) *(volatile gu32_t)sc.entries_host = total;  // (the host sizes the extent pool by it)
        hdr[kHdrHeaviest] = mx;
        hdr[kHdrSlice] = slice;
        hdr[kHdrSplitTiles] = n_heavy;
        hdr[kHdrColourChunks] = colour_chunks;
        hdr[kHdrSplitQ1] = 0u, hdr[kHdrSplitDone] = 0u, hdr[kHdrSplitQ2] = 0u;
        // the frame's tile-store error (entries were dropped: the frame is wrong) becomes the header's published
        // word and reaches the host through mapped memory -- rtr_synchronize and the calls that copy results to
        // the host report it; the live word starts the next frame at zero
        const uint32_t err = __hip_atomic_load(hdr + kHdrErrLive, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        hdr[kHdrError] = err;
        if (err) {
            hdr[kHdrErrLive] = 0u;
            if (sc.err_host) *(volatile gu32_t)sc.err_host = err;  // (plain store: the word lives in host memory; sticky until the host reads it)
        }
        *ts_pool(S) = 0ull;
        *reinterpret_cast<unsigned long long *>(ts_ticket(S)) = 0ull;
    }
    RTR_STAMP(S, 4);
    // Whole frames (and sharded frames whose tile launches are the only writers) never clear the frame
    // buffers: an unsplit tile is written by its one workgroup.  The slices of a split tile meet in
    // memory (atomicMin / atomicAdd), which therefore has to start from the sentinel / zero.
    if (clear_split && n_heavy) {
        const int tw = 1 << g.tw_shift, tpix = 32 << g.tw_shift;
        const uint32_t listed = s_nsplit < 64u ? s_nsplit : 64u;
        auto reset_tile = [&](int tile) {
            const int tx0 = (tile % g.tiles_x) << g.tw_shift, ty0 = (tile / g.tiles_x) * kTileH;
            for (int p = t; p < tpix; p += kBlock) {
                const int x = tx0 + (p & (tw - 1)), y = ty0 + (p >> g.tw_shift);
                if (x < W && y < H) {
                    const size_t gp = (size_t)y * W + x;
                    depth[gp] = RTR_EMPTY;
                    acc[4 * gp] = 0u, acc[4 * gp + 1] = 0u, acc[4 * gp + 2] = 0u, acc[4 * gp + 3] = 0u;
                }
            }
        };
        if (n_heavy <= 64u) {
            for (uint32_t q = 0; q < listed; ++q) reset_tile((int)s_split[q]);
        } else {
            for (int tile = 0; tile < g.ntiles; ++tile)
                if (tile_cnt[tile] > heavy) reset_tile(tile);  // written by this workgroup above (barrier passed)
        }
    }
}

// The launch order of the NEXT frame's tile kernel, from this frame's entry counts (consecutive frames
// look alike): tiles with more than twice the mean entry count first, so the few heavy tiles that
// bound T4 start at once instead of trailing the launch.  One extra workgroup of the tile launch (modes 0
// and 1), beside the ~2000 that are busy with tiles -- off every critical path.
__device__ void next_frame_order(const TileStore &S, int parity, bool lean = false) {
    __shared__ uint32_t s_w[8];
    // (a lean frame orders by the PREVIOUS lean frame's counts: its own are being written by the tile workgroups right now)
    const uint32_t *const tile_cnt = lean ? ts_lcnt(S, parity ^ 1) : ts_tile_cnt(S);
    // (the launch of parity p writes order[p ^ 1]: the tile workgroups of a lean frame may still be reading order[p])
    uint32_t *const perm = ts_perm(S), *const order = ts_order(S, parity ^ 1);
    const int t = threadIdx.x, nt = S.ntiles, step = blockDim.x;
    const uint32_t thr = 2u * (ts_hdr(S)[kHdrEntries] / (uint32_t)nt) + 1u;
    const StoreConsts *const sc = ts_consts(S);
    const uint32_t heavy = sc->heavy;
    uint32_t big = 0, all = 0, over = 0;
    const int per = (nt + step - 1) / step, lo = t * per;  // contiguous tiles per thread: positions stay tile-ordered
    // (every count is read once; per <= 32: nt <= 4096)
    uint32_t big_mask = 0;
    for (int k = 0; k < per && k < 32; ++k)
        if (lo + k < nt) {
            const uint32_t c = tile_cnt[lo + k];
            all += 1;
            big += c > thr ? 1u : 0u;
            big_mask |= (c > thr ? 1u : 0u) << k;
            over += c > heavy ? 1u : 0u;
        }
    // Tiles above the split threshold in this frame -> a mapped host word: the host launches k_tile_split behind a
    // whole frame only while this has been non-zero lately (rtr_ctx::split_host; read without a sync).
    if (!lean) {  // (workgroup-uniform; a lean frame reports through lean_fold)
        typedef uint32_t __attribute__((address_space(1))) *gu32_t;
        const int any_over = __syncthreads_or(over != 0u);
        if (t == 0 && sc->split_host) *(volatile gu32_t)sc->split_host = any_over ? 1u : 0u;
    }
    uint32_t n_big = 0, n_all = 0;
    uint32_t big_before = block_scan(big, s_w, n_big) - big;
    uint32_t all_before = block_scan(all, s_w, n_all) - all;
    for (int k = 0; k < per && k < 32; ++k)
        if (lo + k < nt) {
            const bool b = (big_mask >> k) & 1u;
            const uint32_t pos = b ? big_before : n_big + (all_before - big_before);
            perm[lo + k] = pos;
            order[pos] = (uint32_t)(lo + k);
            big_before += b ? 1u : 0u;
            all_before += 1u;
        }
}

// Lean frames (rtr_kernels.h, ts_off_order).  lean_fold: the per-tile entry counts the tile workgroups of the lean frame
// of parity q stored -> the header words rtr_frame_stats reports and T1 / the host steer by; run by the NEXT lean frame's
// extra workgroup (the frame is complete then) or by rtr_frame_stats.  One workgroup; s_w: 3 x 8 words of LDS.
__device__ void lean_fold(const TileStore &S, int q, uint32_t *s_w /*[24]*/) {
    uint32_t *const flag = ts_lflag(S) + q;
    if (*flag == 0u) return;  // (workgroup-uniform: nothing to fold, or folded already)
    const uint32_t *const cnt = ts_lcnt(S, q);
    const StoreConsts *const sc = ts_consts(S);
    const uint32_t heavy = sc->heavy;
    uint32_t sum = 0, mx = 0, over = 0;
    for (int tile = threadIdx.x; tile < S.ntiles; tile += blockDim.x) {
        const uint32_t c = cnt[tile];
        sum += c;
        mx = c > mx ? c : mx;
        over += c > heavy ? 1u : 0u;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        sum += __shfl_xor(sum, off, 64);
        over += __shfl_xor(over, off, 64);
        const uint32_t o = __shfl_xor(mx, off, 64);
        mx = o > mx ? o : mx;
    }
    const int wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) s_w[wv] = sum, s_w[8 + wv] = mx, s_w[16 + wv] = over;
    __syncthreads();
    if (threadIdx.x == 0) {
        sum = mx = over = 0;
        for (int k = 0; k < nw; ++k) {
            sum += s_w[k];
            mx = s_w[8 + k] > mx ? s_w[8 + k] : mx;
            over += s_w[16 + k];
        }
        uint32_t *const hdr = ts_hdr(S);
        hdr[kHdrItems] = (uint32_t)S.ntiles;
        hdr[kHdrSplitItems] = 0u;
        hdr[kHdrEntries] = sum;
        hdr[kHdrHeaviest] = mx;
        hdr[kHdrSlice] = sc->slice < 1u ? 1u : sc->slice;
        hdr[kHdrSplitTiles] = 0u;  // (a lean frame splits nothing: tiles above the threshold were done by one workgroup each)
        typedef uint32_t __attribute__((address_space(1))) *gu32_t;
        if (sc->split_host) *(volatile gu32_t)sc->split_host = over ? 1u : 0u;  // ... and reported: the split launch comes back
        if (sc->entries_host) *(volatile gu32_t)sc->entries_host = sum;
        *flag = 0u;
    }
}
// The extra workgroup of a lean frame's tile launch: what T1's epilogue does for the other frames, minus everything
// the tile workgroups now do for themselves.  T1 of this frame is complete (kernel boundary), the next T1 has not begun.
__device__ void lean_frame_end(const TileStore &S, int parity) {
    uint32_t *const hdr = ts_hdr(S);
    const int t = threadIdx.x;
    if (t < 64) {  // colour-chunk counts of this frame's T1 (one word per sub-ticket line)
        unsigned long long cc = 0ull;
        if (t < kSubTickets) {
            cc = *ts_sub_colour(S, (uint32_t)t);
            *ts_sub_colour(S, (uint32_t)t) = 0ull;
        }
        uint32_t c32 = (uint32_t)cc;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) c32 += __shfl_xor(c32, off, 64);
        if (t == 0) {
            hdr[kHdrColourChunks] = c32;
            const StoreConsts *const sc = ts_consts(S);
            const uint32_t err = __hip_atomic_load(hdr + kHdrErrLive, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            hdr[kHdrError] = err;
            if (err) {
                typedef uint32_t __attribute__((address_space(1))) *gu32_t;
                hdr[kHdrErrLive] = 0u;
                if (sc->err_host) *(volatile gu32_t)sc->err_host = err;
            }
            *ts_pool(S) = 0ull;
            ts_lflag(S)[parity] = 1u;  // (this frame's counts are on their way: the next fold takes them)
        }
    }
    __shared__ uint32_t s_fold[24];
    lean_fold(S, parity ^ 1, s_fold);
    __syncthreads();
    next_frame_order(S, parity, true);
}
__global__ __launch_bounds__(kBlock) void k_lean_fold(TileStore S, int parity) {
    __shared__ uint32_t s_fold[24];
    lean_fold(S, parity, s_fold);
}
void launch_lean_fold(hipStream_t s, int W, int H, const TileStore &S, int parity) {
    (void)W, (void)H;
    hipLaunchKernelGGL(k_lean_fold, dim3(1), dim3(kBlock), 0, s, S, parity);
}

// PackedXyz helpers ---------------------------------------------------------------
// An axis block is TWO little-endian bit streams (rtr_kernels.h): the FIRST value of every lane -- lane l's b bits at bit
// b l of the A stream, 8 b bytes -- and its other three -- 3 b bits at bit 3 b l of the B stream, 24 b bytes.  A lane
// reads the 8 (16) bytes that start at the DWORD holding its first bit (loads whose lane stride is not a multiple of
// four bytes run at a third of the rate: 2.2-3.8 TB/s against 7.0, tools/align_probe.hip) and shifts its data down by
// the remaining 0..31 bits; b <= 25 keeps shift + b <= 64 and shift + 3 b <= 128.  A fixed number of loads per chunk,
// no branch around any of them; both streams end with spare bytes for the last lane's over-read.
typedef uint32_t u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));
typedef uint32_t u32x2_a4 __attribute__((ext_vector_type(2), aligned(4)));
struct AxisRawA { uint32_t d[2]; };
struct AxisRaw { uint32_t d[4]; };
__device__ __forceinline__ AxisRawA ld_axis_a(const uint8_t *block, uint32_t b, int lane) {
    const uint32_t dw = (b * (uint32_t)lane) >> 5;
    const u32x2_a4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x2_a4 *>(block + 4u * dw));
    return AxisRawA{{v.x, v.y}};
}
__device__ __forceinline__ AxisRaw ld_axis_b(const uint8_t *block, uint32_t b, int lane) {
    const uint32_t dw = (3u * b * (uint32_t)lane) >> 5;
    const u32x4_a4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4_a4 *>(block + 4u * dw));
    return AxisRaw{{v.x, v.y, v.z, v.w}};
}
// value 0 = base | the lane's b bits of the A stream, value k = base | bits [b (k - 1), b k) of its realigned B data.
// Branch-free for every b <= 25 (b = 0: the mask is empty and the value is the base).  A first version picked the dwords
// a value straddles by width class behind wave-uniform branches: ~10 branches per axis made the point kernel 14 us
// slower than the byte-granular form it was meant to beat.  b is wave-uniform.
__device__ __forceinline__ float4 unpack_axis_narrow(const AxisRawA &ra, const AxisRaw &r, uint32_t b, uint32_t base, int lane) {
    // (a VOP3 instruction reads at most one scalar register on gfx950: with mask AND base scalar the compiler splits every
    // v_and_or into two instructions; the base in a vector register keeps it one)
    uint32_t vbase = base;
    asm("" : "+v"(vbase));
    uint32_t sha, shb;  // (b l) & 31, (3 b l) & 31: alignbit takes the low five bits of its shift
    asm("v_mul_u32_u24 %0, %1, %2" : "=v"(sha) : "s"(b), "v"(lane));
    asm("v_mul_u32_u24 %0, %1, %2" : "=v"(shb) : "s"(3u * b), "v"(lane));
    const uint32_t mask = (1u << b) - 1u;  // (b <= 25)
    const uint32_t a0 = __builtin_amdgcn_alignbit(ra.d[1], ra.d[0], sha);
    const uint32_t e0 = __builtin_amdgcn_alignbit(r.d[1], r.d[0], shb), e1 = __builtin_amdgcn_alignbit(r.d[2], r.d[1], shb);
    const uint32_t e2 = __builtin_amdgcn_alignbit(r.d[3], r.d[2], shb);
    const uint32_t f0 = __builtin_amdgcn_alignbit(e1, e0, b), f1 = __builtin_amdgcn_alignbit(e2, e1, b);
    const uint32_t g0 = __builtin_amdgcn_alignbit(f1, f0, b);
    auto and_or = [&](uint32_t e) -> float {  // (the compiler leaves v_and + v_or here)
        uint32_t x;
        asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(x) : "v"(e), "s"(mask), "v"(vbase));
        return __uint_as_float(x);
    };
    return make_float4(and_or(a0), and_or(e0), and_or(f0), and_or(g0));
}
__device__ __forceinline__ float4 unpack_axis(const AxisRawA &ra, const AxisRaw &r, uint32_t b, uint32_t base, int lane) {
    if (b == 32u)  // (lane l's first value is dword l of the A stream, its other three dwords 3 l .. 3 l + 2 of the B stream)
        return make_float4(__uint_as_float(ra.d[0]), __uint_as_float(r.d[0]), __uint_as_float(r.d[1]), __uint_as_float(r.d[2]));
    return unpack_axis_narrow(ra, r, b, base, lane);
}
struct ChunkRawA { AxisRawA a[3]; };
struct ChunkRaw { AxisRaw a[3]; };
__device__ __forceinline__ ChunkRawA load_chunk_a(const uint32_t *__restrict__ planes_a, const uint4 &h0, const uint4 &h1, int lane) {
    const uint8_t *p = reinterpret_cast<const uint8_t *>(planes_a) + (((((uint64_t)h1.y) << 32) | (uint64_t)h1.x) << 3);
    ChunkRawA c;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const uint32_t b = (h0.w >> (6 * a)) & 63u;
        c.a[a] = ld_axis_a(p, b, lane);
        p += 8u * b;
    }
    return c;
}
__device__ __forceinline__ ChunkRaw load_chunk_b(const uint32_t *__restrict__ planes_b, const uint4 &h0, const uint4 &h1, int lane) {
    const uint8_t *p = reinterpret_cast<const uint8_t *>(planes_b) + (((((uint64_t)h1.y) << 32) | (uint64_t)h1.x) * 24u);
    ChunkRaw c;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const uint32_t b = (h0.w >> (6 * a)) & 63u;
        c.a[a] = ld_axis_b(p, b, lane);
        p += 24u * b;
    }
    return c;
}
__device__ __forceinline__ void unpack_chunk(const ChunkRawA &ca, const ChunkRaw &c, uint32_t widths, uint32_t bx, uint32_t by, uint32_t bz,
                                             float4 &X, float4 &Y, float4 &Z, int lane) {
    if (!(widths & kPackWideFlag)) {  // no axis of the chunk needs all 32 bits (the usual case)
        // (a constant axis -- a wall of the synthetic room, 30 % of its axis blocks -- skips the fifteen instructions)
        const uint32_t wx = widths & 63u, wy = (widths >> 6) & 63u, wz = (widths >> 12) & 63u;
        X = wx ? unpack_axis_narrow(ca.a[0], c.a[0], wx, bx, lane) : make_float4(__uint_as_float(bx), __uint_as_float(bx), __uint_as_float(bx), __uint_as_float(bx));
        Y = wy ? unpack_axis_narrow(ca.a[1], c.a[1], wy, by, lane) : make_float4(__uint_as_float(by), __uint_as_float(by), __uint_as_float(by), __uint_as_float(by));
        Z = wz ? unpack_axis_narrow(ca.a[2], c.a[2], wz, bz, lane) : make_float4(__uint_as_float(bz), __uint_as_float(bz), __uint_as_float(bz), __uint_as_float(bz));
    } else {
        X = unpack_axis(ca.a[0], c.a[0], widths & 63u, bx, lane);
        Y = unpack_axis(ca.a[1], c.a[1], (widths >> 6) & 63u, by, lane);
        Z = unpack_axis(ca.a[2], c.a[2], (widths >> 12) & 63u, bz, lane);
    }
}

// T1 ------------------------------------------------------------------------------
// VALU matters here (a loads-only probe streams at 6.9 TB/s, with the projection arithmetic
// at 6.0), so the quad is culled in three wave-uniform steps before the expensive part:
//   1. r.z of the four points (render.cu:37,63); skip the quad if no lane has r.z > 0;
//   2. r.x, r.y and a CONSERVATIVE frustum test (the exact test needs the quotient): a
//      point is certainly outside when r.x < -0.75 r.z or r.x > (W + 0.25) r.z (same in y):
//      the rounded quotient differs from r.x / r.z by < 2^-21 relative, i.e. < 0.01 px for
//      W <= 8192, against margins of 0.25 px; only applied for r.z > 1e-30 so that no
//      underflow enters the argument.  Skip the quad if no lane may be inside;
//   3. the exact contract arithmetic (reciprocal, rintf, range test) for what is left.
// Spatially coherent clouds (LiDAR block order) take the early exits for ~90 % of the waves.
//
// Append: the in-frustum points of a quad (up to 256 per wave) are grouped by storage tile with
// ballots -- consecutive points are spatial neighbours, so one to three tiles cover them -- and
// each group claims its run of stream positions with ONE returning atomic add issued by one lane;
// all claims of a quad, and the 16-byte colour load of the lanes that need it, are in flight
// together and waited for once.  Points of further tiles (incoherent clouds) claim per lane.
//
// CULL (option "cull", off by default, reported separately from the roofline figure): each
// wave first tests the bounding box of its 256-point chunk (k_chunk_bounds, 24 B per chunk)
// against the five frustum half-spaces r.z >= 0, r.x + r.z >= 0, W r.z - r.x >= 0,
// r.y + r.z >= 0, H r.z - r.y >= 0 -- supersets of the exact acceptance region (a kept point
// has its fp32 quotient in [-0.5, W - 0.5], up to 2^-22 relative) -- and skips the chunk,
// loads included, when the box lies entirely on the wrong side of one of them by more than
// 1e-4 x the magnitude of the terms involved (the fp32 evaluation of r.x, r.y, r.z errs by
// < 3e-7 x that magnitude, so no point the exact arithmetic would keep is ever skipped).
// Only spatially coherent point orders have tight chunk boxes (rtr_reorder_points).
// GROUPS = false: a cloud whose consecutive points are unrelated (measured at upload: its 256-point chunks
// span more than half of the cloud) and that the caller asked not to sort -- every quad's points fall
// into as many tiles as it has points, so the grouping rounds are skipped and every point claims per lane.
// PACKED: the coordinates come from the PackedXyz form (x4 = its headers, y4 = its planes, z4 unused): 6-9
// bytes per point instead of 12 for spatially ordered clouds.  A chunk's header is requested one iteration
// before its planes, the planes one iteration before they are decoded.
//
// LANE TEST (the light path of every chunk).  The kernel is bound by the instructions it issues, not by HBM, and
// nine chunks in ten hold no point inside the frustum: decoding and projecting all 256 points of such a chunk only to
// discard them was two thirds of its instructions.  So a lane first decodes and projects ONE of its four points (the
// first: 4 l) and bounds the other three by the chunk's LANE SPREAD s (Cloud::spread: no coordinate of points
// 4 l + 1 .. 4 l + 3 differs from point 4 l's by more than s, measured when the cloud is uploaded): every row of the
// matrix is linear, so |row(p_k) - row(p_0)| <= (|m0| + |m1| + |m2|) s.  With M one of the four margins of the
// conservative test above (r.x + 0.75 r.z, (W + 0.25) r.z - r.x, same in y):
//   the lane holds no point in front of the camera    when r.z(p_0) + lz s <= 0,
//   the lane holds no point inside the frustum         when M(p_0) + lall s < 0 for some margin,
// lz / lall the L1 norms of the rows involved (host: lane_test_consts).  The margin step transfers a bound from p_0 to
// p_k through REAL arithmetic, while the exact test of p_k runs on its ROUNDED rows: the rows' rounding errors
// (<= 4 x 2^-24 of their term magnitudes each) must stay below the margins' 0.25 px = 0.25 r.z, hence it is only applied
// when every point of the lane has r.z > zsafe.  (A point the exact test accepts has r.x >= -0.5 r.z (1 + 3u), u = 2^-24,
// i.e. M >= 0.25 r.z in its rounded rows; each row errs by <= 4 u T from the real one, T the sum of its term magnitudes,
// so M(p_0) as computed is >= 0.25 r.z(p_k) - lall s - 8 u (Tx + (W + 0.25) Tz): a wrong rejection needs
// r.z(p_k) < 32 u (Tx + (W + 0.25) Tz) = 2^-19 (...); zsafe = 2^-18 x that bound over the cloud's bounding box, twice
// what is needed -- ~0.15 m for a room and a 1080p camera.)  Nearer lanes just stay candidates.  A chunk with any
// candidate lane takes the full path below (all four points per lane, the exact arithmetic decides as before), so
// frames stay bit-identical; chunks with a non-finite or huge coordinate carry s = +inf and always take it.
struct LaneTest {
    float lz, lall, zsafe;
};
static LaneTest lane_test_consts(const Proj &P, int W, int H, const float absmax[3]) {
    LaneTest t;
    const float *m = P.m;
    const float l1x = fabsf(m[0]) + fabsf(m[1]) + fabsf(m[2]), l1y = fabsf(m[4]) + fabsf(m[5]) + fabsf(m[6]);
    const float l1z = fabsf(m[8]) + fabsf(m[9]) + fabsf(m[10]);
    const float hi = (float)(W > H ? W : H) + 0.25f;
    t.lz = l1z * 1.001f;
    t.lall = ((l1x > l1y ? l1x : l1y) + hi * l1z) * 1.001f;
    auto mag = [&](int r) { return fabsf(m[4 * r]) * absmax[0] + fabsf(m[4 * r + 1]) * absmax[1] + fabsf(m[4 * r + 2]) * absmax[2] + fabsf(m[4 * r + 3]); };
    const float rx = mag(0), ry = mag(1), rz = mag(2);
    t.zsafe = 0x1p-18f * ((rx > ry ? rx : ry) + hi * rz);
    if (!(t.zsafe >= 1e-30f)) t.zsafe = __builtin_inff();  // (NaN / no finite box: the margin step never applies)
    if (!(t.lz < 3e38f) || !(t.lall < 3e38f)) t.zsafe = __builtin_inff(), t.lz = t.lall = 3e38f;
    return t;
}

template <bool CULL, bool GROUPS, bool PACKED>
__global__ __launch_bounds__(kBlock, (PACKED && !CULL) ? RTR_T1_WAVES : 4) void k_project_bin(const float4 *__restrict__ x4, const float4 *__restrict__ y4,
                                                        const float4 *__restrict__ z4,
                                                        const uint4 *__restrict__ rgba4, uint32_t n4, Proj P, int W,
                                                        int H, TileStore S, const float *__restrict__ bounds,
                                                        int clear_split, uint32_t cblock, int xp, LaneTest lt) {
    (void)xp;
    // (!CULL: `bounds` carries the chunks' lane spreads of an unpacked cloud, or null; the packed form has them in its headers)
    const float *const spread = CULL ? nullptr : bounds;
    const bool lane_test = (clear_split & 4) == 0;
    const uint4 *const pk_hdr = reinterpret_cast<const uint4 *>(x4);
    const uint32_t *const pk_planes = reinterpret_cast<const uint32_t *>(y4);    // the A streams
    const uint32_t *const pk_planes_b = reinterpret_cast<const uint32_t *>(z4);  // the B streams
    const float fW = (float)W, fH = (float)H;
    const float hiW = f_add(fW, 0.25f), hiH = f_add(fH, 0.25f);
    const int lane = threadIdx.x & 63;
    const uint32_t stx = (uint32_t)(W + 31) >> 5;  // storage tiles per row (tile_geom)
    uint32_t *const fill = ts_fill(S);
    // (a context holds < 2^32 points: quad and chunk indices are 32-bit, which keeps scalar registers free)
    // Chunk order: a chunk is 256 consecutive points (one quad per lane); round r of the grid stride
    // is the window of NW consecutive chunks r NW .. r NW + NW - 1, chunk r NW + w going to wave w.
    // A spatially ordered cloud makes a window ~a million neighbouring points: either none of them is
    // in the frustum or nearly all are, and then EVERY resident wave waits for its claims at the
    // same time -- nobody issues loads, HBM drains (T1 208 -> 300 us), and the claims queue up on a
    // handful of stream counters.  So the waves are cut into `phases` groups of consecutive
    // workgroups, and group g starts its rounds at g R / phases (wrapping around): at any moment the
    // groups sit in different windows, about one of them claiming while the others stream, and the
    // four workgroups resident on a CU (b, b + 256, ...) belong to four different groups.  Inside a
    // group neighbouring waves still read neighbouring kilobytes (DRAM row locality: dealing runs of
    // 16 chunks to each wave instead cost +50 us), and every wave still samples the whole cloud.
    const uint32_t nchunks = (n4 + 63u) / 64u, NW = gridDim.x * (kBlock / 64), wave = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
    const uint32_t R = (nchunks + NW - 1u) / NW;
#ifdef RTR_EXPERIMENT  // (how long do the waves run, how are the long chunks dealt: tools/stamps.py)
    const unsigned long long t_wave0 = wall_clock64();
    if (lane == 0 && (wave & 63u) == 0u) atomicMax(ts_dbg(S) + 59, ~t_wave0);
#endif
    // (LOAD BALANCE, measured in round 4 and not kept.  The launch ends with its most loaded wave: the chunks that take
    // the long path -- ~4.5 us each against ~1 us for a chunk that only streams -- come in runs shorter than a round, so
    // the waves hold 5 +- 1.5 of them, up to 9 on C3, and T1 = the stream + 9 x 4.5 us where the average wave has 5;
    // an RTR_EXPERIMENT build's histogram of the waves' durations shows the same +-17 us.  (a) A queue in device memory
    // for the candidates of waves past the previous frame's average + 1..3, served by the waves that have finished --
    // tickets, one waiter per slot, compare-and-swap on leaving so that every chunk is processed exactly once, bit-exact:
    // T1 137-197 us against 123 -- a hand-over costs its wave two dependent round trips, about what the chunk would
    // have cost, and the helpers only exist once their own share is done.  (b) The same inside a workgroup, through LDS,
    // its four waves dealt shares a quarter of a round apart so that their loads are independent: 120.5-122.5 against
    // 123 on C3, 25.8-26.4 against 23.5 on the 1e7-point cloud, whose stream the four fronts slow down.)
    // cblock = 0 (the default): one group, unless the PREVIOUS frame of this tile store had more than a quarter
    // of the cloud inside the frustum (its entry count is still in the header; T1's epilogue rewrites it when
    // every workgroup is past this line).  Then the claims, not the stream, bound the kernel -- ~390 k wave
    // claims on the dozen stream counters of a distant overview -- and 16 groups that sit in different parts
    // of the cloud, i.e. in different tiles, spread them: 1.45 -> 0.87 ms for 1e8 points inside 100 x 40
    // pixels, against +10 us on an ordinary view, which therefore keeps the single dense streaming front.
    // The packed kernel (round 4: the light path reads a quarter of each chunk through an LDS ring and is bound by the
    // instructions it issues, no longer by the stream) takes FIVE groups by default when a wave has at least 16 rounds: the
    // five workgroups of a CU then sit in five stretches of the cloud, so a stretch inside the frustum puts one of a
    // SIMD's five waves on the long path at a time instead of all of them -- 104.1 -> 99.5-100.8 us on C3; equal on the
    // sorted uniform_box and on BASELINE C2's 1e7 points (fewer rounds: one group); the fp32 stream, which IS at HBM's
    // rate, keeps its single front (five groups: 227 us against 201-205).
    const uint32_t auto_groups = ts_hdr(S)[kHdrEntries] > n4 ? 16u : ((PACKED && !CULL && GROUPS && R >= 16u) ? 5u : 1u);
    const uint32_t G = cblock < 1u ? auto_groups : (cblock > gridDim.x ? gridDim.x : cblock);
    const uint32_t phase = (uint32_t)((uint64_t)((blockIdx.x * G) / gridDim.x) * R / G);
    auto chunk_of = [&](uint32_t q) -> uint32_t {  // q-th chunk of this wave, q < R (>= nchunks: none)
        uint32_t r = q + phase;
        r = r >= R ? r - R : r;
        const uint32_t c = r * NW + wave;  // (< nchunks + NW < 2^25: a context holds < 2^32 points)
        return (q < R && c < nchunks) ? c : nchunks;
    };
    // one quad (four points per lane) of the wave; every exit is wave-uniform
    // the matrix rows for the four points of a lane (render.cu:33-40).  Only the r.z row before the first exit: about
    // half of the chunks of an indoor view lie behind the camera, and the r.x / r.y rows are a third of what a chunk
    // that leaves early costs
    struct Rows { float4 X, Y, Z; float rz[4]; };
    // The matrix lives in VECTOR registers: the kernel runs four waves per SIMD (128 vector registers each, 94 used)
    // but is short of scalar ones -- 63 of them spilled into lanes and came back through v_readlane in the hot loop;
    // with the twelve matrix entries out of the way it is 49, and T1 is 3.5 us faster (0.1880 -> 0.1845 ms per frame).
    float mv[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) {
        mv[k] = P.m[k];
        asm volatile("" : "+v"(mv[k]));
    }
#define RTR_M(k) mv[k]
    auto project_rows = [&](const float4 &X, const float4 &Y, const float4 &Z, Rows &r) {
        const float xs[4] = {X.x, X.y, X.z, X.w}, ys[4] = {Y.x, Y.y, Y.z, Y.w}, zs[4] = {Z.x, Z.y, Z.z, Z.w};
        r.X = X, r.Y = Y, r.Z = Z;
#pragma unroll
        for (int k = 0; k < 4; ++k) r.rz[k] = f_add(fmaf(RTR_M(10), zs[k], fmaf(RTR_M(9), ys[k], f_mul(RTR_M(8), xs[k]))), RTR_M(11));
    };
    // the lane test on a lane's first point (see above); wave-uniform result: does any lane stay a candidate
    auto lane_maybe = [&](float x0, float y0, float z0, float sp, bool live) -> bool {
        const float rz0 = f_add(fmaf(RTR_M(10), z0, fmaf(RTR_M(9), y0, f_mul(RTR_M(8), x0))), RTR_M(11));
        const float sz = f_mul(sp, lt.lz);
        const bool front = live && (f_add(rz0, sz) > 0.0f);
        if (__ballot(front) == 0ull) return false;
        const float rx0 = f_add(fmaf(RTR_M(2), z0, fmaf(RTR_M(1), y0, f_mul(RTR_M(0), x0))), RTR_M(3));
        const float ry0 = f_add(fmaf(RTR_M(6), z0, fmaf(RTR_M(5), y0, f_mul(RTR_M(4), x0))), RTR_M(7));
        const float m = fminf(fminf(fmaf(hiW, rz0, -rx0), fmaf(hiH, rz0, -ry0)), fminf(fmaf(0.75f, rz0, rx0), fmaf(0.75f, rz0, ry0)));
        const bool out = (int)(f_sub(rz0, sz) > lt.zsafe) & (int)(m < -f_mul(sp, lt.lall));  // (no branch)
        return __ballot(front && !out) != 0ull;
    };
    uint32_t n_colour = 0;  // chunks of this wave whose colours were loaded (frame statistics)
    // (Skipping the per-point conservative test below for chunks that have been through the lane test -- inside a stretch
    // of the cloud that lies in the frustum nearly every point passes it -- was measured: 135-142 us against 122-124.  A
    // chunk near the camera plane stays a candidate of the lane test, whose margin step needs r.z > zsafe, and it is this
    // per-point test that lets such a chunk go before the exact arithmetic.)
    auto do_quad = [&](uint32_t i, bool live, const Rows &r) {
        const float *rz = r.rz;
        // (one max3 + max + compare instead of four compares and their combination; fmaxf skips NaNs, and an
        // all-NaN quad compares false)
        const bool front = live && (fmaxf(fmaxf(rz[0], rz[1]), fmaxf(rz[2], rz[3])) > 0.0f);  // render.cu:63
        if (__ballot(front) == 0ull) return;
        float rx[4], ry[4];
        {
            const float xs[4] = {r.X.x, r.X.y, r.X.z, r.X.w}, ys[4] = {r.Y.x, r.Y.y, r.Y.z, r.Y.w}, zs[4] = {r.Z.x, r.Z.y, r.Z.z, r.Z.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                rx[k] = f_add(fmaf(RTR_M(2), zs[k], fmaf(RTR_M(1), ys[k], f_mul(RTR_M(0), xs[k]))), RTR_M(3));
                ry[k] = f_add(fmaf(RTR_M(6), zs[k], fmaf(RTR_M(5), ys[k], f_mul(RTR_M(4), xs[k]))), RTR_M(7));
            }
        }
        bool maybe[4], any = false;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            // branch-free: the four margins as fused multiply-adds (one rounding each, < 1e-3 z against margins of
            // 0.25 z: still conservative), their minimum, one compare.  A NaN margin is skipped by fminf: the
            // point then stays a candidate and the exact arithmetic below decides.
            const float z = rz[k];
            const float m = fminf(fminf(fmaf(hiW, z, -rx[k]), fmaf(hiH, z, -ry[k])), fminf(fmaf(0.75f, z, rx[k]), fmaf(0.75f, z, ry[k])));
            const bool out = (z > 1e-30f) && (m < 0.0f);
            maybe[k] = live && (z > 0.0f) && !out;
            any = any || maybe[k];
        }
        if (__ballot(any) == 0ull || RTR_XP(64)) return;
        bool in[4];
        uint32_t st[4], pix[4];
        unsigned long long pm[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            in[k] = false;
            st[k] = 0;
            pix[k] = 0;
            pm[k] = 0ull;
            if (__ballot(maybe[k]) == 0ull) continue;
            float inv = 1.0f / rz[k];                 // correctly rounded (contract option B)
            float fu = rintf(f_mul(rx[k], inv));      // render.cu:65
            float fv = rintf(f_mul(ry[k], inv));      // render.cu:66
            in[k] = maybe[k] && (fu >= 0.0f) && (fu < fW) && (fv >= 0.0f) && (fv < fH);  // render.cu:68
            pm[k] = __ballot(in[k]);
            if (in[k]) {
                const int u = (int)fu, v = (int)fv;
                st[k] = (uint32_t)(v >> 4) * stx + (uint32_t)(u >> 5);
                pix[k] = (uint32_t)(((v & 15) << 5) | (u & 31));
            }
        }
        if ((pm[0] | pm[1] | pm[2] | pm[3]) == 0ull || RTR_XP(8)) return;
        n_colour += 1u;  // (wave-uniform: a scalar register)
        // the lane's four colours in one 16-byte load, in flight together with the claims.  Unconditional
        // (and the claims below write variables that have no other definition): a value that merges with
        // another one at the end of a divergent block is waited for right there, which turned one round
        // trip per quad into five
        const uint4 col = rgba4[i];
        // group by storage tile; group `it` is claimed by lane `it` (every lane is active here: the callers mask
        // points past the end of the cloud with `live` instead of branching around them)
        int grp[4] = {-1, -1, -1, -1};
        uint32_t rank[4] = {0, 0, 0, 0};
        uint32_t covered = 0;
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wuninitialized"
#pragma clang diagnostic ignored "-Wsometimes-uninitialized"
        uint32_t claim[kMaxGroups];  // only the claiming lane's value is ever read (readlane below)
#pragma clang diagnostic pop
        int ng = 0;
#pragma unroll
        for (int it = 0; it < (GROUPS ? kMaxGroups : 0); ++it) {
            const int kk = pm[0] ? 0 : (pm[1] ? 1 : (pm[2] ? 2 : (pm[3] ? 3 : -1)));
            if (kk < 0) continue;  // wave-uniform
            const unsigned long long pk = kk == 0 ? pm[0] : (kk == 1 ? pm[1] : (kk == 2 ? pm[2] : pm[3]));
            const uint32_t sk = kk == 0 ? st[0] : (kk == 1 ? st[1] : (kk == 2 ? st[2] : st[3]));
            const int first = __ffsll((long long)pk) - 1;
            const uint32_t lead = (uint32_t)__builtin_amdgcn_readlane((int)sk, first);
            // ranks are lane-major: the (up to four) entries of a lane are neighbours in the stream, so a
            // lane whose four points share the tile writes them as two 16-byte stores
            uint32_t total = 0, lower = 0;
            bool gk[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                // (pm[k] IS "in the frustum and not grouped yet": one compare, the rest on the scalar unit; the mask comes
                // back as the lane's condition without a vector instruction)
                const unsigned long long m = __ballot(st[k] == lead) & pm[k];
                gk[k] = __builtin_amdgcn_inverse_ballot_w64(m);
                lower += __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                total += (uint32_t)__popcll(m);
                pm[k] &= ~m;
            }
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (gk[k]) {
                    grp[k] = it;
                    rank[k] = lower++;
                }
            if (lane == it) claim[it] = atomicAdd(fill + ((size_t)lead << S.fill_shift), total);
            ng = it + 1;
            covered += total;
            if (it == 3 && covered <= 8u) {            // four tiles, at most two points each: an incoherent cloud (or a
                pm[0] = pm[1] = pm[2] = pm[3] = 0ull;  // sliver of the frustum's edge); more rounds cost more than they
            }                                          // save -- the remaining points claim per lane below
        }
        // whatever is left belongs to a fourth, fifth, ... tile: one claim per point
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (in[k] && grp[k] < 0) {
                rank[k] = atomicAdd(fill + ((size_t)st[k] << S.fill_shift), 1u);
                grp[k] = kMaxGroups;
            }
        // (only the ng groups that exist -- usually two or three of twelve -- cost a readlane and four selects)
        uint32_t bsel[4] = {0u, 0u, 0u, 0u};
        auto select_base = [&](auto self, auto it_tag) -> void {  // nested wave-uniform tests: it < ng, statically indexed
            constexpr int it = decltype(it_tag)::value;
            if constexpr (it < kMaxGroups) {
                if (it < ng) {
                    const uint32_t b_it = (uint32_t)__builtin_amdgcn_readlane((int)claim[it], it);
#pragma unroll
                    for (int k = 0; k < 4; ++k) bsel[k] = grp[k] == it ? b_it : bsel[k];
                    self(self, std::integral_constant<int, it + 1>{});
                }
            }
        };
        if (!RTR_XP(16)) select_base(select_base, std::integral_constant<int, 0>{});  // (xp 16: the claims are issued, never waited for)
        const uint32_t cs[4] = {col.x, col.y, col.z, col.w};
        uint32_t v[4];
        bool dyn = false;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            v[k] = bsel[k] + rank[k];
            dyn = dyn || (in[k] && v[k] >= kS0);
        }
        if (RTR_XP(4)) return;
        if (RTR_XP(512)) {  // claims waited for, nothing stored
            if ((v[0] ^ v[1] ^ v[2] ^ v[3]) == 0x7FFFFFFFu) fill[0] = 0u;
            return;
        }
        if (__ballot(dyn) == 0ull) {  // the usual case: every position lies in its tile's static extent
            const bool quad = in[0] && in[1] && in[2] && in[3] && grp[0] < kMaxGroups && grp[0] == grp[1] &&
                              grp[0] == grp[2] && grp[0] == grp[3];  // same group: same tile, ranks r, r+1, r+2, r+3
            if (quad) {
                typedef unsigned long long ull2 __attribute__((ext_vector_type(2)));
                struct __attribute__((packed, aligned(8))) Pair { ull2 v; };
                Pair *dst = reinterpret_cast<Pair *>(S.ext0 + ((size_t)st[0] << kS0Shift) + v[0]);
                ull2 a, b;
                a.x = make_entry(__float_as_uint(rz[0]), pix[0], cs[0]);
                a.y = make_entry(__float_as_uint(rz[1]), pix[1], cs[1]);
                b.x = make_entry(__float_as_uint(rz[2]), pix[2], cs[2]);
                b.y = make_entry(__float_as_uint(rz[3]), pix[3], cs[3]);
                dst[0].v = a;
                dst[1].v = b;
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (in[k]) S.ext0[((size_t)st[k] << kS0Shift) + v[k]] = make_entry(__float_as_uint(rz[k]), pix[k], cs[k]);
            }
        } else {
            unsigned long long own[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) own[k] = (in[k] && v[k] >= kS0) ? extent_alloc(S, st[k], v[k]) : 0ull;
            __builtin_amdgcn_wave_barrier();  // every allocation of this wave is published before it polls
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (in[k]) {
                    uint64_t *slot = v[k] < kS0 ? S.ext0 + ((size_t)st[k] << kS0Shift) + v[k] : extent_slot(S, st[k], v[k], own[k]);
                    // (a pointer that comes out of memory is "flat" to the compiler; say that it is global memory: with
                    // a flat store possibly in flight every wait of the kernel becomes a full drain)
                    typedef uint64_t __attribute__((address_space(1))) *gslot_t;
                    if (slot) *(gslot_t)slot = make_entry(__float_as_uint(rz[k]), pix[k], cs[k]);
                }
        }
    };

    if (!CULL && PACKED) {
        // HISTORY of this loop (rounds 2-4, one bit stream per axis, a register pipeline like the fp32 loop's below, one
        // stage deeper: header of chunk q + 2, planes of chunk q + 1 and the arithmetic of chunk q in flight together).
        // (Two chunks of planes in flight per wave -- buffers A / B, loop unrolled by two -- lift the loads alone from 95
        // to 87 us (one chunk per wave and memory round trip is 4096 x 1.3 KB / ~1 us = 5.4 TB/s), but the whole kernel
        // gets 5 us slower: 14 more spilled scalar registers and their v_readlane traffic.  Measured in rounds 2 and 3.
        // Round 4, with the lane test: a PAIR of chunks per iteration, the light path on 8-byte loads and the long path
        // re-reading the chunk in full (one copy of it in a two-trip loop): 140-145 us against 127-133; the long path
        // DEFERRED to the wave's own turn (one wave of a SIMD at a time on it, the others streaming; the noted chunk read
        // again): 125-133 against 121-125.  The kernel's duration is a wave's serial chain of iterations -- ~60 light ones
        // of about a memory round trip each and ~5 long ones of 5-6 us of dependent latency -- and neither form shortens
        // that chain; a re-read lengthens it.  An L2 PREFETCH of the chunk after next (its header held one step longer, one
        // dword per 64 bytes of its blocks requested into a register nobody reads): 161-162 us against 122 -- a second
        // pass of every line through L1 and the texture addresser costs far more than the shorter round trip gains.  The
        // planes staged through an LDS RING by LDS-DMA (global_load_lds_dwordx4, two slots per wave, the planes of chunk
        // q + 2 requested as soon as chunk q has been read out of its slot: two chunks in flight per wave, no register holds
        // data in flight; bit-exact at the first attempt): 121.4-121.6 us against 120.2-120.8, 1e7 points 22.1-22.6 against
        // 22.7-22.9 -- twice the bytes in flight buy nothing: the stream part of the launch already runs at the rate
        // the chip sustains, what is left is the chain of the chunks inside the frustum.)  All of that held while a chunk was
        // 1.3 KB; the form below reads a quarter of it.
        //
        // NOW: the light path reads the chunk's A streams only (every lane's FIRST value: a quarter of the chunk); the B
        // streams are requested when the lane test leaves a candidate lane.  Past its last chunk a wave re-requests the
        // cloud's last chunk and skips the position.
        //
        // THE RING.  With a quarter of the bytes per chunk the loop is no longer near HBM's rate but bound by its own
        // chain -- one chunk in flight per wave, one memory round trip per iteration (T1 112-117 us against 119-122: 60 %
        // fewer bytes bought 6 %).  So the A streams of the next kRing chunks are in flight at once, and in no register:
        // a chunk's A streams are one contiguous piece of at most 768 bytes, which ONE global_load_lds_dwordx4 (16 bytes
        // per lane, 48 lanes) lands in an LDS slot of the wave; the headers travel the same way (two lanes' worth,
        // 32 bytes), kLead chunks further ahead, because a chunk's data request needs the offset its header holds -- so
        // no header lives in scalar registers any more, and no scalar load's latency sits on the LDS reads' counter.
        // Iteration q: request header q + kRing + kLead; header q + kRing has landed (vmcnt(2 kLead): the requests return in
        // order, two per iteration) -> request its data into the slot chunk q - 1 has left; data q has landed (vmcnt(2 kRing))
        // -> read header and data q out of their slots.  The compiler knows nothing of these requests (inline assembly):
        // every wait for them is written here; its own waits for ordinary loads (the long path's B streams, colours,
        // claims) drain them too, which is only conservative.
        // The loop is bound by the instructions it issues -- the scalar unit is shared by a CU's twenty waves -- so the
        // ring sizes are powers of two, the chunk id runs along incrementally, and addresses are scalar base + lane offset.
        constexpr int kRing = RTR_T1_RING, kLead = RTR_T1_LEAD, kRingH = kRing + kLead;  // (kLead: iterations a header is ahead of its data)
        static_assert(((kRing + 1) & kRing) == 0 && ((kRingH + 1) & kRingH) == 0, "ring sizes: powers of two");
        typedef uint32_t __attribute__((address_space(3))) lds_u32;
        constexpr int kSlotDw = 256;  // (a chunk's A streams: <= 768 bytes; 48 lanes request 16 bytes each)
        __shared__ __attribute__((aligned(16))) uint32_t s_ring[kBlock / 64][kRing + 1][kSlotDw];  // data
        __shared__ __attribute__((aligned(16))) uint32_t s_rhdr[kBlock / 64][kRingH + 1][8];       // headers
        const uint32_t ring_lds = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)(lds_u32 *)s_ring[threadIdx.x >> 6]);
        const uint32_t rhdr_lds = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)(lds_u32 *)s_rhdr[threadIdx.x >> 6]);
        const uint32_t lane16 = 16u * (uint32_t)lane;
        // chunk of position q, incrementally: c(q + 1) = c(q) + NW, back to the wave's first chunk when the round wraps
        const uint32_t c_first = (uint32_t)__builtin_amdgcn_readfirstlane((int)(phase * NW + wave)), c_wrap = R * NW + wave;
        auto next_chunk = [&](uint32_t c) -> uint32_t {
            c += NW;
            return c >= c_wrap ? c - R * NW : c;
        };
        auto req_hdr = [&](uint32_t q, uint32_t c) {  // (c: position q's chunk; past the cloud's end: its last chunk, masked later)
            const uint32_t cc = c < nchunks ? c : nchunks - 1u;
            const uint32_t voff = 32u * cc + lane16, lds = rhdr_lds + 32u * (q & (uint32_t)kRingH);
            if (lane < 2) asm volatile("s_mov_b32 m0, %2\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(pk_hdr), "s"(lds) : "memory", "m0");
        };
        auto req_data = [&](uint32_t q) {  // position q's header has landed
            const lds_u32 *const hs = (const lds_u32 *)(uintptr_t)(rhdr_lds + 32u * (q & (uint32_t)kRingH));
            const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)hs[4]), hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)hs[5]);
            const uint8_t *src = reinterpret_cast<const uint8_t *>(pk_planes) + (((((uint64_t)hi) << 32) | (uint64_t)lo) << 3);
            const uint32_t lds = ring_lds + 4u * (uint32_t)kSlotDw * (q & (uint32_t)kRing);
            if (lane < 48) asm volatile("s_mov_b32 m0, %2\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(lane16), "s"(src), "s"(lds) : "memory", "m0");
        };
        // (prologue: one drain, ~1.5 us once per launch, so that every wait below may count two requests per iteration)
        uint32_t c_req = c_first;
#pragma unroll
        for (int k = 0; k < kRingH; ++k) {
            req_hdr((uint32_t)k, (uint32_t)k < R ? c_req : nchunks);
            c_req = next_chunk(c_req);
        }
#pragma unroll
        for (int k = 0; k < kRing; ++k) {
            asm volatile("s_waitcnt vmcnt(%0)" : : "n"(kRingH - 1) : "memory");  // header k: kRingH - 1 requests behind it
            req_data((uint32_t)k);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef RTR_EXPERIMENT
        uint32_t xp_sink = 0;
#endif
        uint32_t c_use = c_first;
        for (uint32_t q = 0; q < R; ++q) {
            req_hdr(q + (uint32_t)kRingH, q + (uint32_t)kRingH < R ? c_req : nchunks);
            c_req = next_chunk(c_req);
            asm volatile("s_waitcnt vmcnt(%0)" : : "n"(2 * kLead) : "memory");  // header q + kRing (requested kLead iterations ago, or drained)
            req_data(q + (uint32_t)kRing);
            asm volatile("s_waitcnt vmcnt(%0)" : : "n"(2 * kRing) : "memory");  // data q (requested kRing iterations ago, or drained)
            const lds_u32 *const slot = (const lds_u32 *)(uintptr_t)(ring_lds + 4u * (uint32_t)kSlotDw * (q & (uint32_t)kRing));
            const lds_u32 *const hs = (const lds_u32 *)(uintptr_t)(rhdr_lds + 32u * (q & (uint32_t)kRingH));
            typedef uint32_t u32x4_l __attribute__((ext_vector_type(4)));
            const u32x4_l g0 = *reinterpret_cast<const u32x4_l __attribute__((address_space(3))) *>(hs);
            const u32x4_l g1 = *reinterpret_cast<const u32x4_l __attribute__((address_space(3))) *>(hs + 4);
            const uint32_t ww = (uint32_t)__builtin_amdgcn_readfirstlane((int)g0.w);
            const uint32_t bx = g0.x, by = g0.y, bz = g0.z;  // (vector registers: they are only ever OR-ed into values)
            const uint32_t cq = c_use < nchunks ? (c_use | 0x80000000u) : nchunks - 1u;
            c_use = next_chunk(c_use);
            const uint32_t wx = ww & 63u, wy = (ww >> 6) & 63u, wz = (ww >> 12) & 63u;
            if (!(cq >> 31)) continue;  // (wave-uniform) past the wave's last chunk
            // the lane's two dwords of each A stream (bit b l: dword (b l) >> 5, shift (b l) & 31 -- one product for both)
            ChunkRawA raw;
            uint32_t px, py, pz;
            asm("v_mul_u32_u24 %0, %1, %2" : "=v"(px) : "s"(wx), "v"(lane));
            asm("v_mul_u32_u24 %0, %1, %2" : "=v"(py) : "s"(wy), "v"(lane));
            asm("v_mul_u32_u24 %0, %1, %2" : "=v"(pz) : "s"(wz), "v"(lane));
            {
                const uint32_t ix = px >> 5, iy = 2u * wx + (py >> 5), iz = 2u * (wx + wy) + (pz >> 5);
                raw.a[0].d[0] = slot[ix], raw.a[0].d[1] = slot[ix + 1];
                raw.a[1].d[0] = slot[iy], raw.a[1].d[1] = slot[iy + 1];
                raw.a[2].d[0] = slot[iz], raw.a[2].d[1] = slot[iz + 1];
            }
            Rows r;
            float4 X, Y, Z;
#ifdef RTR_EXPERIMENT
            if (RTR_XP(128)) {  // the stream alone: headers, A streams, loop bookkeeping
                xp_sink ^= raw.a[0].d[0] ^ raw.a[1].d[1] ^ raw.a[2].d[0] ^ raw.a[0].d[1] ^ raw.a[1].d[0] ^ raw.a[2].d[1];
                continue;
            }
#endif
            // lane test: one point per lane; (wave-uniform) chunks with a 32-bit axis or without a finite spread skip it
            bool cand = true;
            const uint32_t sp_c = (uint32_t)__builtin_amdgcn_readfirstlane((int)g1.z);
            if (lane_test && !(ww & kPackWideFlag) && sp_c < 0x7F000000u) {
                // (b = 0: the mask is empty, the value is the base.  Lanes past the cloud's end hold copies of its last
                // quad -- k_pack_write -- so the test needs no mask of its own: the long path has one)
                auto value0 = [&](uint32_t d0, uint32_t d1, uint32_t prod, uint32_t b, uint32_t base) -> float {
                    uint32_t x;
                    asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(x) : "v"(__builtin_amdgcn_alignbit(d1, d0, prod)), "s"((1u << b) - 1u), "v"(base));
                    return __uint_as_float(x);
                };
                const float x0 = value0(raw.a[0].d[0], raw.a[0].d[1], px, wx, bx);
                const float y0 = value0(raw.a[1].d[0], raw.a[1].d[1], py, wy, by);
                const float z0 = value0(raw.a[2].d[0], raw.a[2].d[1], pz, wz, bz);
                cand = lane_maybe(x0, y0, z0, __uint_as_float(sp_c), true);
            }
            if (!cand) continue;
            uint32_t i_c = (cq & 0x7FFFFFFFu) * 64u + (uint32_t)lane;
            const bool live_c = i_c < n4;
            i_c = live_c ? i_c : n4 - 1u;  // (masked lanes: any valid address for the colour load)
            {
                const uint32_t sbx = (uint32_t)__builtin_amdgcn_readfirstlane((int)bx), sby = (uint32_t)__builtin_amdgcn_readfirstlane((int)by);
                const uint32_t sbz = (uint32_t)__builtin_amdgcn_readfirstlane((int)bz);
                const uint4 hc0 = make_uint4(sbx, sby, sbz, ww);
                const uint4 hc1 = make_uint4((uint32_t)__builtin_amdgcn_readfirstlane((int)g1.x), (uint32_t)__builtin_amdgcn_readfirstlane((int)g1.y), 0u, 0u);
                ChunkRaw raw_b;
#ifdef RTR_EXPERIMENT
                if (RTR_XP(1024)) {  // (what the B streams' round trip costs: the A data in their place, wrong frames)
#pragma unroll
                    for (int a = 0; a < 3; ++a) raw_b.a[a].d[0] = raw.a[a].d[0], raw_b.a[a].d[1] = raw.a[a].d[1], raw_b.a[a].d[2] = raw.a[a].d[0], raw_b.a[a].d[3] = raw.a[a].d[1];
                } else
#endif
                raw_b = load_chunk_b(pk_planes_b, hc0, hc1, lane);
                unpack_chunk(raw, raw_b, ww, sbx, sby, sbz, X, Y, Z, lane);
                project_rows(X, Y, Z, r);
            }
#ifdef RTR_EXPERIMENT
            if (RTR_XP(256)) {  // ... + decode + the three matrix rows
                xp_sink ^= __float_as_uint(r.rz[0]) ^ __float_as_uint(r.rz[1]) ^ __float_as_uint(r.rz[2]) ^ __float_as_uint(r.rz[3]);
                continue;
            }
#endif
            do_quad(i_c, live_c, r);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the requests past the wave's last chunk: into LDS, before it is left)
#ifdef RTR_EXPERIMENT
        if (xp_sink == 0x12345678u) fill[0] = 0u;  // practically never; keeps the work alive
#endif

    } else if (!CULL) {
        // Software pipeline: the coordinates of the wave's NEXT quad are requested as soon as the current
        // ones have gone through the matrix rows, i.e. before the long part of an in-frustum quad (claims,
        // colour load, stores).  A wave waiting for its claims then still has three kilobyte-loads in
        // flight, which is what keeps HBM busy with only 4 waves per SIMD.  Lanes past the end of the cloud
        // re-read its last quad and are masked (`live`).
        float4 X = make_float4(0.f, 0.f, 0.f, 0.f), Y = X, Z = X;
        uint32_t i = 0, spb = 0x7F800000u;
        bool have = false, live = false;
        auto fetch = [&](uint32_t q) {
            const uint32_t c = (uint32_t)__builtin_amdgcn_readfirstlane((int)(q < R ? chunk_of(q) : nchunks));
            have = c < nchunks;  // wave-uniform
            if (have) {
                i = c * 64u + (uint32_t)lane;
                live = i < n4;
                const uint32_t ic = live ? i : n4 - 1u;
                if (spread) spb = __float_as_uint(spread[c]);  // (a scalar load, in flight with the coordinates)
                X = ld_stream(x4 + ic);
                Y = ld_stream(y4 + ic);
                Z = ld_stream(z4 + ic);
            }
        };
        fetch(0);
        for (uint32_t q = 0; q < R; ++q) {
            Rows r;
            const bool have_c = have, live_c = live;
            const uint32_t i_c = i < n4 ? i : n4 - 1u;  // (masked lanes past the end: any valid address for the colour load)
            bool cand = have_c;
            if (have_c && lane_test && spread && spb < 0x7F000000u)
                cand = lane_maybe(X.x, Y.x, Z.x, __uint_as_float(spb), live_c);  // (the lane test: one point per lane)
            if (cand) project_rows(X, Y, Z, r);
            fetch(q + 1);
            if (cand) do_quad(i_c, live_c, r);
        }
    } else {
        // 64 of the wave's chunks are tested at once, one per lane, then only the survivors are
        // streamed: the box test costs 1/64 and its load latency is paid once per 64 chunks.
        float pl[5][4], plm[5][3], pld[5];  // half-space coefficients, |coefficients| row sums, |offset| sums
        {
            const float comb[5][3] = {{0.f, 0.f, 1.f}, {1.f, 0.f, 1.f}, {-1.f, 0.f, fW}, {0.f, 1.f, 1.f}, {0.f, -1.f, fH}};
#pragma unroll
            for (int q = 0; q < 5; ++q) {
#pragma unroll
                for (int k = 0; k < 4; ++k) pl[q][k] = comb[q][0] * P.m[k] + comb[q][1] * P.m[4 + k] + comb[q][2] * P.m[8 + k];
#pragma unroll
                for (int k = 0; k < 3; ++k)
                    plm[q][k] = fabsf(comb[q][0] * P.m[k]) + fabsf(comb[q][1] * P.m[4 + k]) + fabsf(comb[q][2] * P.m[8 + k]);
                pld[q] = fabsf(comb[q][0] * P.m[3]) + fabsf(comb[q][1] * P.m[7]) + fabsf(comb[q][2] * P.m[11]);
            }
        }
        for (uint32_t g0 = 0; g0 < R; g0 += 64) {
            const uint32_t chunk = chunk_of(g0 + (uint32_t)lane);
            const bool valid = chunk < nchunks;
            if (__ballot(valid) == 0ull) continue;
            bool keep = false;
            if (valid) {
                const float *b = bounds + 6 * (size_t)chunk;
                const float lo[3] = {b[0], b[1], b[2]}, hi[3] = {b[3], b[4], b[5]};
                bool culled = false;
#pragma unroll
                for (int q = 0; q < 5; ++q) {
                    float v = pl[q][3], m = pld[q];
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        float t0 = pl[q][k] * lo[k], t1 = pl[q][k] * hi[k];
                        v += t0 > t1 ? t0 : t1;
                        float e0 = fabsf(lo[k]), e1 = fabsf(hi[k]);
                        m += plm[q][k] * (e0 > e1 ? e0 : e1);
                    }
                    culled = culled || (v < -1e-4f * m);  // NaN / inf boxes compare false: never culled
                }
                keep = !culled;
            }
            unsigned long long mask = __ballot(keep);
            while (mask) {
                const int l = __ffsll((long long)mask) - 1;
                mask &= mask - 1;
                const uint32_t i = chunk_of(g0 + (uint32_t)l) * 64u + (uint32_t)lane;
                const bool live = i < n4;  // (lanes past the end of the cloud re-read its last quad, masked)
                const uint32_t ic = live ? i : n4 - 1u;
                float4 X, Y, Z;
                if (PACKED) {
                    const uint32_t cc = (uint32_t)__builtin_amdgcn_readfirstlane((int)(i >> 6));
                    const uint4 h0 = pk_hdr[2 * (size_t)cc], h1 = pk_hdr[2 * (size_t)cc + 1];
                    const ChunkRawA raw_a = load_chunk_a(pk_planes, h0, h1, lane);
                    const ChunkRaw raw = load_chunk_b(pk_planes_b, h0, h1, lane);
                    unpack_chunk(raw_a, raw, h0.w, h0.x, h0.y, h0.z, X, Y, Z, lane);
                } else {
                    X = ld_stream(x4 + ic), Y = ld_stream(y4 + ic), Z = ld_stream(z4 + ic);
                }
                Rows r;
                project_rows(X, Y, Z, r);
                do_quad(ic, live, r);
            }
        }
    }
#ifdef RTR_EXPERIMENT
    if (lane == 0) {
        const unsigned long long t_end = wall_clock64();
        // (same-address atomics serialise, ~90 per us, and hold up the claims behind them: a SAMPLE of the waves reports)
        if ((blockIdx.x & 7u) == 0u) {  // histogram of the waves' own durations, 10 us bins from 40 us
            const unsigned long long us = (t_end - t_wave0) / 100ull;
            const int bin = us < 60ull ? 0 : (us >= 200ull ? 7 : (int)((us - 60ull) / 20ull));
            atomicAdd(ts_dbg(S) + 40 + bin, 1ull);
        }
        if ((wave & 63u) == 0u) {
            atomicMax(ts_dbg(S) + 57, t_end);
            atomicMax(ts_dbg(S) + 61, (unsigned long long)n_colour);
            atomicMax(ts_dbg(S) + 56, ~t_end);
            atomicAdd(ts_dbg(S) + 58, t_end);
            atomicAdd(ts_dbg(S) + 60, 1ull);
            atomicAdd(ts_dbg(S) + 62, (unsigned long long)n_colour);
        }
    }
#endif
    if (clear_split & 8) {
        // a LEAN frame (lean_frame_end): no ticket, no epilogue -- the tile kernel's workgroups read the stream counters
        // themselves.  Only the colour-chunk statistic leaves, one fire-and-forget add per wave that has any.
        if (lane == 0 && n_colour) atomicAdd(ts_sub_colour(S, wave & (uint32_t)(kSubTickets - 1)), (unsigned long long)n_colour);
        return;
    }
    // every claim of this workgroup has returned (its value was used); the workgroup that takes the
    // last ticket sees every stream length final
    // The ticket is the low word of a 64-bit counter whose high word sums the workgroups' colour-chunk counts
    // (frame statistics: bench.py prices the kernel by the bytes it moves) -- one atomic per workgroup for both.
    __shared__ uint32_t s_last, s_colour_total;
    __shared__ uint32_t s_colour[kBlock / 64];
#ifdef RTR_EXPERIMENT
    const unsigned long long t_done = wall_clock64();
#endif
    if (lane == 0) s_colour[threadIdx.x >> 6] = n_colour;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t mine = 0;
#pragma unroll
        for (int w = 0; w < kBlock / 64; ++w) mine += s_colour[w];
        // two levels (rtr_kernels.h, sub[]): workgroup b arrives at word b % 32; the last arrival there zeroes the word
        // for the next frame and carries the group's colour count to the ticket proper
        const uint32_t ng = gridDim.x < (uint32_t)kSubTickets ? gridDim.x : (uint32_t)kSubTickets, grp = blockIdx.x % ng;
        const uint32_t gsize = (gridDim.x - grp + ng - 1u) / ng;
        unsigned long long *const sub = ts_sub(S, grp);
        const unsigned long long old = atomicAdd(sub, 1ull | ((unsigned long long)mine << 32));
        uint32_t last = 0u;
        if ((uint32_t)old == gsize - 1u) {
            __hip_atomic_store(sub, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned long long group_colour = (old >> 32) + mine;
            const unsigned long long old2 = atomicAdd(reinterpret_cast<unsigned long long *>(ts_ticket(S)), 1ull | (group_colour << 32));
            last = (uint32_t)old2 == ng - 1u ? 1u : 0u;
            s_colour_total = (uint32_t)(old2 >> 32) + (uint32_t)group_colour;
        }
        s_last = last;
    }
    __syncthreads();
#ifdef RTR_EXPERIMENT
    if (s_last && threadIdx.x == 0) ts_dbg(S)[0] = t_done;
#endif
    if (s_last && !RTR_XP(32)) bin_epilogue(S, W, H, clear_split, s_colour_total);
    if (s_last && RTR_XP(32) && threadIdx.x == 0) *reinterpret_cast<unsigned long long *>(ts_ticket(S)) = 0ull;  // (only together with xp 8: nothing was claimed)
}

// Option "overlap": T1 runs beside the previous frame's tail, which still reads and writes the frame buffers,
// so its epilogue must not reset the split tiles' pixels (clear_split = 0); this launch does it on the tail's
// stream instead, right before the tile kernel (one workgroup per tile, only split tiles do anything).
__global__ __launch_bounds__(kBlock) void k_reset_split(int W, int H, TileStore S, uint32_t *__restrict__ depth,
                                                        uint32_t *__restrict__ acc) {
    const TileGeom g = tile_geom(W, H);
    const int tile = blockIdx.x;
    if (ts_tile_cnt(S)[tile] <= ts_consts(S)->heavy) return;
    const int tw = 1 << g.tw_shift, tpix = 32 << g.tw_shift;
    const int tx0 = (tile % g.tiles_x) << g.tw_shift, ty0 = (tile / g.tiles_x) * kTileH;
    for (int p = threadIdx.x; p < tpix; p += kBlock) {
        const int x = tx0 + (p & (tw - 1)), y = ty0 + (p >> g.tw_shift);
        if (x < W && y < H) {
            const size_t gp = (size_t)y * W + x;
            depth[gp] = RTR_EMPTY;
            reinterpret_cast<uint4 *>(acc)[gp] = make_uint4(0u, 0u, 0u, 0u);
        }
    }
}
void launch_reset_split(hipStream_t s, int W, int H, const TileStore &S, uint32_t *depth, uint32_t *acc) {
    hipLaunchKernelGGL(k_reset_split, dim3(tile_geom(W, H).ntiles), dim3(kBlock), 0, s, W, H, S, depth, acc);
}

// frames without points: the epilogue alone
__global__ __launch_bounds__(kBlock) void k_bin_empty(int W, int H, TileStore S, int clear_split) {
    bin_epilogue(S, W, H, clear_split, 0u);
}

__device__ __forceinline__ float min2(float a, float b) { return a < b ? a : b; }  // project_cloud.cu:46-49

// F1 folded into T4: the finished depth tile is still in LDS, so the four min-pool levels
// (A8, project_cloud.cu:28-53) of the tile and its min / max partial (A12, render.cu:168-240)
// are produced here instead of by k_pyramid re-reading the depth buffer.  Same rules as
// k_pyramid: level i pixel (x, y) exists iff x < w[i], y < h[i]; min / max over the depth BIT
// patterns of rows < n_eff_rows, sentinel skipped.  `scr` = 4 * tpix free LDS words.
__device__ __forceinline__ void tile_pyramid(const uint32_t *s_depth, uint32_t *scr, const TileGeom &g, int tx0, int ty0,
                                             const FilterLevels &L, uint32_t n_eff_rows, uint32_t *part_min,
                                             uint32_t *part_max, int tile, int tid, int nthreads) {
    const int tw = 1 << g.tw_shift, w1 = tw >> 1, n1 = w1 * 16, sh1 = g.tw_shift - 1;  // widths are powers of two
    float *s1 = reinterpret_cast<float *>(scr), *s2 = s1 + n1, *s3 = s2 + (n1 >> 2);
    uint32_t *s_mm = scr + n1 + (n1 >> 2) + (n1 >> 4);  // [2 * waves]
    uint32_t lo = 0xFFFFFFFFu, hi = 0u;
    for (int q = tid; q < n1; q += nthreads) {
        const int lx = q & (w1 - 1), ly = q >> sh1, gx = (tx0 >> 1) + lx, gy = (ty0 >> 1) + ly;
        const uint32_t *c = s_depth + (2 * ly) * tw + 2 * lx;
        const uint32_t b[4] = {c[0], c[1], c[tw], c[tw + 1]};
        float v = min2(min2(__uint_as_float(b[0]), __uint_as_float(b[1])), min2(__uint_as_float(b[2]), __uint_as_float(b[3])));
        if (gx < L.w[1] && gy < L.h[1]) {
            L.lv[1][(size_t)gy * L.w[1] + gx] = v;
            if ((uint32_t)(2 * gy) < n_eff_rows) {
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (b[k] != RTR_EMPTY) {
                        lo = b[k] < lo ? b[k] : lo;
                        hi = b[k] > hi ? b[k] : hi;
                    }
            }
        }
        s1[q] = v;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        uint32_t ol = __shfl_xor(lo, off, 64), oh = __shfl_xor(hi, off, 64);
        lo = ol < lo ? ol : lo;
        hi = oh > hi ? oh : hi;
    }
    const int nw = nthreads >> 6;
    if ((tid & 63) == 0) {
        s_mm[tid >> 6] = lo;
        s_mm[nw + (tid >> 6)] = hi;
    }
    __syncthreads();
    if (tid == 0) {
        uint32_t a = s_mm[0], b = s_mm[nw];
        for (int k = 1; k < nw; ++k) {
            a = s_mm[k] < a ? s_mm[k] : a;
            b = s_mm[nw + k] > b ? s_mm[nw + k] : b;
        }
        part_min[tile] = a;  // (same-address global atomics instead cost T4 25 us)
        part_max[tile] = b;
    }
    const int w2 = w1 >> 1, n2 = n1 >> 2;
    for (int q = tid; q < n2; q += nthreads) {
        const int lx = q & (w2 - 1), ly = q >> (sh1 - 1), gx = (tx0 >> 2) + lx, gy = (ty0 >> 2) + ly;
        const float *c = s1 + (2 * ly) * w1 + 2 * lx;
        float v = min2(min2(c[0], c[1]), min2(c[w1], c[w1 + 1]));
        if (gx < L.w[2] && gy < L.h[2]) L.lv[2][(size_t)gy * L.w[2] + gx] = v;
        s2[q] = v;
    }
    __syncthreads();
    const int w3 = w2 >> 1, n3 = n2 >> 2;
    for (int q = tid; q < n3; q += nthreads) {
        const int lx = q & (w3 - 1), ly = q >> (sh1 - 2), gx = (tx0 >> 3) + lx, gy = (ty0 >> 3) + ly;
        const float *c = s2 + (2 * ly) * w2 + 2 * lx;
        float v = min2(min2(c[0], c[1]), min2(c[w2], c[w2 + 1]));
        if (gx < L.w[3] && gy < L.h[3]) L.lv[3][(size_t)gy * L.w[3] + gx] = v;
        s3[q] = v;
    }
    __syncthreads();
    const int w4 = w3 >> 1, n4 = n3 >> 2;
    for (int q = tid; q < n4; q += nthreads) {
        const int lx = q & (w4 - 1), ly = q >> (sh1 - 3), gx = (tx0 >> 4) + lx, gy = (ty0 >> 4) + ly;
        const float *c = s3 + (2 * ly) * w3 + 2 * lx;
        if (gx < L.w[4] && gy < L.h[4]) L.lv[4][(size_t)gy * L.w[4] + gx] = min2(min2(c[0], c[1]), min2(c[w3], c[w3 + 1]));
    }
}


// The three truncating divisions of the resolve (render.cu:147-162) with ONE reciprocal: for count < 2^16 the
// sums are < 2^24, exact as floats; sum * rcp(count) is within 1e-4 of the quotient (v_rcp_f32: 1 ulp), so its
// floor is the exact quotient or one off, which the remainder settles.  The u32 division the compiler
// expands costs ~20 vector instructions each, and the tile kernel is half VALU-bound.
__device__ __forceinline__ uint8_t quot_u8(uint32_t a, uint32_t c, float rc) {
    uint32_t q = (uint32_t)((float)a * rc);
    const int32_t r = (int32_t)(a - q * c);
    q = r < 0 ? q - 1u : ((uint32_t)r >= c ? q + 1u : q);
    return (uint8_t)q;
}
__device__ __forceinline__ void resolve3(uint32_t a0, uint32_t a1, uint32_t a2, uint32_t c, uint8_t &q0, uint8_t &q1, uint8_t &q2) {
    if (c == 0u) {
        q0 = q1 = q2 = 0;
    } else if (c < 65536u) {
        const float rc = __builtin_amdgcn_rcpf((float)c);
        q0 = quot_u8(a0, c, rc), q1 = quot_u8(a1, c, rc), q2 = quot_u8(a2, c, rc);
    } else {
        q0 = (uint8_t)(a0 / c), q1 = (uint8_t)(a1 / c), q2 = (uint8_t)(a2 / c);
    }
}

// Owner-computes sharded frames: which rank produces screen tile `tile`.  Every rank evaluates this on the same
// gathered occupancy bitmaps, so all agree: one of the ranks that have points in the tile (the (tile mod n)-th of the
// n occupying ranks: spatially compact point slices give a rank mostly its own tiles, a hash-ordered cloud deals
// them round-robin), -1 when nobody has.  `mask`: the occupying ranks.
constexpr int kOccWordsPerRank = 128;
__device__ __forceinline__ int tile_owner(const uint32_t *occ_all, int world, int tile, uint32_t &mask) {
    mask = 0u;
    for (int r = 0; r < world; ++r) mask |= ((occ_all[r * kOccWordsPerRank + ((tile & 4095) >> 5)] >> (tile & 31)) & 1u) << r;
    if (mask == 0u) return -1;
    int pick = tile % __popc(mask);
    uint32_t m = mask;
    while (pick--) m &= m - 1u;
    return __ffs((int)m) - 1;
}
constexpr int kSegCap4 = 512;  // segments of one tile over all occupying ranks (mode 4; in dynamic LDS)

// T4: per-tile LDS z-buffer.  MODE 0 = whole frame (min + accumulate + resolve, writes
// depth / image / optionally the accumulators; for a split tile only the min phase), MODE 3 = the
// second phase of the split tiles of a whole frame; MODE 1 = min only (depth = min(depth,
// tile min): the phase call before the multi-GPU MIN all-reduce); MODE 2 = accumulate
// only against the depth buffer in memory (acc += tile sums); MODE 4 = MODE 0 for the tiles tile_owner()
// gives to this rank, over the entries of every occupying rank (read from the peers' tile stores); MODE 5 = the min
// phase of the SLICES of a whole frame's split tiles (their minima meet in the depth buffer), MODE 3 their second
// phase -- both in a launch of their own (k_tile_split), empty on ordinary frames, so that MODE 0 stays at 56
// registers (four workgroups per CU: two rounds over the 2040 tiles of a 1080p frame instead of three).
// 512 threads and eight entries in flight per thread.  Work item = tile | slice << 12 |
// (slices - 1) << 22 from T1's epilogue: an unsplit tile (one slice) is owned by one workgroup,
// the slices of a split tile are merged through the frame buffers.
template <int MODE>
__device__ __forceinline__ void tile_body(const TileStore &S, const TileGeom &g, int W, int H, float window,
                                          uint32_t *__restrict__ depth, uint32_t *__restrict__ acc,
                                          uint8_t *__restrict__ img, int write_acc, const TilePyr &pyr, const Sliced &dsl) {
    extern __shared__ uint32_t s_mem[];
    __shared__ unsigned long long s_seg_p0[MODE == 4 ? 1 : kMaxSegs];
    __shared__ uint32_t s_seg_n0[MODE == 4 ? 1 : kMaxSegs], s_seg_pb0[MODE == 4 ? 1 : kMaxSegs];
    __shared__ uint32_t s_nseg, s_flag;
    __shared__ uint4 s_cnt4[MODE == 4 ? kMaxPeers : 1];
    const int tpix = 32 << g.tw_shift;  // pixels per tile
    // (mode 4: the segment table covers every occupying rank and lives behind the tile buffers in dynamic LDS)
    unsigned long long *const s_seg_p = MODE == 4 ? reinterpret_cast<unsigned long long *>(s_mem + 6 * tpix) : s_seg_p0;
    uint32_t *const s_seg_n = MODE == 4 ? s_mem + 6 * tpix + 2 * kSegCap4 : s_seg_n0;
    uint32_t *const s_seg_pb = MODE == 4 ? s_mem + 6 * tpix + 3 * kSegCap4 : s_seg_pb0;
    // MODE 0 is compact: 256 threads and 3.75 LDS words per pixel (depth, the PACKED accumulators, the resolved
    // colours), so that eight workgroups fit a CU -- the 2040 tiles of a 1080p frame are resident at once instead of
    // taking two rounds of a 12 us dependent chain (record, entries, two LDS passes, write-out, pyramid).  A tile
    // that needs the wide accumulators (some pixel blends more than 257 points) is redone in two halves of 16 rows,
    // the wide layout of one half at a time in the same words.
    constexpr bool kCompact = MODE == 0;
    uint32_t *s_depth = s_mem;          // [tpix]
    uint32_t *s_acc = s_mem + tpix;     // [4 * tpix]; compact: [2 * tpix]
    uint8_t *s_rgb = reinterpret_cast<uint8_t *>(s_mem + (kCompact ? 3 : 5) * tpix);  // [3 * tpix] (MODE 0, 3, 4)
    const int tid = threadIdx.x;
    constexpr uint32_t T = kCompact ? kTileThreadsCompact : kTileThreads;
    constexpr int TB = kCompact ? RTR_TILE0_BATCH : kTileBatch;  // entries per thread of a one-batch tile
    constexpr int TP2 = TB / 2, TP4 = TB / 4;                     // ... per stream of a 2- / 4-stream tile
    const int tw = 1 << g.tw_shift;
    // MODE 1 / 2, bit 1: this launch is the only writer of the frame buffer (no rtr_clear before it):
    // store the tile's depth / sums instead of folding them into what memory holds
    const bool overwrite = (write_acc & 2) != 0;
    // bit 2 (sharded frames, rtr_p2p_render): the peers read this rank's depth / accumulators only under the
    // tiles its occupancy bitmap lists (T1's epilogue: tiles with entries), so a tile WITHOUT local entries
    // is not written at all -- with N ranks ~(N - 1) / N of the tiles of a rank's two tile passes
    const bool sparse = (write_acc & 4) != 0;
    const bool lean = MODE == 0 && (write_acc & 8) != 0;  // (see lean_frame_end)
    const int lean_parity = (write_acc >> 4) & 1;          // (every mode-0 / mode-1 launch: which order[] NOT to write)
    // a lean frame's tile workgroup starts with a chain of dependent round trips: launch order -> stream counters ->
    // entries.  Bit 5: every tile workgroup of the launch is resident at once (launch_tile), so the launch order buys
    // nothing and workgroup b takes tile b.  Bit 6 (the host expects full tiles: the previous frame had them): the
    // first batch of entries is requested BEFORE the counters are known -- the addresses depend on the tile alone --
    // and masked when they arrive; with sparse tiles (bit 6 clear) the batch is requested after the counters, from
    // indices clamped to the stream's length, so that a near-empty tile does not pull 16 KB of stale entries.
    const bool lean_ident = lean && (write_acc & 32) != 0;
    const bool lean_early = lean && RTR_TILE0_HEAD && (write_acc & 64) != 0;
    write_acc &= 1;
    const uint4 *const records = reinterpret_cast<const uint4 *>(ts_items(S));
    // Records [0, ntiles): one tile each, for workgroups 0 .. ntiles - 1; records [ntiles, ...): the slices of
    // split tiles, dealt round-robin to the remaining workgroups (mode 3: to all of them) up to kItemEnd.
    const uint32_t nt = (uint32_t)g.ntiles;
    if ((MODE == 0 || MODE == 1) && blockIdx.x == gridDim.x - 1) {  // one extra workgroup, beside ~2000 busy ones
        if (lean) lean_frame_end(S, lean_parity);
        else next_frame_order(S, lean_parity);
        return;
    }
    const bool tile_wg = MODE != 3 && MODE != 5 && blockIdx.x < nt;
    if ((MODE == 4 || MODE == 0) && !tile_wg) return;  // (mode 0: the slices of split tiles have launches of their own -- with
                                                        // them in this kernel it needs 80 registers and spills, without 56)
    const uint32_t split_step = (MODE == 3 || MODE == 5) ? gridDim.x : gridDim.x - nt - ((MODE == 0 || MODE == 1) ? 1u : 0u);
    // (a tile workgroup's record always exists; the others first learn how many slice records there are)
    const uint32_t n_split = tile_wg ? 0u : ts_hdr(S)[kHdrSplitItems];
    const uint32_t first = tile_wg ? blockIdx.x : nt + ((MODE == 3 || MODE == 5) ? blockIdx.x : blockIdx.x - nt);
    // MODE 5 / 3 (k_tile_split): the slice records are taken from a queue instead of being dealt by workgroup index, so
    // that a workgroup which waits between the two phases only ever waits for workgroups that are RUNNING (each holds a
    // record it took) -- never for one that has not been scheduled yet (CU masks, another kernel on the chip)
    constexpr bool kQueue = MODE == 3 || MODE == 5;
    __shared__ uint32_t s_ticket;
    auto take = [&]() -> uint32_t {
        __syncthreads();
        if (tid == 0) s_ticket = atomicAdd(ts_hdr(S) + (MODE == 5 ? kHdrSplitQ1 : kHdrSplitQ2), 1u);
        __syncthreads();
        return nt + s_ticket;
    };

    for (uint32_t item_i = kQueue ? take() : first; tile_wg || item_i < nt + n_split; item_i = kQueue ? take() : item_i + split_step) {
#ifdef RTR_EXPERIMENT
        const bool stamp = MODE == 0 && (blockIdx.x == 5 || blockIdx.x == 700);  // (words 40..: T1's wave histogram)
        const int sb = blockIdx.x == 5 ? 8 : (blockIdx.x == 700 ? 24 : 40);
#define RTR_TSTAMP(k) do { if (stamp && tid == 0) ts_dbg(S)[sb + (k)] = wall_clock64(); } while (0)
#else
#define RTR_TSTAMP(k) do { } while (0)
#endif
        RTR_TSTAMP(0);
        typedef const unsigned long long __attribute__((address_space(1))) *entries_t;
        constexpr unsigned long long kPadMin = (unsigned long long)RTR_EMPTY << 33;  // never lowers a minimum
        constexpr unsigned long long kPadAcc = 0x7F800000ull << 33;                  // +inf fails every window test
        unsigned long long r[TB];  // the first batch of the tile's entries (see below)
        // request a tile's first batch without knowing the streams' lengths (the addresses depend on the tile alone; e <
        // T * PER lies inside the static extent whatever the length): mask_batch() settles what counts
        auto request_batch = [&](auto per_tag, int tx_, int ty_) {
            constexpr int PER = decltype(per_tag)::value;
#pragma unroll
            for (int q = 0; q < TB / PER; ++q) {
                const int st = stream_tile(g, tx_, ty_, q);
                const entries_t src = (entries_t)(S.ext0 + ((size_t)(st >= 0 ? st : 0) << kS0Shift));
#pragma unroll
                for (int j = 0; j < PER; ++j) r[q * PER + j] = src[tid * PER + j];
            }
        };
        // one 32-byte record per work item: everything the workgroup needs to find its entries
        uint4 rec0, rec1;
        uint32_t occ4 = 0u;  // mode 4: the ranks that have points in this tile
        if (MODE == 4) {     // tile = workgroup; the lengths of this rank's own streams by tile
            if (tile_owner(dsl.occ_all, dsl.peers, (int)item_i, occ4) != dsl.rank) break;  // (workgroup-uniform) another
                                                                       // rank's tile, or nobody's: the collector clears it
            const uint4 c4 = ts_cnt4(S)[item_i];
            rec0 = make_uint4(item_i, c4.x, c4.y, c4.z);
            rec1 = make_uint4(c4.w, 0u, 0u, 0u);
        } else if (MODE == 0 && lean) {
            // a lean frame: no work list -- the tile of this launch position, its stream counters read here (T1 is
            // complete) and reset for the next frame (each counter belongs to exactly one tile), its entry count stored
            // for the frame's statistics
            const uint32_t tl = lean_ident ? item_i : ts_order(S, lean_parity)[item_i];
            const int ltx = (int)tl % g.tiles_x, lty = (int)tl / g.tiles_x;
            if (lean_early) {  // (workgroup-uniform)
                if (g.tw_shift == 5) request_batch(std::integral_constant<int, TB / 2>{}, ltx, lty);
                else request_batch(std::integral_constant<int, TB / 4>{}, ltx, lty);
            }
            uint32_t *const fill = ts_fill(S);
            uint32_t f[4] = {0u, 0u, 0u, 0u};
            const int nsl = 2 << (g.tw_shift - 5);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int st = q < nsl ? stream_tile(g, ltx, lty, q) : -1;
                if (st >= 0) f[q] = fill[(size_t)st << S.fill_shift];
            }
            rec0 = make_uint4(tl, f[0], f[1], f[2]);
            rec1 = make_uint4(f[3], 0u, 0u, 0u);
            __syncthreads();  // every wave has read the counters before they are reset
            if (tid == 0) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int st = q < nsl ? stream_tile(g, ltx, lty, q) : -1;
                    if (st >= 0 && f[q]) fill[(size_t)st << S.fill_shift] = 0u;
                }
                ts_lcnt(S, lean_parity)[tl] = f[0] + f[1] + f[2] + f[3];  // (summed up by the next fold)
            }
        } else {
            rec0 = records[2 * (size_t)item_i], rec1 = records[2 * (size_t)item_i + 1];
        }
        const uint32_t item = rec0.x;
        if (item == kItemSkip) break;  // a split tile's own slot (workgroup-uniform)
        const uint32_t sub = (item >> 12) & 1023u, nsub = (item >> 22) + 1u;
        // (k_tile_split: every record of the split region goes through the frame buffers, also a split tile that ends up
        // with ONE slice -- the slice size grows with the frame's split entries, T1's epilogue -- whose own record says skip)
        const bool split = nsub > 1u || MODE == 3 || MODE == 5;
        const bool no_local = (MODE == 1 || MODE == 2) && sparse && (rec0.y + rec0.z + rec0.w + rec1.x) == 0u;
        if (MODE == 1 && no_local) break;  // (workgroup-uniform; an unsplit tile's workgroup has this one item)
        const int tile = (int)(item & 4095u);
        const int tx = tile % g.tiles_x, ty = tile / g.tiles_x;
        const int tx0 = tx << g.tw_shift, ty0 = ty * kTileH;
        // (the segment pointers come out of LDS as integers: say that they are global memory, or the loads go flat)
        // Most tiles hold fewer entries than one batch of the workgroup (8 per thread) in the static extents
        // of their two (four) streams: those are requested right here, before the LDS tile is even
        // initialised, read ONCE and kept in registers across the barrier between the two reference passes.
        // Register group q of a thread holds CONSECUTIVE entries of stream q.
        const int ns = 2 << (g.tw_shift - 5);  // streams
        const bool two = ns == 2;
        const uint32_t lim = T * (two ? TP2 : TP4);  // (<= 2048 < kS0: inside the static extent)
        static_assert(T * TP2 <= kS0, "one batch must fit the static extent");
        const bool one_batch = (MODE == 0 || (MODE == 4 && occ4 == (1u << dsl.rank))) && !split && rec0.y <= lim && rec0.z <= lim &&
                               (ns == 2 || (rec0.w <= lim && rec1.x <= lim));
        // MODE 0, tiles BEYOND one batch (most tiles of a 1e8-point frame): the first batch of every stream is requested
        // and kept in registers all the same -- only what lies behind it is swept, i.e. read twice (round 3 swept
        // everything: 108 MB moved for 54 MB of entries)
        const bool head = RTR_TILE0_HEAD && MODE == 0 && !split && !one_batch;
        auto load_batch = [&](auto per_tag, auto loaded_tag) {  // (compile-time register -> stream mapping: everything stays in registers)
            constexpr int PER = decltype(per_tag)::value;
            constexpr bool kLoaded = decltype(loaded_tag)::value;  // request_batch() has been here: only mask
#pragma unroll
            for (int q = 0; q < TB / PER; ++q) {
                const int st = stream_tile(g, tx, ty, q);
                const entries_t src = (entries_t)(S.ext0 + ((size_t)(st >= 0 ? st : 0) << kS0Shift));
                const uint32_t cq = q == 0 ? rec0.y : (q == 1 ? rec0.z : (q == 2 ? rec0.w : rec1.x));
                // unconditional loads (a load that merges with a constant at the end of a branch is waited for on the
                // spot), masked after; a thread whose entries lie past the stream's end re-reads the stream's first ones
                // (lines its neighbours hold anyway) instead of pulling stale memory: a sparse frame's tile kernel moved
                // 16 KB per tile whatever the tile held
                const uint32_t b = tid * PER, bc = b < cq ? b : 0u;
#pragma unroll
                for (int j = 0; j < PER; ++j) {
                    const unsigned long long got = kLoaded ? r[q * PER + j] : src[bc + j];
                    r[q * PER + j] = (st >= 0 && b + j < cq) ? got : kPadAcc;  // (+inf lowers no minimum)
                }
            }
        };
        if (one_batch || head) {
            if (lean_early) {
                if (two) load_batch(std::integral_constant<int, TP2>{}, std::true_type{});
                else load_batch(std::integral_constant<int, TP4>{}, std::true_type{});
            } else {
                if (two) load_batch(std::integral_constant<int, TP2>{}, std::false_type{});
                else load_batch(std::integral_constant<int, TP4>{}, std::false_type{});
            }
        }
        auto stream_pb = [&](int q) -> uint32_t {  // where stream q's 32x16 storage tile sits in the processing tile
            const int per_row = 1 << (g.tw_shift - 5);
            return (uint32_t)(((q >> (g.tw_shift - 5)) * 16) * tw + (q & (per_row - 1)) * 32);
        };
        __syncthreads();  // the previous item's LDS is no longer read
        if (tid == 0) s_nseg = 0;
        if (!one_batch) __syncthreads();  // workgroup-uniform
        if (MODE == 4 && !one_batch) {
            // the same pieces in the store of EVERY occupying rank: their stream lengths first (one 16-byte read per
            // rank, over xGMI for the peers), then static and dynamic extents through the mapped stores
            if (tid < dsl.peers)
                s_cnt4[tid] = ((occ4 >> tid) & 1u)
                                  ? reinterpret_cast<const uint4 *>(dsl.tab->meta[tid] + ts_off_cnt4(S.nst, S.ntiles))[item_i]
                                  : make_uint4(0u, 0u, 0u, 0u);
            __syncthreads();
            if (tid < kMaxSegs) {
                const int s = tid / kDirK, k = tid % kDirK;
                const int st = s < (2 << (g.tw_shift - 5)) ? stream_tile(g, tx, ty, s) : -1;
                const unsigned long long e_lo = k == 0 ? 0ull : ((unsigned long long)kS0 << (k - 1)), e_hi = (unsigned long long)kS0 << k;
                for (int r = 0; r < dsl.peers; ++r) {
                    const uint4 c4 = s_cnt4[r];
                    const unsigned long long cnt = s == 0 ? c4.x : (s == 1 ? c4.y : (s == 2 ? c4.z : c4.w));
                    if (st < 0 || cnt <= e_lo) continue;
                    const unsigned long long hi = cnt < e_hi ? cnt : e_hi;
                    const unsigned long long *dir_r =
                        reinterpret_cast<const unsigned long long *>(dsl.tab->meta[r] + ts_off_dir(S.nst, S.ntiles));
                    const uint64_t *p = k == 0 ? dsl.tab->ext0[r] + ((size_t)st << kS0Shift)
                                               : dsl.tab->dyn[r] + (dir_r[(size_t)st * kDirK + k] >> 24);
                    const uint32_t q = atomicAdd(&s_nseg, 1u);
                    if (q < (uint32_t)kSegCap4) {
                        s_seg_p[q] = (unsigned long long)p;
                        s_seg_n[q] = (uint32_t)(hi - e_lo);
                        s_seg_pb[q] = stream_pb(s);
                    } else {
                        store_error_now(S, 4u);  // (more pieces than the table holds: entries would be dropped)
                    }
                }
            }
            __syncthreads();
            if (tid == 0 && s_nseg > (uint32_t)kSegCap4) s_nseg = kSegCap4;
        }
        // the contiguous pieces of this item's entries: per stream the static extent and the dynamic ones,
        // clipped to the slice [sub, sub + 1) / nsub of the stream
        if (MODE != 4 && !one_batch && tid < kMaxSegs) {
            const int s = tid / kDirK, k = tid % kDirK;
            const int st = s < (2 << (g.tw_shift - 5)) ? stream_tile(g, tx, ty, s) : -1;
            const unsigned long long cnt = s == 0 ? rec0.y : (s == 1 ? rec0.z : (s == 2 ? rec0.w : rec1.x));
            const unsigned long long e_lo = k == 0 ? 0ull : ((unsigned long long)kS0 << (k - 1));
            if (st >= 0 && cnt > e_lo) {
                const unsigned long long a = cnt * sub / nsub, b = cnt * (sub + 1u) / nsub;
                const unsigned long long e_hi = (unsigned long long)kS0 << k;
                unsigned long long lo = a > e_lo ? a : e_lo;
                const unsigned long long hi = b < e_hi ? b : e_hi;
                if (head && k == 0 && lo < lim) lo = lim;  // (the stream's first batch is in registers)
                // (an extent that was never handed out in this frame, or that lies beyond the pool -- the adaptive pool
                // overflowed, T1 has dropped its entries and flagged the frame -- is not read: the frame is repeated)
                const unsigned long long de = k == 0 ? 0ull : ts_dir(S)[(size_t)st * kDirK + k];
                const bool there = k == 0 || ((uint32_t)(de & 0xFFFFFFull) == S.seq && (de >> 24) + e_lo <= ts_consts(S)->dyn_cap);
                if (hi > lo && there) {
                    const uint64_t *p = k == 0 ? S.ext0 + ((size_t)st << kS0Shift) + lo : ts_consts(S)->dyn + (de >> 24) + (lo - e_lo);
                    const uint32_t q = atomicAdd(&s_nseg, 1u);
                    s_seg_p[q] = (unsigned long long)p;
                    s_seg_n[q] = (uint32_t)(hi - lo);
                    s_seg_pb[q] = stream_pb(s);
                }
            }
        }
        uint32_t occ_mask = 0;  // MODE 2, sharded whole frame: the ranks that have points in this tile
        if (MODE == 2 && dsl.peers)
            for (int r = 0; r < dsl.peers; ++r)
                occ_mask |= ((dsl.occ_all[r * 128 + ((tile & 4095) >> 5)] >> (tile & 31)) & 1u) << r;
        // depth tile
        for (int p = tid; p < tpix; p += T) {
            if (MODE == 2 || MODE == 3) {
                int x = tx0 + (p & (tw - 1)), y = ty0 + (p >> g.tw_shift);
                if (x < W && y < H) {
                    const size_t gp = (size_t)y * W + x;
                    if (MODE == 2 && dsl.peers) {  // sharded frame: MIN over the local depth of the ranks that occupy this tile
                        uint32_t v = RTR_EMPTY;
                        for (int r = 0; r < dsl.peers; ++r)
                            if ((occ_mask >> r) & 1u) {
                                const uint32_t u = static_cast<const uint32_t *>(dsl.src.p[r])[gp];
                                v = u < v ? u : v;
                            }
                        s_depth[p] = v;
                        // (the peers are reading this rank's `depth` right now: the completed frame goes to a buffer
                        // of its own and reaches RTR_BUF_DEPTH once every rank is past this launch)
                        if (sub == 0) dsl.out[gp] = v;
                    } else if (dsl.chunk) {  // sharded frame: the global minimum lies in the ranks' reduced slices
                        const uint32_t v = static_cast<const uint32_t *>(dsl.src.p[gp / dsl.chunk])[gp];
                        s_depth[p] = v;
                        if (sub == 0) depth[gp] = v;  // ... and this launch is what completes RTR_BUF_DEPTH
                    } else {
                        s_depth[p] = depth[gp];
                    }
                } else {
                    s_depth[p] = RTR_EMPTY;
                }
            } else {
                s_depth[p] = RTR_EMPTY;
            }
        }
        __syncthreads();
        RTR_TSTAMP(1);
        const uint32_t nseg = s_nseg;
        uint32_t n_local = 0;
        for (uint32_t q = 0; q < nseg; ++q) n_local += s_seg_n[q];
        if (one_batch || head) n_local = rec0.y + rec0.z + rec0.w + rec1.x;
        const bool do_min = MODE == 1 || MODE == 0 || MODE == 4 || MODE == 5;
        const bool do_acc = MODE == 2 || MODE == 3 || ((MODE == 0 || MODE == 4) && !split);
        // Accumulators: the exact layout is two 64-bit words per pixel, (c0 | c1 << 32) and
        // (c2 | count << 32).  LDS atomics are what bounds this kernel (about one lane per clock
        // and CU), so items with <= 60000 entries first try ONE packed word per pixel and entry,
        // c0 | c1 << 16 | c2 << 32 | count << 48: no 16-bit field can overflow while a pixel's
        // count is <= 257 (255 * 257 = 65535), a carry can only move upwards inside the pixel's
        // own word (so the count field never reads low), and the count cannot wrap below 65536
        // entries.  If any pixel ends with count > 257 the whole item is redone with the wide layout.
        unsigned long long *s_acc64 = reinterpret_cast<unsigned long long *>(s_acc);
        bool narrow = do_acc && !no_local && (n_local <= 60000u);
        // (compact: a tile too heavy for the packed layout -- only with the split threshold raised -- goes straight to
        // the two wide halves below)
        const bool force_wide = kCompact && do_acc && !no_local && !narrow;
        int cur_half = -1;  // compact, wide layout: the half of the tile (rows < 16 / >= 16) the accumulators hold
        if (do_acc && !no_local && !force_wide)
            for (int p = tid; p < (narrow ? 2 : 4) * tpix; p += T) s_acc[p] = 0;
        auto wide_slot = [&](uint32_t p, bool &mine) -> uint32_t {  // pixel -> its slot in the wide accumulators
            if (!kCompact) {
                mine = true;
                return p;
            }
            const uint32_t h = p >= (uint32_t)(tpix >> 1) ? 1u : 0u;
            mine = (int)h == cur_half;
            return p - h * (uint32_t)(tpix >> 1);
        };
        // A thread's registers of one segment hold CONSECUTIVE entries of the stream, i.e. consecutive points
        // of the cloud (T1 writes a lane's four points next to each other), which mostly fall on the same
        // pixel: they are merged in registers first, one LDS atomic per run -- LDS atomics are what bounds
        // this kernel (about one lane per clock and CU).
        auto pixel_of = [&](unsigned long long e, uint32_t pb) -> uint32_t {
            const uint32_t px = (uint32_t)(e >> 24) & 511u;
            return pb + ((px >> 5) << g.tw_shift) + (px & 31u);
        };
        // (Both passes first read the LDS depth under ALL of the thread's entries, then work through them: the reads of
        // a pass are then in flight together instead of one dependent LDS round trip per entry -- the compiler cannot
        // hoist them itself past the atomics on the same array.  A stale early-z value is >= the true minimum: it
        // can only cause a redundant atomic.)
        auto min_runs = [&](auto per_tag) {
            constexpr int PER = decltype(per_tag)::value;
            uint32_t pix[TB], cur[TB];
#pragma unroll
            for (int k = 0; k < TB; ++k) pix[k] = pixel_of(r[k], stream_pb(k / PER));
#pragma unroll
            for (int k = 0; k < TB; ++k) cur[k] = __hip_atomic_load(&s_mem[pix[k]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#pragma unroll
            for (int k0 = 0; k0 < TB; k0 += PER) {
                uint32_t run_p = pix[k0], run_d = (uint32_t)(r[k0] >> 33), run_c = cur[k0];
#pragma unroll
                for (int j = 1; j < PER; ++j) {
                    const uint32_t pj = pix[k0 + j], dj = (uint32_t)(r[k0 + j] >> 33);
                    if (pj != run_p) {
                        if (run_d < run_c) atomicMin(&s_mem[run_p], run_d);
                        run_p = pj;
                        run_d = dj;
                        run_c = cur[k0 + j];
                    } else {
                        run_d = dj < run_d ? dj : run_d;
                    }
                }
                if (run_d < run_c) atomicMin(&s_mem[run_p], run_d);
            }
        };
        auto acc_runs = [&](auto per_tag, bool packed) {
            constexpr int PER = decltype(per_tag)::value;
            auto flush = [&](uint32_t p, unsigned long long v) {  // v: c0 | c1 << 16 | c2 << 32 | count << 48, <= 8 entries
                if (v == 0ull) return;
                if (packed) {
                    atomicAdd(s_acc64 + p, v);
                } else {
                    bool mine;
                    const uint32_t pp = wide_slot(p, mine);
                    if (mine) {
                        atomicAdd(s_acc64 + 2 * pp + 0, (v & 0xFFFFull) | (((v >> 16) & 0xFFFFull) << 32));
                        atomicAdd(s_acc64 + 2 * pp + 1, ((v >> 32) & 0xFFFFull) | ((v >> 48) << 32));
                    }
                }
            };
            uint32_t pix[TB];
            float m[TB];
#pragma unroll
            for (int k = 0; k < TB; ++k) pix[k] = pixel_of(r[k], stream_pb(k / PER));
#pragma unroll
            for (int k = 0; k < TB; ++k) m[k] = __uint_as_float(s_mem[pix[k]]);  // s_depth: final since the barrier
            auto value_of = [&](unsigned long long e, float mk) -> unsigned long long {
                if (__uint_as_float((uint32_t)(e >> 33)) > f_add(mk, window)) return 0ull;  // render.cu:106, then :125-128
                return (e & 0xFFull) | (((e >> 8) & 0xFFull) << 16) | (((e >> 16) & 0xFFull) << 32) | (1ull << 48);
            };
#pragma unroll
            for (int k0 = 0; k0 < TB; k0 += PER) {
                uint32_t run_p = pix[k0];
                unsigned long long run_v = value_of(r[k0], m[k0]);
#pragma unroll
                for (int j = 1; j < PER; ++j) {
                    const uint32_t pj = pix[k0 + j];
                    const unsigned long long vj = value_of(r[k0 + j], m[k0 + j]);
                    if (pj != run_p) {
                        flush(run_p, run_v);
                        run_p = pj;
                        run_v = vj;
                    } else {
                        run_v += vj;
                    }
                }
                flush(run_p, run_v);
            }
        };
        // Tiles with more entries than one batch: every piece of their streams, a batch at a time, with the NEXT batch
        // requested before the current one is worked through.  As in the one-batch case a thread holds CONSECUTIVE
        // entries of the stream -- consecutive points of the cloud, mostly on the same pixel -- and merges them in
        // registers: one LDS atomic per run instead of one per entry (the two passes of a C3 tile are LDS-atomic bound:
        // 8 resident tiles x 3.3 k entries per CU and pass).  The loads are unconditional from clamped indices (a
        // load whose result merges with a constant is waited for on the spot) and masked when they are used.
        constexpr int SB = kCompact ? RTR_TILE0_SWEEP : TB;  // entries per thread and buffer
        constexpr bool kSweepAhead = kCompact ? (RTR_TILE0_AHEAD != 0) : true;
        auto min_seq = [&](const unsigned long long *v, uint32_t pb) __attribute__((always_inline)) {
            uint32_t pix[SB], cur[SB];
#pragma unroll
            for (int k = 0; k < SB; ++k) pix[k] = pixel_of(v[k], pb);
#pragma unroll
            for (int k = 0; k < SB; ++k) cur[k] = __hip_atomic_load(&s_mem[pix[k]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            uint32_t run_p = pix[0], run_d = (uint32_t)(v[0] >> 33), run_c = cur[0];
#pragma unroll
            for (int j = 1; j < SB; ++j) {
                const uint32_t pj = pix[j], dj = (uint32_t)(v[j] >> 33);
                if (pj != run_p) {
                    if (run_d < run_c) atomicMin(&s_mem[run_p], run_d);
                    run_p = pj;
                    run_d = dj;
                    run_c = cur[j];
                } else {
                    run_d = dj < run_d ? dj : run_d;
                }
            }
            if (run_d < run_c) atomicMin(&s_mem[run_p], run_d);
        };
        auto acc_seq = [&](const unsigned long long *v, uint32_t pb, bool packed) __attribute__((always_inline)) {
            auto flush = [&](uint32_t p, unsigned long long w) __attribute__((always_inline)) {  // w: c0 | c1 << 16 | c2 << 32 | count << 48, <= SB entries
                if (w == 0ull) return;
                if (packed) {
                    atomicAdd(s_acc64 + p, w);
                } else {
                    bool mine;
                    const uint32_t pp = wide_slot(p, mine);
                    if (mine) {
                        atomicAdd(s_acc64 + 2 * pp + 0, (w & 0xFFFFull) | (((w >> 16) & 0xFFFFull) << 32));
                        atomicAdd(s_acc64 + 2 * pp + 1, ((w >> 32) & 0xFFFFull) | ((w >> 48) << 32));
                    }
                }
            };
            uint32_t pix[SB];
            float m[SB];
#pragma unroll
            for (int k = 0; k < SB; ++k) pix[k] = pixel_of(v[k], pb);
#pragma unroll
            for (int k = 0; k < SB; ++k) m[k] = __uint_as_float(s_mem[pix[k]]);  // s_depth: final since the barrier
            auto value_of = [&](unsigned long long e, float mk) -> unsigned long long {
                if (__uint_as_float((uint32_t)(e >> 33)) > f_add(mk, window)) return 0ull;  // render.cu:106, then :125-128
                return (e & 0xFFull) | (((e >> 8) & 0xFFull) << 16) | (((e >> 16) & 0xFFull) << 32) | (1ull << 48);
            };
            uint32_t run_p = pix[0];
            unsigned long long run_v = value_of(v[0], m[0]);
#pragma unroll
            for (int j = 1; j < SB; ++j) {
                const uint32_t pj = pix[j];
                const unsigned long long vj = value_of(v[j], m[j]);
                if (pj != run_p) {
                    flush(run_p, run_v);
                    run_p = pj;
                    run_v = vj;
                } else {
                    run_v += vj;
                }
            }
            flush(run_p, run_v);
        };
        auto sweep = [&](auto proc, unsigned long long pad) __attribute__((always_inline)) {
            constexpr uint32_t STEP = SB * T;
            for (uint32_t q = 0; q < nseg; ++q) {
                // (workgroup-uniform values out of LDS: into scalar registers, so that every load is base + a 32-bit
                // lane offset instead of holding a 64-bit address of its own)
                const unsigned long long pv = s_seg_p[q];
                const entries_t ent = (entries_t)(((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(pv >> 32)) << 32) |
                                                  (unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)pv));
                const uint32_t n = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_seg_n[q]);  // (n >= 1)
                const uint32_t pb = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_seg_pb[q]);
                unsigned long long ra[SB], rb[SB];
                // (a thread's SB entries from ONE address: threads past the end re-read the piece's first entries, the
                // last thread inside it reads up to SB - 1 entries past it -- inside the next extent, or inside the
                // slack every allocation of entries ends with; both are masked by index in work())
                auto fetch = [&](unsigned long long *dst, uint32_t base) __attribute__((always_inline)) {
                    const uint32_t b = base + (uint32_t)tid * SB, bc = b < n ? b : 0u;
#pragma unroll
                    for (int k = 0; k < SB; ++k) dst[k] = ent[bc + k];
                };
                auto work = [&](const unsigned long long *src, uint32_t base) __attribute__((always_inline)) {
                    unsigned long long v[SB];
#pragma unroll
                    for (int k = 0; k < SB; ++k) v[k] = (base + (uint32_t)tid * SB + k < n) ? src[k] : pad;
                    proc(v, pb);
                };
                if (kSweepAhead) {
                    fetch(ra, 0u);
                    for (uint32_t base = 0; base < n; base += 2 * STEP) {  // (workgroup-uniform)
                        fetch(rb, base + STEP);
                        work(ra, base);
                        if (base + STEP >= n) break;
                        fetch(ra, base + 2 * STEP);
                        work(rb, base + STEP);
                    }
                } else {
                    for (uint32_t base = 0; base < n; base += STEP) {
                        fetch(ra, base);
                        work(ra, base);
                    }
                }
            }
        };
        if (one_batch) {
#ifdef RTR_EXPERIMENT
            if (stamp && tid == 0 && r[0] != 1ull) ts_dbg(S)[sb + 11] = wall_clock64();  // first entry has arrived
            if (stamp && tid == 0 && (r[3] ^ r[7]) != 1ull) ts_dbg(S)[sb + 12] = wall_clock64();  // all have
#endif
            if (two) min_runs(std::integral_constant<int, TP2>{});
            else min_runs(std::integral_constant<int, TP4>{});
        } else if (do_min) {
            if (head) {
                if (two) min_runs(std::integral_constant<int, TP2>{});
                else min_runs(std::integral_constant<int, TP4>{});
            }
            sweep([&](const unsigned long long *v, uint32_t pb) __attribute__((always_inline)) { min_seq(v, pb); }, kPadMin);
        }
        RTR_TSTAMP(2);
        __syncthreads();
        RTR_TSTAMP(3);
        auto accumulate = [&](bool packed) __attribute__((always_inline)) {
            if (one_batch || head) {
                if (two) acc_runs(std::integral_constant<int, TP2>{}, packed);
                else acc_runs(std::integral_constant<int, TP4>{}, packed);
                if (one_batch) return;
            }
            sweep([&](const unsigned long long *v, uint32_t pb) __attribute__((always_inline)) { acc_seq(v, pb, packed); }, kPadAcc);
        };
        if (do_acc) {
            if (!force_wide) accumulate(narrow);
            RTR_TSTAMP(9);
            __syncthreads();
            RTR_TSTAMP(10);
            if (narrow && !((MODE == 0 || MODE == 4) && !split)) {  // (whole unsplit tiles check while they write out, below)
                int over = 0;
                for (int p = tid; p < tpix; p += T) over |= (s_acc64[p] >> 48) > 257ull;
                if (__syncthreads_or(over)) {  // rare: some pixel blends more than 257 points
                    narrow = false;
                    for (int p = tid; p < 4 * tpix; p += T) s_acc[p] = 0;
                    __syncthreads();
                    accumulate(false);
                    __syncthreads();
                }
            }
        }
        RTR_TSTAMP(4);
        auto sums_of = [&](int p, uint32_t &a0, uint32_t &a1, uint32_t &a2, uint32_t &c) {
            if (narrow) {
                const unsigned long long pk = s_acc64[p];
                a0 = (uint32_t)(pk & 0xFFFFu);
                a1 = (uint32_t)((pk >> 16) & 0xFFFFu);
                a2 = (uint32_t)((pk >> 32) & 0xFFFFu);
                c = (uint32_t)(pk >> 48);
            } else {
                const int pp = (kCompact && cur_half > 0) ? p - (tpix >> 1) : p;
                a0 = s_acc[4 * pp], a1 = s_acc[4 * pp + 1], a2 = s_acc[4 * pp + 2], c = s_acc[4 * pp + 3];
            }
        };
        bool finish = !split;  // this workgroup resolves / writes the image rows and emits the pyramid
        if (split) {
            // the slices of a split tile meet in the frame buffers, which hold the sentinel / zero (or what
            // the caller's rtr_clear and earlier passes left there)
            for (int p = tid; p < tpix; p += T) {
                const int x = tx0 + (p & (tw - 1)), y = ty0 + (p >> g.tw_shift);
                if (!(x < W && y < H)) continue;
                const size_t gp = (size_t)y * W + x;
                if (do_min) {
                    const uint32_t v = s_depth[p];
                    if (v != RTR_EMPTY && v < ld_fresh(depth + gp)) atomicMin(depth + gp, v);
                } else {
                    uint32_t a0, a1, a2, c;
                    sums_of(p, a0, a1, a2, c);
                    if (c) {
                        unsigned long long *a = reinterpret_cast<unsigned long long *>(acc) + 2 * gp;
                        atomicAdd(a + 0, (unsigned long long)a0 | ((unsigned long long)a1 << 32));
                        atomicAdd(a + 1, (unsigned long long)a2 | ((unsigned long long)c << 32));
                    }
                }
            }
            if (MODE == 5) {  // this slice's minima are in the depth buffer
                __threadfence();
                __syncthreads();
                if (tid == 0) atomicAdd(ts_hdr(S) + kHdrSplitDone, 1u);
            }
            if (MODE == 3) {  // the last slice to arrive resolves the tile from the summed accumulators
                __threadfence();
                __syncthreads();
                if (tid == 0) s_flag = atomicAdd(ts_hctr(S) + tile, 1u) == nsub - 1u ? 1u : 0u;
                __syncthreads();
                finish = s_flag != 0u;
                if (finish) {
                    __threadfence();
                    narrow = false;
                    for (int p = tid; p < tpix; p += T) {
                        const int x = tx0 + (p & (tw - 1)), y = ty0 + (p >> g.tw_shift);
                        uint4 v = make_uint4(0u, 0u, 0u, 0u);
                        if (x < W && y < H) {
                            const unsigned long long *a = reinterpret_cast<const unsigned long long *>(acc) + 2 * ((size_t)y * W + x);
                            const unsigned long long lo = __hip_atomic_load(a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            const unsigned long long hi = __hip_atomic_load(a + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            v = make_uint4((uint32_t)lo, (uint32_t)(lo >> 32), (uint32_t)hi, (uint32_t)(hi >> 32));
                        }
                        reinterpret_cast<uint4 *>(s_acc)[p] = v;
                    }
                    __syncthreads();
                }
            } else if (MODE == 2) {
                finish = sub == 0;  // only for the pyramid below (from the global depth tile every slice loaded)
            }
        }
        // write-out: rows of the tile are contiguous in memory
        int over = 0;  // packed accumulators only: some pixel blended more than 257 points
        auto write_out = [&]() {  // (compact, wide layout: the pixels of the half the accumulators hold)
            const int p_lo = (kCompact && cur_half > 0) ? (tpix >> 1) : 0, p_hi = (kCompact && cur_half == 0) ? (tpix >> 1) : tpix;
            for (int p = p_lo + tid; p < p_hi; p += T) {
                int x = tx0 + (p & (tw - 1)), y = ty0 + (p >> g.tw_shift);
                bool inb = x < W && y < H;
                size_t gp = (size_t)y * W + x;
                if (MODE == 0 || MODE == 4) {
                    if (inb) depth[gp] = s_depth[p];
                } else if (MODE == 1) {
                    if (inb) {
                        const uint32_t v = s_depth[p];
                        if (overwrite || v < depth[gp]) depth[gp] = v;
                    }
                }
                if (MODE != 1) {
                    uint32_t a0, a1, a2, c;
                    sums_of(p, a0, a1, a2, c);
                    over |= (narrow && c > 257u) ? 1 : 0;
                    if (MODE == 2) {
                        if (inb) {
                            uint4 o = overwrite ? make_uint4(0u, 0u, 0u, 0u) : reinterpret_cast<uint4 *>(acc)[gp];
                            reinterpret_cast<uint4 *>(acc)[gp] = make_uint4(o.x + a0, o.y + a1, o.z + a2, o.w + c);
                        }
                    } else {
                        if ((MODE == 0 || MODE == 4) && inb && write_acc) reinterpret_cast<uint4 *>(acc)[gp] = make_uint4(a0, a1, a2, c);
                        uint8_t q0, q1, q2;
                        resolve3(a0, a1, a2, c, q0, q1, q2);  // render.cu:147-162
                        s_rgb[3 * p + 0] = q0;
                        s_rgb[3 * p + 1] = q1;
                        s_rgb[3 * p + 2] = q2;
                    }
                }
            }
        };
        if ((!split || (MODE == 3 && finish)) && !no_local && !force_wide) write_out();
        RTR_TSTAMP(5);
        if (((MODE == 0 || MODE == 4) && !split) || (MODE == 3 && finish)) {
            // (the barrier the image rows need anyway also carries the verdict on the packed accumulators: an
            // unsplit tile of a whole frame only ever OVERWRITES memory, so writing it out twice is harmless)
            if ((MODE == 0 || MODE == 4) && (narrow || force_wide)) {
                if (__syncthreads_or(over | (force_wide ? 1 : 0))) {  // rare: redo the tile with the wide layout
                    narrow = false;
#pragma unroll 1
                    for (int half = 0; half < (kCompact ? 2 : 1); ++half) {
                        if (kCompact) {
                            cur_half = half;
                            // (this rare path must not shape the kernel: everything a one-batch tile derives from its
                            // registers is invariant over the two halves, and hoisted out of this loop it spilled)
#pragma unroll
                            for (int k = 0; k < TB; ++k) asm volatile("" : "+v"(r[k]));
                        }
                        for (int p = tid; p < (kCompact ? 2 : 4) * tpix; p += T) s_acc[p] = 0;
                        __syncthreads();
                        accumulate(false);
                        __syncthreads();
                        write_out();
                        __syncthreads();
                    }
                }
            } else {
                __syncthreads();
            }
            // image rows of the tile as dwords when the row segment is whole and 4-byte aligned
            const int row_dw = (3 * tw) >> 2;  // 24 or 48 dwords per tile row
            const bool fast = (tx0 + tw <= W) && ((W & 3) == 0);
            if (fast) {
                const uint32_t *s_rgb32 = reinterpret_cast<const uint32_t *>(s_rgb);
                for (int q = tid; q < row_dw * kTileH; q += T) {
                    int rr = q / row_dw, dw = q - rr * row_dw, y = ty0 + rr;
                    if (y < H) reinterpret_cast<uint32_t *>(img + ((size_t)y * W + tx0) * 3)[dw] = s_rgb32[rr * row_dw + dw];
                }
            } else {
                for (int q = tid; q < 3 * tpix; q += T) {
                    int p = q / 3, ch = q - 3 * p;
                    int x = tx0 + (p & (tw - 1)), y = ty0 + (p >> g.tw_shift);
                    if (x < W && y < H) img[((size_t)y * W + x) * 3 + ch] = s_rgb[q];
                }
            }
            RTR_TSTAMP(6);
            if (pyr.enable)  // s_acc is free now (the colours were resolved into s_rgb before the barrier)
                tile_pyramid(s_depth, s_acc, g, tx0, ty0, pyr.L, pyr.n_eff_rows, pyr.part_min, pyr.part_max, tile, tid, T);
            RTR_TSTAMP(7);
        }
        if (MODE == 2 && pyr.enable && finish) {  // sharded frames: s_depth holds the GLOBAL minimum of the tile here
            __syncthreads();                      // the sums have been read out of s_acc
            tile_pyramid(s_depth, s_acc, g, tx0, ty0, pyr.L, pyr.n_eff_rows, pyr.part_min, pyr.part_max, tile, tid, T);
        }
        if (tile_wg) break;
    }
}

template <int MODE>
__global__ __launch_bounds__(MODE == 0 ? kTileThreadsCompact : kTileThreads, MODE == 0 ? RTR_TILE0_WAVES : RTR_TILE_WAVES) void k_tile(TileStore S, TileGeom g, int W, int H, float window,
                                                        uint32_t *__restrict__ depth, uint32_t *__restrict__ acc,
                                                        uint8_t *__restrict__ img, int write_acc, TilePyr pyr,
                                                        Sliced dsl) {
    tile_body<MODE>(S, g, W, H, window, depth, acc, img, write_acc, pyr, dsl);
}

// The slices of a whole frame's split tiles, both phases in ONE launch (ordinary frames have no split tile: every
// workgroup leaves after one scalar load -- as two launches, modes 5 and 3, the empty pair cost ~9 us of a 64 us
// C2 frame).  Between the phases a workgroup waits until EVERY slice's minima are in the depth buffer.  The slices
// are taken from a queue, so whoever waits has seen the queue empty: each outstanding slice is held by a workgroup
// that is running, and the wait ends whatever else shares the chip and however few of the 256 workgroups fit on
// it at once (CU masks of option "tail_cus").  It is bounded all the same (tile-store error 8 instead of a hung
// queue).  T1's epilogue zeroes the three words.
__global__ __launch_bounds__(kTileThreads) void k_tile_split(TileStore S, TileGeom g, int W, int H, float window,
                                                              uint32_t *__restrict__ depth, uint32_t *__restrict__ acc,
                                                              uint8_t *__restrict__ img, int write_acc, TilePyr pyr,
                                                              Sliced nosl /* (all zero; a kernel argument so that it stays out of scratch) */) {
    const uint32_t n_split = (uint32_t)__builtin_amdgcn_readfirstlane((int)ts_hdr(S)[kHdrSplitItems]);
    if (n_split == 0u) return;  // (grid-uniform)
    TilePyr none{};
    none.enable = 0;
    tile_body<5>(S, g, W, H, window, depth, acc, img, 0, none, nosl);
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t *const done = ts_hdr(S) + kHdrSplitDone;
        int polls = 0;
        while (__hip_atomic_load(done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < n_split) {
            if (++polls > (1 << 22)) {
                store_error_now(S, 8u);
                break;
            }
            __builtin_amdgcn_s_sleep(8);
        }
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // the second phase reads the depth buffer with plain loads
    tile_body<3>(S, g, W, H, window, depth, acc, img, write_acc, pyr, nosl);
}

constexpr int kGridCacheDevices = 64;
void launch_project_bin(hipStream_t s, const Cloud &c, const Proj &P, int W, int H, const TileStore &S,
                        const float *bounds, int clear_split, int phases, int xp, hipEvent_t ev_start, hipEvent_t ev_stop) {
    uint64_t n4 = (c.n + 3) / 4;
    if (n4 == 0) {
        if (clear_split & 8) return;  // (a lean frame: the stream counters are zero and stay so)
        hipExtLaunchKernelGGL(k_bin_empty, dim3(1), dim3(kBlock), 0, s, ev_start, ev_stop, 0, W, H, S, clear_split);
        return;
    }
    // (hipExtLaunchKernelGGL with null events is a plain launch; with events the dispatch packet itself carries
    // the start / stop time stamps: no extra packets around the kernel, unlike hipEventRecord pairs)
    const bool packed = c.pk.hdr != nullptr;
    const LaneTest lt = lane_test_consts(P, W, H, c.absmax);
    const dim3 block(kBlock);
    const float4 *x = packed ? (const float4 *)c.pk.hdr : (const float4 *)c.x;
    const float4 *y = packed ? (const float4 *)c.pk.planes : (const float4 *)c.y;
    const float4 *z = packed ? (const float4 *)c.pk.planes_b : (const float4 *)c.z;
    const uint4 *col = (const uint4 *)c.rgba;
    // The default grid is what is RESIDENT at once (the kernel is a grid-stride loop over windows of the cloud: a
    // workgroup that has to wait for a slot runs the whole loop as a second round).  A packed chunk is ~1.3 KB in flight
    // per wave instead of 3 KB, so the packed kernels take a fifth workgroup per CU when their registers admit it (<= 96:
    // a build at 100 registers launched with the fixed 1280 of before ran 170 instead of 140 us); asked of the runtime
    // once per kernel.
    // (cached per kernel AND device: contexts on different devices must not share the first caller's CU count; the
    // table is written with relaxed atomics -- two threads that race compute the same value)
    int dev_now = 0;
    if (hipGetDevice(&dev_now) != hipSuccess || dev_now < 0 || dev_now >= kGridCacheDevices) dev_now = kGridCacheDevices;
    auto default_grid = [&](auto kernel, int *cache) -> int {
        int cached = dev_now < kGridCacheDevices ? __atomic_load_n(&cache[dev_now], __ATOMIC_RELAXED) : 0;
        if (cached == 0) {
            int per_cu = 0, cus = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, kBlock, 0) != hipSuccess ||
                hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev_now < kGridCacheDevices ? dev_now : 0) != hipSuccess ||
                per_cu < 1 || cus < 1)
                cached = kDefaultPointGrid;
            else
                cached = cus * (per_cu < 4 ? per_cu : (packed && per_cu >= 5 ? 5 : 4));
            if (dev_now < kGridCacheDevices) __atomic_store_n(&cache[dev_now], cached, __ATOMIC_RELAXED);
        }
        return cached;
    };
#define RTR_T1(CULL, GROUPS, PACKED)                                                                                          \
    do {                                                                                                                      \
        static int cached_grid[kGridCacheDevices] = {0};                                                                      \
        const dim3 grid(point_grid(n4, c.grid == kDefaultPointGrid ? default_grid(k_project_bin<CULL, GROUPS, PACKED>, cached_grid) : c.grid)); \
        hipExtLaunchKernelGGL((k_project_bin<CULL, GROUPS, PACKED>), grid, block, 0, s, ev_start, ev_stop, 0, x, y, z, col,    \
                              (uint32_t)n4, P, W, H, S, CULL ? bounds : (packed ? nullptr : c.spread), clear_split,          \
                              (uint32_t)phases, xp, lt);                                                                      \
    } while (0)
    if (bounds) {
        if (packed) RTR_T1(true, true, true); else RTR_T1(true, true, false);
    } else if (c.incoherent) {
        if (packed) RTR_T1(false, false, true); else RTR_T1(false, false, false);
    } else {
        if (packed) RTR_T1(false, true, true); else RTR_T1(false, true, false);
    }
#undef RTR_T1
}

// bounding box of every 256-point chunk (the unit one wave of T1 handles per iteration):
// bounds[6 c .. 6 c + 5] = min x, y, z, max x, y, z; NaN padding is ignored.
// spread[c] = the chunk's lane spread (Cloud::spread): max over lanes, axes and k = 1..3 of |v[4 l + k] - v[4 l]|,
// rounded up; +inf when the chunk holds a NaN / infinite / huge coordinate (such a chunk never takes T1's lane test).
__global__ __launch_bounds__(kBlock) void k_chunk_bounds(const float4 *__restrict__ x4, const float4 *__restrict__ y4,
                                                         const float4 *__restrict__ z4, uint64_t n4,
                                                         float *__restrict__ bounds, float *__restrict__ spread) {
    const uint64_t nchunks = (n4 + 63) / 64;
    const int lane = threadIdx.x & 63;
    const float inf = __uint_as_float(0x7F800000u);
    for (uint64_t c = ((uint64_t)blockIdx.x * kBlock + threadIdx.x) >> 6; c < nchunks;
         c += ((uint64_t)gridDim.x * kBlock) >> 6) {
        float lo[3] = {inf, inf, inf}, hi[3] = {-inf, -inf, -inf};
        float sp = 0.f;
        uint64_t i = c * 64 + lane;
        if (i < n4) {
            const float4 v[3] = {x4[i], y4[i], z4[i]};
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                lo[k] = fminf(fminf(v[k].x, v[k].y), fminf(fminf(v[k].z, v[k].w), lo[k]));  // fminf ignores NaN
                hi[k] = fmaxf(fmaxf(v[k].x, v[k].y), fmaxf(fmaxf(v[k].z, v[k].w), hi[k]));
                const float d1 = fabsf(v[k].y - v[k].x), d2 = fabsf(v[k].z - v[k].x), d3 = fabsf(v[k].w - v[k].x);
                const float a = fmaxf(fmaxf(fabsf(v[k].x), fabsf(v[k].y)), fmaxf(fabsf(v[k].z), fabsf(v[k].w)));
                const bool sane = (v[k].x == v[k].x) && (v[k].y == v[k].y) && (v[k].z == v[k].z) && (v[k].w == v[k].w) && a <= 1e30f;
                sp = sane ? fmaxf(sp, fmaxf(d1, fmaxf(d2, d3))) : inf;
            }
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                lo[k] = fminf(lo[k], __shfl_xor(lo[k], off, 64));
                hi[k] = fmaxf(hi[k], __shfl_xor(hi[k], off, 64));
            }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) sp = fmaxf(sp, __shfl_xor(sp, off, 64));
        if (lane == 0) {
            float *b = bounds + 6 * c;
            b[0] = lo[0]; b[1] = lo[1]; b[2] = lo[2];
            b[3] = hi[0]; b[4] = hi[1]; b[5] = hi[2];
            // (one rounding per difference: a relative 2^-23 up covers it; the test's slack adds more)
            if (spread) spread[c] = sp * 1.000001f;
        }
    }
}

void launch_chunk_bounds(hipStream_t s, const Cloud &c, float *bounds, float *spread) {
    uint64_t n4 = (c.n + 3) / 4;
    if (n4 == 0) return;
    uint64_t blocks = ((n4 + 63) / 64 + 3) / 4;
    hipLaunchKernelGGL(k_chunk_bounds, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(kBlock), 0, s,
                       (const float4 *)c.x, (const float4 *)c.y, (const float4 *)c.z, n4, bounds, spread);
}


// PackedXyz: measure (one wave per chunk: unsigned min / max of the bit patterns -> base and width per axis)
__device__ __forceinline__ void chunk_bits(const uint4 *__restrict__ x4, const uint4 *__restrict__ y4,
                                           const uint4 *__restrict__ z4, uint64_t n4, uint64_t c, int lane, uint32_t v[3][4]) {
    uint64_t i = c * 64 + lane;
    i = i < n4 ? i : n4 - 1;  // lanes past the end repeat the last quad (T1 masks them): they widen nothing
    const uint4 q[3] = {x4[i], y4[i], z4[i]};
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        v[a][0] = q[a].x; v[a][1] = q[a].y; v[a][2] = q[a].z; v[a][3] = q[a].w;
    }
}
__global__ __launch_bounds__(kBlock) void k_pack_measure(const uint4 *__restrict__ x4, const uint4 *__restrict__ y4,
                                                         const uint4 *__restrict__ z4, uint64_t n4, uint4 *__restrict__ hdr,
                                                         uint32_t *__restrict__ chunk_planes) {
    const uint64_t nchunks = (n4 + 63) / 64;
    const int lane = threadIdx.x & 63;
    for (uint64_t c = ((uint64_t)blockIdx.x * kBlock + threadIdx.x) >> 6; c < nchunks; c += ((uint64_t)gridDim.x * kBlock) >> 6) {
        uint32_t v[3][4], diff[3], first[3];
        chunk_bits(x4, y4, z4, n4, c, lane, v);
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            first[a] = (uint32_t)__shfl((int)v[a][0], 0, 64);
            diff[a] = (v[a][0] ^ first[a]) | (v[a][1] ^ first[a]) | (v[a][2] ^ first[a]) | (v[a][3] ^ first[a]);
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) diff[a] |= (uint32_t)__shfl_xor((int)diff[a], off, 64);
        }
        if (lane == 0) {
            uint32_t w[3], base[3];
#pragma unroll
            for (int a = 0; a < 3; ++a) {  // bits below the common prefix of the chunk's 256 bit patterns
                const uint32_t nb = diff[a] == 0u ? 0u : 32u - (uint32_t)__clz((int)diff[a]);
                w[a] = nb > kPackMaxBits ? 32u : nb;  // (a lane's shift + 4 b bits must fit its 16-byte load)
                base[a] = w[a] == 32u ? 0u : (first[a] >> w[a]) << w[a];
            }
            const uint32_t wide = (w[0] == 32u || w[1] == 32u || w[2] == 32u) ? kPackWideFlag : 0u;
            hdr[2 * c] = make_uint4(base[0], base[1], base[2], w[0] | (w[1] << 6) | (w[2] << 12) | wide);
            chunk_planes[c] = w[0] + w[1] + w[2];  // in units of 32 bytes
        }
    }
}
// exclusive scan of the chunks' plane counts -> hdr[2 c + 1]; one workgroup (a one-off at upload: ~1 ms per 1e8 points)
__global__ __launch_bounds__(512) void k_pack_scan(const uint32_t *__restrict__ chunk_planes, uint64_t nchunks,
                                                   uint4 *__restrict__ hdr, uint64_t *__restrict__ total_planes,
                                                   const float *__restrict__ spread) {
    __shared__ uint32_t s_w[8];
    uint64_t carry = 0;
    for (uint64_t c0 = 0; c0 < nchunks; c0 += 512) {
        const uint64_t c = c0 + threadIdx.x;
        const uint32_t v = c < nchunks ? chunk_planes[c] : 0u;
        uint32_t tot = 0;
        const uint32_t incl = block_scan(v, s_w, tot);
        if (c < nchunks) {
            const uint64_t off = carry + (incl - v);
            // (.z: the chunk's lane spread, +inf -- no lane test -- when it was not measured)
            hdr[2 * c + 1] = make_uint4((uint32_t)off, (uint32_t)(off >> 32), spread ? __float_as_uint(spread[c]) : 0x7F800000u, 0u);
        }
        carry += tot;
    }
    if (threadIdx.x == 0) *total_planes = carry;
}
__global__ __launch_bounds__(kBlock) void k_pack_write(const uint4 *__restrict__ x4, const uint4 *__restrict__ y4,
                                                       const uint4 *__restrict__ z4, uint64_t n4, const uint4 *__restrict__ hdr,
                                                       uint32_t *__restrict__ planes_a, uint32_t *__restrict__ planes_b) {
    // (a one-off at upload.)  The two streams of an axis block are assembled in LDS -- every lane ORs its first value in
    // at bit b l of the A stream and its other three at bit 3 b l of the B stream -- and leave as whole dwords, coalesced.
    __shared__ uint32_t s_blk[kBlock / 64][256 + 8];
    const uint64_t nchunks = (n4 + 63) / 64;
    const int lane = threadIdx.x & 63;
    uint32_t *const blk_a = s_blk[threadIdx.x >> 6], *const blk_b = blk_a + 64 + 4;  // (A: <= 64 dwords, B: <= 192)
    for (uint64_t c = ((uint64_t)blockIdx.x * kBlock + threadIdx.x) >> 6; c < nchunks; c += ((uint64_t)gridDim.x * kBlock) >> 6) {
        uint32_t v[3][4];
        chunk_bits(x4, y4, z4, n4, c, lane, v);
        const uint4 h0 = hdr[2 * c], h1 = hdr[2 * c + 1];
        const uint64_t off = (((uint64_t)h1.y) << 32) | (uint64_t)h1.x;  // 32-byte units over both streams
        uint32_t *pa = planes_a + off * 2, *pb = planes_b + off * 6;
#pragma unroll 1
        for (int a = 0; a < 3; ++a) {
            const uint32_t b = (h0.w >> (6 * a)) & 63u;  // wave-uniform
            if (b == 0u) continue;
            const uint32_t ndw_a = 2u * b, ndw_b = 6u * b;  // dwords of the two streams
            for (uint32_t j = (uint32_t)lane; j < ndw_a + 2u; j += 64u) blk_a[j] = 0u;
            for (uint32_t j = (uint32_t)lane; j < ndw_b + 4u; j += 64u) blk_b[j] = 0u;
            __builtin_amdgcn_wave_barrier();
            const uint32_t m = b == 32u ? 0xFFFFFFFFu : ((1u << b) - 1u);
            {  // first value
                const uint32_t bit = b * (uint32_t)lane, dw = bit >> 5, sh = bit & 31u;
                const unsigned long long w = (unsigned long long)(v[a][0] & m) << sh;
                if ((uint32_t)w) atomicOr(&blk_a[dw], (uint32_t)w);
                if ((uint32_t)(w >> 32)) atomicOr(&blk_a[dw + 1], (uint32_t)(w >> 32));
            }
            {  // the other three
                unsigned __int128 blob = 0;
#pragma unroll
                for (int k = 1; k < 4; ++k) blob |= (unsigned __int128)(v[a][k] & m) << (b * (uint32_t)(k - 1));
                const uint32_t bit = 3u * b * (uint32_t)lane, dw = bit >> 5, sh = bit & 31u;
                // (the blob, 3 b <= 96 bits, shifted by sh < 32 bits, spans at most four dwords)
                const uint32_t lo32[3] = {(uint32_t)blob, (uint32_t)(blob >> 32), (uint32_t)(blob >> 64)};
                uint32_t carry = 0;
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const uint32_t w = (lo32[j] << sh) | carry;
                    carry = sh ? (lo32[j] >> (32u - sh)) : 0u;
                    if (w) atomicOr(&blk_b[dw + j], w);
                }
                if (carry) atomicOr(&blk_b[dw + 3], carry);
            }
            __builtin_amdgcn_wave_barrier();
            for (uint32_t j = (uint32_t)lane; j < ndw_a; j += 64u) pa[j] = blk_a[j];
            for (uint32_t j = (uint32_t)lane; j < ndw_b; j += 64u) pb[j] = blk_b[j];
            __builtin_amdgcn_wave_barrier();
            pa += ndw_a;
            pb += ndw_b;
        }
    }
}
__global__ __launch_bounds__(kBlock) void k_pack_verify(const uint4 *__restrict__ x4, const uint4 *__restrict__ y4,
                                                        const uint4 *__restrict__ z4, uint64_t n4, const uint4 *__restrict__ hdr,
                                                        const uint32_t *__restrict__ planes, const uint32_t *__restrict__ planes_b,
                                                        unsigned long long *mismatches) {
    const uint64_t nchunks = (n4 + 63) / 64;
    const int lane = threadIdx.x & 63;
    unsigned long long bad = 0;
    for (uint64_t c = ((uint64_t)blockIdx.x * kBlock + threadIdx.x) >> 6; c < nchunks; c += ((uint64_t)gridDim.x * kBlock) >> 6) {
        uint32_t v[3][4];
        chunk_bits(x4, y4, z4, n4, c, lane, v);
        const uint4 h0 = hdr[2 * c], h1 = hdr[2 * c + 1];
        const ChunkRawA raw_a = load_chunk_a(planes, h0, h1, lane);
        const ChunkRaw raw = load_chunk_b(planes_b, h0, h1, lane);
        float4 X, Y, Z;
        unpack_chunk(raw_a, raw, h0.w, h0.x, h0.y, h0.z, X, Y, Z, lane);
        const uint32_t got[3][4] = {{__float_as_uint(X.x), __float_as_uint(X.y), __float_as_uint(X.z), __float_as_uint(X.w)},
                                    {__float_as_uint(Y.x), __float_as_uint(Y.y), __float_as_uint(Y.z), __float_as_uint(Y.w)},
                                    {__float_as_uint(Z.x), __float_as_uint(Z.y), __float_as_uint(Z.z), __float_as_uint(Z.w)}};
        if (c * 64 + lane < n4)
#pragma unroll
            for (int k = 0; k < 4; ++k) bad += (got[0][k] != v[0][k] || got[1][k] != v[1][k] || got[2][k] != v[2][k]) ? 1ull : 0ull;
    }
    if (bad) atomicAdd(mismatches, bad);
}

// the SoA arrays back from the packed form (bit for bit: the form is lossless) -- for the calls that read fp32
// coordinates when the context keeps only the packed form resident (rtr_api.hip, ensure_soa)
__global__ __launch_bounds__(kBlock) void k_unpack_soa(const uint4 *__restrict__ hdr, const uint32_t *__restrict__ planes,
                                                       const uint32_t *__restrict__ planes_b,
                                                       uint64_t n4, float4 *__restrict__ x4, float4 *__restrict__ y4,
                                                       float4 *__restrict__ z4) {
    const uint64_t nchunks = (n4 + 63) / 64;
    const int lane = threadIdx.x & 63;
    for (uint64_t c = ((uint64_t)blockIdx.x * kBlock + threadIdx.x) >> 6; c < nchunks; c += ((uint64_t)gridDim.x * kBlock) >> 6) {
        const uint4 h0 = hdr[2 * c], h1 = hdr[2 * c + 1];
        const ChunkRawA raw_a = load_chunk_a(planes, h0, h1, lane);
        const ChunkRaw raw = load_chunk_b(planes_b, h0, h1, lane);
        float4 X, Y, Z;
        unpack_chunk(raw_a, raw, h0.w, h0.x, h0.y, h0.z, X, Y, Z, lane);
        const uint64_t i = c * 64 + lane;
        if (i < n4) x4[i] = X, y4[i] = Y, z4[i] = Z;
    }
}

static unsigned pack_grid(uint64_t n4) {
    const uint64_t blocks = ((n4 + 63) / 64 + 3) / 4;
    return (unsigned)(blocks < 8192 ? (blocks ? blocks : 1) : 8192);
}
void pack_measure(hipStream_t s, const Cloud &c, uint4 *hdr, uint32_t *chunk_planes, uint64_t *total_planes) {
    const uint64_t n4 = (c.n + 3) / 4;
    if (n4 == 0) return;
    hipLaunchKernelGGL(k_pack_measure, dim3(pack_grid(n4)), dim3(kBlock), 0, s, (const uint4 *)c.x, (const uint4 *)c.y,
                       (const uint4 *)c.z, n4, hdr, chunk_planes);
    hipLaunchKernelGGL(k_pack_scan, dim3(1), dim3(512), 0, s, chunk_planes, (n4 + 63) / 64, hdr, total_planes, c.spread);
}
void pack_write(hipStream_t s, const Cloud &c, const uint4 *hdr, uint32_t *planes, uint32_t *planes_b) {
    const uint64_t n4 = (c.n + 3) / 4;
    if (n4 == 0) return;
    hipLaunchKernelGGL(k_pack_write, dim3(pack_grid(n4)), dim3(kBlock), 0, s, (const uint4 *)c.x, (const uint4 *)c.y,
                       (const uint4 *)c.z, n4, hdr, planes, planes_b);
}
void pack_verify(hipStream_t s, const Cloud &c, const uint4 *hdr, const uint32_t *planes, const uint32_t *planes_b, uint64_t *mismatches) {
    const uint64_t n4 = (c.n + 3) / 4;
    if (n4 == 0) return;
    hipLaunchKernelGGL(k_pack_verify, dim3(pack_grid(n4)), dim3(kBlock), 0, s, (const uint4 *)c.x, (const uint4 *)c.y,
                       (const uint4 *)c.z, n4, hdr, planes, planes_b, (unsigned long long *)mismatches);
}

void unpack_to_soa(hipStream_t s, const PackedXyz &pk, uint64_t n, float *x, float *y, float *z) {
    const uint64_t n4 = (n + 3) / 4;
    if (n4 == 0) return;
    hipLaunchKernelGGL(k_unpack_soa, dim3(pack_grid(n4)), dim3(kBlock), 0, s, pk.hdr, pk.planes, pk.planes_b, n4, (float4 *)x, (float4 *)y, (float4 *)z);
}

void launch_tile(hipStream_t s, int mode, int W, int H, const TileStore &S, float window, uint32_t *depth,
                 uint32_t *acc, uint8_t *img, int write_acc, const TilePyr *pyr, const Sliced *depth_slices) {
    TileGeom g = tile_geom(W, H);
    Sliced nosl{};
    nosl.chunk = 0;
    nosl.peers = 0;
    nosl.out = nullptr;
    nosl.rank = 0;
    nosl.tab = nullptr;
    size_t tpix = (size_t)32 << g.tw_shift;
    size_t lds = (mode == 1 ? tpix : 5 * tpix) * sizeof(uint32_t) + ((mode == 0 || mode == 3 || mode == 4) ? 3 * tpix : 0);
    TilePyr none{};
    none.enable = 0;
    const dim3 grid(g.ntiles + kHeavyExtra), grid1(g.ntiles + kHeavyExtra + 1), block(kTileThreads);
    if (mode == 0) {  // (the tiles + one workgroup for the next frame's launch order; split tiles' slices: k_tile_split)
        const size_t lds0 = 3 * tpix * sizeof(uint32_t) + 3 * tpix;
        // bit 5 (workgroup b takes tile b) is granted when the whole launch is resident at once: asked of the runtime once
        // per device and tile size
        int flags = write_acc & (1 | 8 | 16 | 64);  // (8, 16: lean frame, parity; 64: first batch before the counters)
        if (write_acc & 32) {
            static int resident[kGridCacheDevices][2] = {{0}};
            int dev_now = 0;
            if (hipGetDevice(&dev_now) == hipSuccess && dev_now >= 0 && dev_now < kGridCacheDevices) {
                int cap = __atomic_load_n(&resident[dev_now][g.tw_shift - 5], __ATOMIC_RELAXED);
                if (cap == 0) {
                    int per_cu = 0, cus = 0;
                    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_tile<0>, kTileThreadsCompact, lds0) != hipSuccess ||
                        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev_now) != hipSuccess || per_cu < 1 || cus < 1)
                        cap = -1;
                    else
                        cap = per_cu * cus;
                    __atomic_store_n(&resident[dev_now][g.tw_shift - 5], cap, __ATOMIC_RELAXED);
                }
                if (cap >= g.ntiles + 1) flags |= 32;
            }
        }
        hipLaunchKernelGGL(k_tile<0>, dim3(g.ntiles + 1), dim3(kTileThreadsCompact), lds0, s, S, g, W, H, window, depth, acc, img, flags,
                           pyr ? *pyr : none, nosl);
    }
    else if (mode == 3)  // the split tiles' slices, min phase then second phase: every workgroup leaves at once on ordinary frames
        hipLaunchKernelGGL(k_tile_split, dim3(kSplitGrid), block, 5 * tpix * sizeof(uint32_t) + 3 * tpix, s, S, g, W, H, window, depth,
                           acc, img, write_acc & 1, pyr ? *pyr : none, nosl);
    else if (mode == 4)  // owner-computes sharded frame: one workgroup per tile, the segment table behind the tile buffers
        hipLaunchKernelGGL(k_tile<4>, dim3(g.ntiles), block, lds + tpix * sizeof(uint32_t) + kSegCap4 * 16, s, S, g, W, H, window, depth,
                           acc, img, write_acc & 1, pyr ? *pyr : none, *depth_slices);
    else if (mode == 1)
        hipLaunchKernelGGL(k_tile<1>, grid1, block, lds, s, S, g, W, H, window, depth, acc, img, write_acc & (6 | 16), none, nosl);
    else  // mode 2 always writes the accumulators; bit 1 of write_acc = overwrite; pyr: also emit the pyramid
        hipLaunchKernelGGL(k_tile<2>, grid, block, lds, s, S, g, W, H, window, depth, acc, img, 1 | (write_acc & 6),
                           pyr ? *pyr : none, depth_slices ? *depth_slices : nosl);
}

// ---------------------------------------------------------------------------------
// A6 resolvePass (render.cu:132-163): truncating u32 division, 0 where count == 0.
// Four pixels per thread so the 12 output bytes go out as three dword stores.
__global__ __launch_bounds__(kBlock) void k_resolve(const uint4 *__restrict__ acc, uint32_t *__restrict__ img32,
                                                    uint8_t *__restrict__ img, size_t npix) {
    size_t q = (size_t)blockIdx.x * kBlock + threadIdx.x;  // pixel quad
    size_t base = q * 4;
    if (base >= npix) return;
    uint32_t out[12];
    int cnt = (npix - base) < 4 ? (int)(npix - base) : 4;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        uint4 a = (k < cnt) ? acc[base + k] : make_uint4(0, 0, 0, 0);
        uint32_t c = a.w;
        out[3 * k + 0] = c ? a.x / c : 0u;
        out[3 * k + 1] = c ? a.y / c : 0u;
        out[3 * k + 2] = c ? a.z / c : 0u;
    }
    if (cnt == 4) {
        img32[q * 3 + 0] = (out[0] & 0xFF) | ((out[1] & 0xFF) << 8) | ((out[2] & 0xFF) << 16) | ((out[3] & 0xFF) << 24);
        img32[q * 3 + 1] = (out[4] & 0xFF) | ((out[5] & 0xFF) << 8) | ((out[6] & 0xFF) << 16) | ((out[7] & 0xFF) << 24);
        img32[q * 3 + 2] =
            (out[8] & 0xFF) | ((out[9] & 0xFF) << 8) | ((out[10] & 0xFF) << 16) | ((out[11] & 0xFF) << 24);
    } else {
        for (int k = 0; k < 3 * cnt; ++k) img[base * 3 + k] = (uint8_t)out[k];
    }
}

void launch_resolve(hipStream_t s, const uint32_t *acc, uint8_t *img, size_t npix) {
    size_t quads = (npix + 3) / 4;
    int grid = (int)((quads + kBlock - 1) / kBlock);
    hipLaunchKernelGGL(k_resolve, dim3(grid), dim3(kBlock), 0, s, (const uint4 *)acc, (uint32_t *)img, img, npix);
}

// ---------------------------------------------------------------------------------
// depth-heuristic prefilter (project_cloud.cu:28-187, launch sequence :331-392).
// The reference runs ~20 dependent launches, 12 malloc/free and ~20 device syncs per
// frame; here it is 5 launches on pre-allocated levels:
//   F1 k_pyramid : all min-pool levels (A8) of a 32x32 depth tile through LDS + per-tile
//                  min / max partials (A12)
//   F2..F4 k_up  : per level: Laplacian of the parent (A9, recomputed per child), compare
//                  (A10) and in-place bilinear fill of rejected pixels (A11)
//   F5 k_final   : level-0 compare + removeMask + fp16 tensor (A13), four pixels a thread
// Arithmetic is op-for-op that of oracle/rtr_oracle.c.

// F1.  A8 reduce (project_cloud.cu:28-53) for every level at once.  Level i pixel (x, y)
// exists iff x < w[i], y < h[i] (dims halve with integer division), and then all four
// children exist.  A12 (render.cu:168-240): min / max of the depth BIT PATTERNS over the
// first n_eff pixels, sentinel skipped; per-tile partials, reduced by F2's workgroup 0.
__global__ __launch_bounds__(kBlock) void k_pyramid(FilterLevels L, int tiles_x, uint32_t n_eff_rows,
                                                    uint32_t *__restrict__ part_min, uint32_t *__restrict__ part_max) {
    __shared__ float s1[256], s2[64], s3[16];
    __shared__ uint32_t s_mm[8];
    const int t = threadIdx.x;
    const int tx = blockIdx.x % tiles_x, ty = blockIdx.x / tiles_x;
    uint32_t lo = 0xFFFFFFFFu, hi = 0u;
    {
        const int lx = t & 15, ly = t >> 4;
        const int gx = tx * 16 + lx, gy = ty * 16 + ly;
        float v = 0.0f;
        if (L.levels >= 1 && gx < L.w[1] && gy < L.h[1]) {
            const float *r0 = L.lv[0] + (size_t)(2 * gy) * L.w[0] + 2 * gx;
            float2 a = *reinterpret_cast<const float2 *>(r0);
            float2 b = *reinterpret_cast<const float2 *>(r0 + L.w[0]);
            v = min2(min2(a.x, a.y), min2(b.x, b.y));
            L.lv[1][(size_t)gy * L.w[1] + gx] = v;
            if ((uint32_t)(2 * gy) < n_eff_rows) {  // n_eff_rows is even: both rows or none
                const uint32_t q[4] = {__float_as_uint(a.x), __float_as_uint(a.y), __float_as_uint(b.x),
                                       __float_as_uint(b.y)};
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (q[k] != RTR_EMPTY) {
                        lo = q[k] < lo ? q[k] : lo;
                        hi = q[k] > hi ? q[k] : hi;
                    }
            }
        }
        s1[t] = v;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        uint32_t ol = __shfl_xor(lo, off, 64), oh = __shfl_xor(hi, off, 64);
        lo = ol < lo ? ol : lo;
        hi = oh > hi ? oh : hi;
    }
    if ((t & 63) == 0) {
        s_mm[t >> 6] = lo;
        s_mm[4 + (t >> 6)] = hi;
    }
    __syncthreads();
    if (t == 0) {
        uint32_t a = s_mm[0], b = s_mm[4];
        for (int k = 1; k < 4; ++k) {
            a = s_mm[k] < a ? s_mm[k] : a;
            b = s_mm[4 + k] > b ? s_mm[4 + k] : b;
        }
        part_min[blockIdx.x] = a;
        part_max[blockIdx.x] = b;
    }
    if (L.levels >= 2 && t < 64) {
        const int lx = t & 7, ly = t >> 3, gx = tx * 8 + lx, gy = ty * 8 + ly;
        float v = 0.0f;
        if (gx < L.w[2] && gy < L.h[2]) {
            const float *c = s1 + (2 * ly) * 16 + 2 * lx;
            v = min2(min2(c[0], c[1]), min2(c[16], c[17]));
            L.lv[2][(size_t)gy * L.w[2] + gx] = v;
        }
        s2[t] = v;
    }
    __syncthreads();
    if (L.levels >= 3 && t < 16) {
        const int lx = t & 3, ly = t >> 2, gx = tx * 4 + lx, gy = ty * 4 + ly;
        float v = 0.0f;
        if (gx < L.w[3] && gy < L.h[3]) {
            const float *c = s2 + (2 * ly) * 8 + 2 * lx;
            v = min2(min2(c[0], c[1]), min2(c[8], c[9]));
            L.lv[3][(size_t)gy * L.w[3] + gx] = v;
        }
        s3[t] = v;
    }
    __syncthreads();
    if (L.levels >= 4 && t < 4) {
        const int lx = t & 1, ly = t >> 1, gx = tx * 2 + lx, gy = ty * 2 + ly;
        if (gx < L.w[4] && gy < L.h[4]) {
            const float *c = s3 + (2 * ly) * 4 + 2 * lx;
            L.lv[4][(size_t)gy * L.w[4] + gx] = min2(min2(c[0], c[1]), min2(c[4], c[5]));
        }
    }
}

// generic single-level reduce for pyramids deeper than 4 levels (levels 5..8)
__global__ __launch_bounds__(kBlock) void k_reduce(const float *__restrict__ hi, float *__restrict__ lo, int w, int h) {
    int idx = blockIdx.x * kBlock + threadIdx.x;
    if (idx >= w * h) return;
    int x = idx % w, y = idx / w;
    const float2 *r0 = (const float2 *)(hi + (size_t)(2 * y) * (2 * w)) + x;
    const float2 *r1 = (const float2 *)(hi + (size_t)(2 * y + 1) * (2 * w)) + x;
    float2 a = *r0, b = *r1;
    lo[idx] = min2(min2(a.x, a.y), min2(b.x, b.y));
}

// The reference tests `cur >= MAX_FLOAT` with MAX_FLOAT = 3.4028e38 (a double literal,
// project_cloud.cu:21,97), i.e. in double precision.  0x7F7FFF8C is the smallest float whose value
// is >= that literal, so the float comparison below is the same predicate for every float input
// (NaN: false either way) without the f64 convert + compare.
__device__ __forceinline__ bool at_max_float(float cur) { return cur >= __uint_as_float(0x7F7FFF8Cu); }

// A pyramid level as the kernels below read it: in memory (row stride = its width) or as a
// window of it staged in LDS.  w / h are the dims the reference's launch sequence uses for the
// level (heights are the doubled TRUNCATED ones, project_cloud.cu:360-361).
struct MemLevel {
    const float *p; int w, h;
    __device__ __forceinline__ float at(int x, int y) const { return p[(size_t)y * w + x]; }
};
struct LdsLevel {
    const float *p; int ox, oy, sw;  // window origin and row stride
    int w, h;
    __device__ __forceinline__ float at(int x, int y) const { return p[(y - oy) * sw + (x - ox)]; }
};

// A9 laplacianKernel (project_cloud.cu:55-79) for ONE low-res pixel: border -> 0, else the
// nine-tap fmaf chain in row-major order (zero-weight corners included) compared with thr.
template <class Lv>
__device__ __forceinline__ bool lap_flag(const Lv &lo, int x, int y, float thr) {
    if (x == 0 || x == lo.w - 1 || y == 0 || y == lo.h - 1) return false;
    float sum = 0.0f;
    sum = fmaf(lo.at(x - 1, y - 1), 0.0f, sum);
    sum = fmaf(lo.at(x, y - 1), 1.0f, sum);
    sum = fmaf(lo.at(x + 1, y - 1), 0.0f, sum);
    sum = fmaf(lo.at(x - 1, y), 1.0f, sum);
    sum = fmaf(lo.at(x, y), -4.0f, sum);
    sum = fmaf(lo.at(x + 1, y), 1.0f, sum);
    sum = fmaf(lo.at(x - 1, y + 1), 0.0f, sum);
    sum = fmaf(lo.at(x, y + 1), 1.0f, sum);
    sum = fmaf(lo.at(x + 1, y + 1), 0.0f, sum);
    return sum > thr;
}

template <class Lv>
__device__ __forceinline__ float lo_px(const Lv &lo, int x, int y) {
    return (x >= 0 && x < lo.w && y >= 0 && y < lo.h) ? lo.at(x, y) : -1.0f;  // project_cloud.cu:81-86
}

// A10 compareImgsKernel (project_cloud.cu:88-126) for one hi-res pixel
template <class Lv>
__device__ __forceinline__ bool keep_px(const Lv &lo, float cur, int x, int y, float strength, float thr) {
    if (at_max_float(cur)) return false;  // MAX_FLOAT, project_cloud.cu:21,97
    const int lx = x >> 1, ly = y >> 1;
    if (lap_flag(lo, lx, ly, thr)) {  // only interior pixels can be flagged: the nine taps are in range
        bool keep = false;
#pragma unroll
        for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
            for (int dx = -1; dx <= 1; ++dx) keep = keep | (cur <= f_mul(lo.at(lx + dx, ly + dy), strength));
        return keep;
    }
    return cur <= f_mul(lo_px(lo, lx, ly), strength);
}

// keep_px for four horizontally adjacent pixels x..x+3 (x % 4 == 0) of one row: they have two
// parents, whose 3x3 neighbourhoods share a 4x3 window of the lower level, so the twelve taps
// (and their products with `strength`) are fetched once and everything is branch-free.
// Bit k of the result = keep_px(lo, d[k], x + k, y, ...).
template <class Lv>
__device__ __forceinline__ uint32_t keep_quad(const Lv &lo, const float d[4], int x, int y, float strength, float thr) {
    const int lx = x >> 1, ly = y >> 1;
    float v[3][4], p[3][4];
    if (lx >= 1 && lx + 2 < lo.w && ly >= 1 && ly + 1 < lo.h) {  // the whole 4x3 window is inside the level
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) v[j][i] = lo.at(lx - 1 + i, ly - 1 + j);
    } else {
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) v[j][i] = lo_px(lo, lx - 1 + i, ly - 1 + j);
    }
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) p[j][i] = f_mul(v[j][i], strength);
    uint32_t bits = 0;
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        const int X = lx + a;
        bool flag = false;
        if (!(X == 0 || X == lo.w - 1 || ly == 0 || ly == lo.h - 1)) {  // A9: the nine taps are in range here
            float sum = 0.0f;
            sum = fmaf(v[0][a], 0.0f, sum);
            sum = fmaf(v[0][a + 1], 1.0f, sum);
            sum = fmaf(v[0][a + 2], 0.0f, sum);
            sum = fmaf(v[1][a], 1.0f, sum);
            sum = fmaf(v[1][a + 1], -4.0f, sum);
            sum = fmaf(v[1][a + 2], 1.0f, sum);
            sum = fmaf(v[2][a], 0.0f, sum);
            sum = fmaf(v[2][a + 1], 1.0f, sum);
            sum = fmaf(v[2][a + 2], 0.0f, sum);
            flag = sum > thr;
        }
#pragma unroll
        for (int k = 2 * a; k < 2 * a + 2; ++k) {
            const float cur = d[k];
            bool any = false;
#pragma unroll
            for (int j = 0; j < 3; ++j)
#pragma unroll
                for (int i = a; i < a + 3; ++i) any = any | (cur <= p[j][i]);
            const bool keep = !at_max_float(cur) & (flag ? any : (cur <= p[1][a + 1]));
            bits |= keep ? (1u << k) : 0u;
        }
    }
    return bits;
}

// A11 resize (project_cloud.cu:128-161): bilinear x2 up-sample of lo at hi-res pixel (x, y)
template <class Lv>
__device__ __forceinline__ float bilerp(const Lv &lo, int x, int y) {
    float inX = f_sub(f_add((float)x, 0.5f) / 2.0f, 0.5f);
    float inY = f_sub(f_add((float)y, 0.5f) / 2.0f, 0.5f);
    int x0 = (int)floorf(inX), x1 = x0 + 1, y0 = (int)floorf(inY), y1 = y0 + 1;
    x0 = x0 < 0 ? 0 : (x0 >= lo.w ? lo.w - 1 : x0);
    x1 = x1 < 0 ? 0 : (x1 >= lo.w ? lo.w - 1 : x1);
    y0 = y0 < 0 ? 0 : (y0 >= lo.h ? lo.h - 1 : y0);
    y1 = y1 < 0 ? 0 : (y1 >= lo.h ? lo.h - 1 : y1);
    float wx = f_sub(inX, (float)x0), wy = f_sub(inY, (float)y0);
    float v0 = fmaf(wx, lo.at(x1, y0), f_mul(f_sub(1.0f, wx), lo.at(x0, y0)));
    float v1 = fmaf(wx, lo.at(x1, y1), f_mul(f_sub(1.0f, wx), lo.at(x0, y1)));
    return fmaf(wy, v1, f_mul(f_sub(1.0f, wy), v0));
}

// F2..F4: level i (lo, lw x lh) against level i-1 (hi, 2lw x 2lh): compare, and where the
// pixel is rejected or empty write the bilinear x2 up-sample of lo in place (A11,
// project_cloud.cu:128-161).  Workgroup 0 of the first call also folds the min / max
// partials of F1 into minmax[2].
__global__ __launch_bounds__(kBlock) void k_up(const float *__restrict__ lo, float *__restrict__ hi, int lw, int lh,
                                               float strength, float thr, const uint32_t *__restrict__ part_min,
                                               const uint32_t *__restrict__ part_max, int nparts,
                                               uint32_t *__restrict__ minmax) {
    if (part_min != nullptr && blockIdx.x == 0) {
        __shared__ uint32_t s_mm[8];
        uint32_t a = 0xFFFFFFFFu, b = 0u;
        for (int k = threadIdx.x; k < nparts; k += kBlock) {
            uint32_t pa = part_min[k], pb = part_max[k];
            a = pa < a ? pa : a;
            b = pb > b ? pb : b;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            uint32_t oa = __shfl_xor(a, off, 64), ob = __shfl_xor(b, off, 64);
            a = oa < a ? oa : a;
            b = ob > b ? ob : b;
        }
        if ((threadIdx.x & 63) == 0) {
            s_mm[threadIdx.x >> 6] = a;
            s_mm[4 + (threadIdx.x >> 6)] = b;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int k = 1; k < 4; ++k) {
                a = s_mm[k] < a ? s_mm[k] : a;
                b = s_mm[4 + k] > b ? s_mm[4 + k] : b;
            }
            a = s_mm[0] < a ? s_mm[0] : a;
            b = s_mm[4] > b ? s_mm[4] : b;
            minmax[0] = a;
            minmax[1] = b;
        }
    }
    const int ow = 2 * lw, oh = 2 * lh;
    int idx = blockIdx.x * kBlock + threadIdx.x;
    if (idx >= ow * oh) return;
    int x = idx % ow, y = idx / ow;
    const MemLevel lv{lo, lw, lh};
    if (keep_px(lv, hi[idx], x, y, strength, thr)) return;
    hi[idx] = bilerp(lv, x, y);
}

__device__ __forceinline__ uint32_t to_half_bits(float f) {
    if (f != f) return 0x7E00u;  // canonical NaN (oracle does the same)
    return __half_as_ushort(__float2half_rn(f));
}
__device__ __forceinline__ float half_round(float f) { return __half2float(__float2half_rn(f)); }
// colour plane value of byte b: half(float(half(b)) / 255.0f) (A13); 256 possible results
__device__ __forceinline__ uint32_t colour_half(uint32_t b) { return to_half_bits(half_round((float)b) / 255.0f); }

// Level-0 compare (A10) + removeMask (A13, project_cloud.cu:163-187) for four horizontally
// adjacent pixels (W % 16 == 0), tensor plane stride W*H (the reference strides by W*H_eff:
// quirk Q3).  Rows >= H_eff never saw the pyramid test: their mask is "non-empty".
template <class Lv>
__device__ __forceinline__ void final_quad(const Lv &l1, float4 d4, uint32_t iw0, uint32_t iw1, uint32_t iw2,
                                           float *__restrict__ depth, uint8_t *__restrict__ img,
                                           uint8_t *__restrict__ mask, uint16_t *__restrict__ tensor, float mn,
                                           float range, int x, int y, size_t idx, size_t npix, bool in_domain,
                                           float strength, float thr, const uint16_t *lut = nullptr) {
    float d[4] = {d4.x, d4.y, d4.z, d4.w};
    uint32_t iw[3] = {iw0, iw1, iw2};
    uint32_t mbits = 0;
    uint32_t th[5][4];
    uint32_t kbits = 0;
    if (in_domain) {
        kbits = keep_quad(l1, d, x, y, strength, thr);
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) kbits |= !at_max_float(d[k]) ? (1u << k) : 0u;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const bool keep = (kbits >> k) & 1u;
        if (!keep) {
            d[k] = -1.0f;
            th[0][k] = th[1][k] = th[2][k] = th[3][k] = 0;
            th[4][k] = 0xBC00u;
        } else {
            mbits |= 0xFFu << (8 * k);
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                int byte = 3 * k + c;
                const uint32_t b = (iw[byte >> 2] >> (8 * (byte & 3))) & 0xFFu;
                th[c][k] = lut ? lut[b] : colour_half(b);
            }
            th[3][k] = 0x3C00u;  // half(float(half(255)) / 255.0f)
            th[4][k] = to_half_bits(half_round(f_sub(d[k], mn)) / range);
        }
    }
    // zero the colour bytes of rejected pixels
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (!((mbits >> (8 * k)) & 1u)) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                int byte = 3 * k + c;
                iw[byte >> 2] &= ~(0xFFu << (8 * (byte & 3)));
            }
        }
    *reinterpret_cast<float4 *>(depth + idx) = make_float4(d[0], d[1], d[2], d[3]);
    uint32_t *ip = reinterpret_cast<uint32_t *>(img + idx * 3);
    ip[0] = iw[0];
    ip[1] = iw[1];
    ip[2] = iw[2];
    *reinterpret_cast<uint32_t *>(mask + idx) = mbits;
#pragma unroll
    for (int c = 0; c < 5; ++c)
        *reinterpret_cast<uint2 *>(tensor + (size_t)c * npix + idx) =
            make_uint2(th[c][0] | (th[c][1] << 16), th[c][2] | (th[c][3] << 16));
}

// F5 of the generic sequence: final_quad over ALL W*H pixels, level 1 read from memory
__global__ __launch_bounds__(kBlock) void k_final(const float *__restrict__ l1, float *__restrict__ depth,
                                                  uint8_t *__restrict__ img, uint8_t *__restrict__ mask,
                                                  uint16_t *__restrict__ tensor, const uint32_t *__restrict__ minmax,
                                                  int W, int H, int lw, int lh, float strength, float thr) {
    const size_t npix = (size_t)W * H, q = (size_t)blockIdx.x * kBlock + threadIdx.x, idx = q * 4;
    if (idx >= npix) return;
    const int x = (int)(idx % W), y = (int)(idx / W);
    const float mn = __uint_as_float(minmax[0]), range = f_sub(__uint_as_float(minmax[1]), mn);
    const float4 d4 = *reinterpret_cast<float4 *>(depth + idx);
    const uint32_t *ip = reinterpret_cast<const uint32_t *>(img + idx * 3);
    final_quad(MemLevel{l1, lw, lh}, d4, ip[0], ip[1], ip[2], depth, img, mask, tensor, mn, range, x, y, idx, npix,
               y < 2 * lh, strength, thr);
}

// The whole prefilter after the pyramid in ONE launch (default four levels): a workgroup owns a
// 64x16 block of the frame and recomputes, in LDS, the part of every level its block depends on.
// k_up(i) only reads level i (final) and rewrites single pixels of level i-1, and a level i-1
// pixel depends on the 3x3 neighbourhood of its parent, so the block needs level 1 on a 34x10
// window, level 2 on 20x8, level 3 on 12x6 and the untouched level 4 on 8x5: ~0.5 % more
// arithmetic instead of three dependent launches and their round trips (the k_up / k_final
// sequence costs 31 us alone and 170 us beside a streaming kernel; the pyramid levels are
// not written back, nothing reads them after the prefilter).
// Window of level i for the 64x16 block at (X0, Y0): origin X0 / 2^i - 1 (level 1) or - 2 (levels
// 2..4), same in y; the extents follow from "parents of the window plus their 3x3 neighbours",
// [(a >> 1) - 1, ((b - 1) >> 1) + 2) for a child window [a, b): 34x10, 20x8, 12x6, 8x5.
// (64 wide so that a wave's stores are whole 128 B lines in every output plane.)
constexpr int kFuseW = 64, kFuseH = 16;
constexpr int kF1x = 34, kF1y = 10, kF2x = 20, kF2y = 8, kF3x = 12, kF3y = 6, kF4x = 8, kF4y = 5;

template <int SX, int SY>
__device__ __forceinline__ void stage_level(float *dst, int ox, int oy, const float *__restrict__ src, int lw, int lh) {
    for (int q = threadIdx.x; q < SX * SY; q += kBlock) {
        const int gx = ox + q % SX, gy = oy + q / SX;
        dst[q] = (gx >= 0 && gx < lw && gy >= 0 && gy < lh) ? src[(size_t)gy * lw + gx] : 0.0f;
    }
}
template <int SX, int SY>
__device__ __forceinline__ void up_level(float *hi, int ox, int oy, int hw, int hh, const LdsLevel &lo, float strength,
                                         float thr) {
    for (int q = threadIdx.x; q < SX * SY; q += kBlock) {
        const int gx = ox + q % SX, gy = oy + q / SX;
        if (gx < 0 || gx >= hw || gy < 0 || gy >= hh) continue;
        if (!keep_px(lo, hi[q], gx, gy, strength, thr)) hi[q] = bilerp(lo, gx, gy);
    }
}

#ifdef RTR_EXPERIMENT
__device__ unsigned long long g_fstamp[2][8];  // time stamps inside two workgroups of k_filter4 (tools/stamps.py)
#define RTR_FSTAMP(k) do { if (threadIdx.x == 0 && (blockIdx.x == 5 || blockIdx.x == 1900)) g_fstamp[blockIdx.x == 5 ? 0 : 1][k] = wall_clock64(); } while (0)
void read_filter_stamps(hipStream_t s, unsigned long long *out16) {
    (void)hipMemcpyFromSymbolAsync(out16, HIP_SYMBOL(g_fstamp), sizeof g_fstamp, 0, hipMemcpyDeviceToHost, s);
}
#else
#define RTR_FSTAMP(k) do { } while (0)
#endif
__global__ __launch_bounds__(kBlock) void k_filter4(const float *__restrict__ g1, const float *__restrict__ g2,
                                                    const float *__restrict__ g3, const float *__restrict__ g4,
                                                    int h4, float *__restrict__ depth, uint8_t *__restrict__ img,
                                                    uint8_t *__restrict__ mask, uint16_t *__restrict__ tensor,
                                                    const uint32_t *__restrict__ part_min,
                                                    const uint32_t *__restrict__ part_max, int nparts,
                                                    uint32_t *__restrict__ minmax, int W, int H, int blocks_x,
                                                    float strength, float thr, Sliced isl,
                                                    const float *__restrict__ depth_src) {
    __shared__ float s1[kF1x * kF1y], s2[kF2x * kF2y], s3[kF3x * kF3y], s4[kF4x * kF4y];
    __shared__ uint32_t s_mm[8];
    __shared__ uint16_t s_lut[256];  // colour byte -> fp16 bits: one IEEE division per thread instead of twelve
    static_assert(kBlock == 256, "one table entry per thread");
    RTR_FSTAMP(0);
    s_lut[threadIdx.x] = (uint16_t)colour_half(threadIdx.x);
    const int X0 = (blockIdx.x % blocks_x) * kFuseW, Y0 = (blockIdx.x / blocks_x) * kFuseH;
    const int t = threadIdx.x, x = X0 + 4 * (t & 15), y = Y0 + (t >> 4);
    const bool inb = x < W && y < H;
    const size_t npix = (size_t)W * H, idx = (size_t)y * W + x;
    // the frame-sized loads go first: everything below overlaps their round trip
    float4 d4 = make_float4(0.f, 0.f, 0.f, 0.f);
    uint32_t iw0 = 0, iw1 = 0, iw2 = 0;
    if (inb) {
        d4 = *reinterpret_cast<const float4 *>(depth_src + idx);
        // sharded frame: the resolved image still lies in the ranks' slices (16-pixel multiples: a quad has one owner)
        const uint8_t *src = isl.chunk ? static_cast<const uint8_t *>(isl.src.p[idx / isl.chunk]) : img;
        const uint32_t *ip = reinterpret_cast<const uint32_t *>(src + idx * 3);
        iw0 = ip[0]; iw1 = ip[1]; iw2 = ip[2];
    }
    const int w1 = W >> 1, w2 = W >> 2, w3 = W >> 3, w4 = W >> 4, h3 = 2 * h4, h2 = 4 * h4, h1 = 8 * h4;
    const int x1 = (X0 >> 1) - 1, y1 = (Y0 >> 1) - 1, x2 = (X0 >> 2) - 2, y2 = (Y0 >> 2) - 2;
    const int x3 = (X0 >> 3) - 2, y3 = (Y0 >> 3) - 2, x4 = (X0 >> 4) - 2, y4 = (Y0 >> 4) - 2;
    stage_level<kF1x, kF1y>(s1, x1, y1, g1, w1, h1);
    stage_level<kF2x, kF2y>(s2, x2, y2, g2, w2, h2);
    stage_level<kF3x, kF3y>(s3, x3, y3, g3, w3, h3);
    stage_level<kF4x, kF4y>(s4, x4, y4, g4, w4, h4);
    RTR_FSTAMP(1);  // frame loads requested, levels staged
    // A12: every workgroup folds the per-tile min / max partials itself (16 KB out of L2) rather than waiting for a fold
    // launch -- behind the level staging, so that its wait is not also the wait for the frame-sized loads above
    uint32_t fa = 0xFFFFFFFFu, fb = 0u;
    {   // (16 bytes per load: 2040 workgroups read these 16 KB at the same moment)
        const int n4p = nparts >> 2;
        const uint4 *const pm4 = reinterpret_cast<const uint4 *>(part_min), *const px4 = reinterpret_cast<const uint4 *>(part_max);
        for (int k = t; k < n4p; k += kBlock) {
            const uint4 a = pm4[k], b = px4[k];
            const uint32_t a2 = a.x < a.y ? a.x : a.y, a3 = a.z < a.w ? a.z : a.w, b2 = b.x > b.y ? b.x : b.y, b3 = b.z > b.w ? b.z : b.w;
            const uint32_t pa = a2 < a3 ? a2 : a3, pb = b2 > b3 ? b2 : b3;
            fa = pa < fa ? pa : fa;
            fb = pb > fb ? pb : fb;
        }
        for (int k = 4 * n4p + t; k < nparts; k += kBlock) {
            const uint32_t pa = part_min[k], pb = part_max[k];
            fa = pa < fa ? pa : fa;
            fb = pb > fb ? pb : fb;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const uint32_t oa = __shfl_xor(fa, off, 64), ob = __shfl_xor(fb, off, 64);
        fa = oa < fa ? oa : fa;
        fb = ob > fb ? ob : fb;
    }
    if ((t & 63) == 0) {
        s_mm[t >> 6] = fa;
        s_mm[4 + (t >> 6)] = fb;
    }
    __syncthreads();
    RTR_FSTAMP(2);  // partials folded
    up_level<kF3x, kF3y>(s3, x3, y3, w3, h3, LdsLevel{s4, x4, y4, kF4x, w4, h4}, strength, thr);
    __syncthreads();
    up_level<kF2x, kF2y>(s2, x2, y2, w2, h2, LdsLevel{s3, x3, y3, kF3x, w3, h3}, strength, thr);
    __syncthreads();
    up_level<kF1x, kF1y>(s1, x1, y1, w1, h1, LdsLevel{s2, x2, y2, kF2x, w2, h2}, strength, thr);
    __syncthreads();
    RTR_FSTAMP(3);  // three up steps
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        fa = s_mm[k] < fa ? s_mm[k] : fa;
        fb = s_mm[4 + k] > fb ? s_mm[4 + k] : fb;
    }
    if (blockIdx.x == 0 && t == 0) {
        minmax[0] = fa;
        minmax[1] = fb;
    }
    const float mn = __uint_as_float(fa), range = f_sub(__uint_as_float(fb), mn);
    if (!inb) return;
    final_quad(LdsLevel{s1, x1, y1, kF1x, w1, h1}, d4, iw0, iw1, iw2, depth, img, mask, tensor, mn, range,
               x, y, idx, npix, y < 2 * h1, strength, thr, s_lut);
    RTR_FSTAMP(4);  // final step issued (its stores are on their way)
}

// A14 applyDepthFilter (project_cloud.cu:331-392)
void launch_filter(hipStream_t s, const FilterLevels &L, uint32_t *depth_bits, uint8_t *img, uint8_t *mask,
                   uint16_t *tensor, uint32_t *minmax, uint32_t *part_min, uint32_t *part_max, int W, int H,
                   float strength, float thr, int pyramid_parts, const Sliced *img_slices, const uint32_t *depth_src) {
    auto blocks = [](size_t n) { return dim3((unsigned)((n + kBlock - 1) / kBlock)); };
    const int nl = L.levels;
    const int h_eff = (H >> nl) << nl;
    const int tiles_x = (W + 31) / 32, tiles_y = (H + 31) / 32;
    int nparts = tiles_x * tiles_y;
    if (nl == 4) {  // default: everything after the pyramid in one launch
        if (pyramid_parts == 0)  // phase call: T4 did not produce the levels and the min / max partials
            hipLaunchKernelGGL(k_pyramid, dim3(nparts), dim3(kBlock), 0, s, L, tiles_x, (uint32_t)h_eff, part_min, part_max);
        const int bx = (W + kFuseW - 1) / kFuseW, by = (H + kFuseH - 1) / kFuseH;
        hipLaunchKernelGGL(k_filter4, dim3(bx * by), dim3(kBlock), 0, s, L.lv[1], L.lv[2], L.lv[3], L.lv[4], L.h[4],
                           (float *)depth_bits, img, mask, tensor, part_min, part_max,
                           pyramid_parts > 0 ? pyramid_parts : nparts, minmax, W, H, bx, strength, thr,
                           img_slices ? *img_slices : Sliced{}, (const float *)(depth_src ? depth_src : depth_bits));
        return;
    }
    {
        FilterLevels L4 = L;
        if (L4.levels > 4) L4.levels = 4;
        hipLaunchKernelGGL(k_pyramid, dim3(nparts), dim3(kBlock), 0, s, L4, tiles_x, (uint32_t)h_eff, part_min, part_max);
    }
    for (int i = 5; i <= nl; ++i)
        hipLaunchKernelGGL(k_reduce, blocks((size_t)L.w[i] * L.h[i]), dim3(kBlock), 0, s, L.lv[i - 1], L.lv[i], L.w[i],
                           L.h[i]);
    int cw = L.w[nl], ch = L.h[nl];  // the reference doubles the TRUNCATED dims (project_cloud.cu:360-361)
    bool first = true;
    for (int i = nl; i >= 2; --i) {
        hipLaunchKernelGGL(k_up, blocks((size_t)4 * cw * ch), dim3(kBlock), 0, s, L.lv[i], L.lv[i - 1], cw, ch, strength,
                           thr, first ? part_min : nullptr, first ? part_max : nullptr, nparts, minmax);
        first = false;
        cw *= 2;
        ch *= 2;
    }
    if (first) {  // levels == 1: nothing folded the partials yet
        hipLaunchKernelGGL(k_up, dim3(1), dim3(kBlock), 0, s, L.lv[1], L.lv[0], 0, 0, strength, thr, part_min, part_max,
                           nparts, minmax);
    }
    size_t quads = ((size_t)W * H + 3) / 4;
    hipLaunchKernelGGL(k_final, blocks(quads), dim3(kBlock), 0, s, L.lv[1], (float *)depth_bits, img, mask, tensor,
                       minmax, W, H, cw, ch, strength, thr);
}

// ---------------------------------------------------------------------------------
// Peer-to-peer exchange between the contexts of one node (one process per GPU,
// the peers' frame buffers mapped through hipIpc, xGMI reads).  The pixel range is cut into
// `world` slices of `chunk` pixels; a rank reduces ITS slice over all ranks' buffers (pull), then
// every rank collects the reduced slices.  Data only ever crosses a process boundary between
// kernels -- a buffer is written by one launch and read remotely by a LATER one, ordered by the
// flag barrier below, and never written while a peer may read it (the one-barrier depth exchange of
// rtr_p2p_render writes the completed depth to a buffer of its own for that reason) -- so ordinary
// (coarse-grained) device memory is enough; only the flags live in uncached memory and are accessed
// with system-scope atomics.
//
// Barrier: every rank owns flags[world]; a rank entering barrier number `seq` stores seq into
// flags[rank] of every peer and then waits until its own flags[r] >= seq for all r.  All ranks run
// the same sequence of barriers, so the counter can simply increase.  The wait is bounded
// (wall_clock64 ticks): a rank that never arrives makes the others give up, count the event in
// `status` (mapped host memory) and go on -- the host then drops back to the collectives.
__global__ void k_p2p_sync(uint32_t *__restrict__ my_flags, PeerSet peer_flags, int rank, int world, uint32_t seq,
                           uint32_t *__restrict__ status, unsigned long long timeout_ticks) {
    const int t = threadIdx.x;
    if (t >= world || t == rank) return;
    __hip_atomic_store(static_cast<uint32_t *>(peer_flags.p[t]) + rank, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    const unsigned long long t0 = wall_clock64();
    while (__hip_atomic_load(my_flags + t, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < seq) {
        if (wall_clock64() - t0 > timeout_ticks) {
            *reinterpret_cast<volatile uint32_t *>(status) = 1u;  // plain store: the word lives in host memory
            break;
        }
        __builtin_amdgcn_s_sleep(8);
    }
}

// The same barrier, and behind it every rank's occupancy bitmap copied into local memory (occ_all[r * words
// + w]): the accumulate pass of a sharded frame then knows which peers to read for each of its tiles without
// a remote round trip per workgroup.
__global__ void k_p2p_sync_gather(uint32_t *__restrict__ my_flags, PeerSet peer_flags, int rank, int world, uint32_t seq,
                                  uint32_t *__restrict__ status, unsigned long long timeout_ticks, PeerSet occ, int words,
                                  uint32_t *__restrict__ occ_all) {
    const int t = threadIdx.x;
    if (t < world && t != rank) {
        __hip_atomic_store(static_cast<uint32_t *>(peer_flags.p[t]) + rank, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        const unsigned long long t0 = wall_clock64();
        while (__hip_atomic_load(my_flags + t, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < seq) {
            if (wall_clock64() - t0 > timeout_ticks) {
                *reinterpret_cast<volatile uint32_t *>(status) = 1u;
                break;
            }
            __builtin_amdgcn_s_sleep(8);
        }
    }
    __atomic_thread_fence(__ATOMIC_ACQUIRE);  // (one wave: every lane is past its peer's flag here)
    __builtin_amdgcn_wave_barrier();
    for (int i = t; i < world * words; i += 64)
        occ_all[i] = __hip_atomic_load(static_cast<const uint32_t *>(occ.p[i / words]) + i % words, __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_SYSTEM);
}

// Tile occupancy: a rank whose binned frame has no entry in a screen tile has left that tile of
// its depth buffer at the sentinel and of its accumulators at zero (rtr_clear), so nobody needs
// to pull it.  Point slices of a spatially ordered cloud are spatially compact, hence most tiles
// are occupied on one or two ranks only and the pulls shrink accordingly (exactly: MIN with the
// sentinel and SUM with zero change nothing).  occ_r[t >> 5] bit (t & 31) = rank r has entries in
// tile t; all ones when a rank's frame did not come from the bins.
constexpr int kOccWords = 128;  // 4096 tiles

__global__ void k_p2p_occupancy(const uint32_t *__restrict__ tile_cnt, int ntiles, uint32_t *__restrict__ occ) {
    const int w = threadIdx.x;  // one 32-tile word per thread, kOccWords threads
    uint32_t bits = 0;
    for (int b = 0; b < 32; ++b) {
        const int t = 32 * w + b;
        if (tile_cnt == nullptr || (t < ntiles && tile_cnt[t] != 0u)) bits |= 1u << b;
    }
    occ[w] = bits;
}

// every rank's occupancy words into LDS (world x kOccWords dwords, one remote round trip per block)
__device__ __forceinline__ void stage_occupancy(uint32_t *s_occ, const PeerSet &occ, int world) {
    for (int i = threadIdx.x; i < world * kOccWords; i += kBlock)
        s_occ[i] = static_cast<const uint32_t *>(occ.p[i / kOccWords])[i % kOccWords];
    __syncthreads();
}
__device__ __forceinline__ bool occupied(const uint32_t *s_occ, int r, int tile) {
    tile &= 32 * kOccWords - 1;  // frames beyond 4096 tiles never come from the bins: all ones anyway
    return (s_occ[r * kOccWords + (tile >> 5)] >> (tile & 31)) & 1u;
}

// depth: red[first .. first + count) = MIN over ranks of depth_r[...] (u32 bit patterns, render.cu:81)
__global__ __launch_bounds__(kBlock) void k_p2p_depth_reduce(PeerSet depth, PeerSet occ, uint32_t *__restrict__ red,
                                                             size_t first, size_t count, int world, int W, TileGeom g) {
    __shared__ uint32_t s_occ[kMaxPeers * kOccWords];
    stage_occupancy(s_occ, occ, world);
    const size_t q = ((size_t)blockIdx.x * kBlock + threadIdx.x) * 4;
    if (q >= count) return;
    if (q + 4 <= count && (W & 3) == 0) {  // the quad lies in one row and one tile
        const size_t p = first + q;
        const int tile = (int)((p / W) >> 5) * g.tiles_x + (int)((p % W) >> g.tw_shift);
        uint4 m = make_uint4(RTR_EMPTY, RTR_EMPTY, RTR_EMPTY, RTR_EMPTY);
        for (int r = 0; r < world; ++r) {
            if (!occupied(s_occ, r, tile)) continue;
            const uint4 v = *reinterpret_cast<const uint4 *>(static_cast<const uint32_t *>(depth.p[r]) + p);
            m.x = v.x < m.x ? v.x : m.x; m.y = v.y < m.y ? v.y : m.y;
            m.z = v.z < m.z ? v.z : m.z; m.w = v.w < m.w ? v.w : m.w;
        }
        *reinterpret_cast<uint4 *>(red + p) = m;
    } else {
        for (size_t i = q; i < count && i < q + 4; ++i) {
            const size_t p = first + i;
            const int tile = (int)((p / W) >> 5) * g.tiles_x + (int)((p % W) >> g.tw_shift);
            uint32_t m = RTR_EMPTY;
            for (int r = 0; r < world; ++r) {
                if (!occupied(s_occ, r, tile)) continue;
                const uint32_t v = static_cast<const uint32_t *>(depth.p[r])[p];
                m = v < m ? v : m;
            }
            red[p] = m;
        }
    }
}

// dst[i] = src_owner(i)[i] for every 16-byte element i in [0, n16): collects the slices every
// rank reduced (depth: 4 pixels per element; image: slices are multiples of 16 pixels = 48 bytes,
// so an element never straddles two owners).  The < 16 trailing bytes of the buffer are copied by
// k_p2p_gather_tail.
__global__ __launch_bounds__(kBlock) void k_p2p_gather(PeerSet src, uint4 *__restrict__ dst, size_t chunk16, size_t n16,
                                                       int skip_owner) {
    const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n16) return;
    const int owner = (int)(i / chunk16);
    if (owner == skip_owner) return;  // this rank's own slice when it is already in place
    dst[i] = static_cast<const uint4 *>(src.p[owner])[i];
}
__global__ void k_p2p_gather_tail(PeerSet src, uint8_t *__restrict__ dst, size_t from, size_t to, int owner, int rank) {
    if (owner == rank) return;
    for (size_t i = from + threadIdx.x; i < to; i += blockDim.x) dst[i] = static_cast<const uint8_t *>(src.p[owner])[i];
}

// colour: img[first .. first + count) = resolve(SUM over ranks of acc_r[...]) (render.cu:125-128,147-162;
// u32 sums wrap like the reference's atomicAdd).  first % 4 == 0; four pixels per thread.
__global__ __launch_bounds__(kBlock) void k_p2p_acc_resolve(PeerSet acc, PeerSet occ, uint8_t *__restrict__ img,
                                                            size_t first, size_t count, int world, int W, TileGeom g) {
    __shared__ uint32_t s_occ[kMaxPeers * kOccWords];
    stage_occupancy(s_occ, occ, world);
    const size_t q = ((size_t)blockIdx.x * kBlock + threadIdx.x) * 4;
    if (q >= count) return;
    const int cnt = (count - q) < 4 ? (int)(count - q) : 4;
    uint32_t out[12];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        uint4 a = make_uint4(0, 0, 0, 0);
        if (k < cnt) {
            const size_t p = first + q + k;
            const int tile = (int)((p / W) >> 5) * g.tiles_x + (int)((p % W) >> g.tw_shift);
            for (int r = 0; r < world; ++r) {
                if (!occupied(s_occ, r, tile)) continue;
                const uint4 v = static_cast<const uint4 *>(acc.p[r])[p];
                a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
            }
        }
        const uint32_t c = a.w;
        out[3 * k + 0] = c ? a.x / c : 0u;
        out[3 * k + 1] = c ? a.y / c : 0u;
        out[3 * k + 2] = c ? a.z / c : 0u;
    }
    uint8_t *o = img + (first + q) * 3;
    if (cnt == 4) {
        uint32_t *o32 = reinterpret_cast<uint32_t *>(o);
        o32[0] = (out[0] & 0xFF) | ((out[1] & 0xFF) << 8) | ((out[2] & 0xFF) << 16) | ((out[3] & 0xFF) << 24);
        o32[1] = (out[4] & 0xFF) | ((out[5] & 0xFF) << 8) | ((out[6] & 0xFF) << 16) | ((out[7] & 0xFF) << 24);
        o32[2] = (out[8] & 0xFF) | ((out[9] & 0xFF) << 8) | ((out[10] & 0xFF) << 16) | ((out[11] & 0xFF) << 24);
    } else {
        for (int k = 0; k < 3 * cnt; ++k) o[k] = (uint8_t)out[k];
    }
}

// Owner-computes sharded frames, the frame owner's last step: one workgroup per screen tile.  A tile another rank
// produced is copied from that rank's depth buffer / image exchange copy (row pieces of 128 / 96 bytes over xGMI); a
// tile nobody has points in is cleared; either way the depth tile passes through LDS, so the prefilter's four
// min-pool levels and the tile's min / max partial are emitted here (what k_tile does for the tiles this rank
// produced itself, which this kernel leaves alone).
__global__ __launch_bounds__(kBlock) void k_p2p_collect(TileGeom g, int W, int H, const OwnedTab *__restrict__ tab,
                                                       const uint32_t *__restrict__ occ_all, int world, int rank,
                                                       uint32_t *__restrict__ depth, uint8_t *__restrict__ img, TilePyr pyr) {
    extern __shared__ uint32_t s_col[];  // [tpix] depth tile, then 4 * tpix words of pyramid scratch
    const int tile = blockIdx.x, tid = threadIdx.x;
    uint32_t mask;
    const int owner = tile_owner(occ_all, world, tile, mask);
    if (owner == rank) return;  // (workgroup-uniform) produced here, pyramid included
    const int tpix = 32 << g.tw_shift, tw = 1 << g.tw_shift;
    const int tx0 = (tile % g.tiles_x) << g.tw_shift, ty0 = (tile / g.tiles_x) * kTileH;
    const uint32_t *src_d = owner >= 0 ? tab->depth[owner] : nullptr;
    const uint8_t *src_i = owner >= 0 ? tab->ximg[owner] : nullptr;
    for (int p = tid; p < tpix; p += kBlock) {
        const int x = tx0 + (p & (tw - 1)), y = ty0 + (p >> g.tw_shift);
        uint32_t v = RTR_EMPTY;
        if (x < W && y < H) {
            const size_t gp = (size_t)y * W + x;
            if (src_d) v = src_d[gp];
            depth[gp] = v;
        }
        s_col[p] = v;
    }
    const int row_b = 3 * tw;  // image bytes per tile row
    if ((tx0 + tw <= W) && ((W & 3) == 0)) {  // whole, 4-byte aligned row pieces: dwords
        const int row_dw = row_b >> 2;
        for (int q = tid; q < row_dw * kTileH; q += kBlock) {
            const int rr = q / row_dw, dw = q - rr * row_dw, y = ty0 + rr;
            if (y < H) {
                const size_t off = ((size_t)y * W + tx0) * 3;
                reinterpret_cast<uint32_t *>(img + off)[dw] = src_i ? reinterpret_cast<const uint32_t *>(src_i + off)[dw] : 0u;
            }
        }
    } else {
        for (int q = tid; q < 3 * tpix; q += kBlock) {
            const int p = q / 3, ch = q - 3 * p;
            const int x = tx0 + (p & (tw - 1)), y = ty0 + (p >> g.tw_shift);
            if (x < W && y < H) {
                const size_t off = ((size_t)y * W + x) * 3 + ch;
                img[off] = src_i ? src_i[off] : (uint8_t)0;
            }
        }
    }
    if (pyr.enable) {
        __syncthreads();
        tile_pyramid(s_col, s_col + tpix, g, tx0, ty0, pyr.L, pyr.n_eff_rows, pyr.part_min, pyr.part_max, tile, tid, kBlock);
    }
}
void launch_p2p_collect(hipStream_t s, int W, int H, const OwnedTab *tab, const uint32_t *occ_all, int world, int rank,
                        uint32_t *depth, uint8_t *img, const TilePyr *pyr) {
    const TileGeom g = tile_geom(W, H);
    TilePyr none{};
    none.enable = 0;
    const size_t tpix = (size_t)32 << g.tw_shift;
    hipLaunchKernelGGL(k_p2p_collect, dim3(g.ntiles), dim3(kBlock), 5 * tpix * sizeof(uint32_t), s, g, W, H, tab, occ_all, world,
                       rank, depth, img, pyr ? *pyr : none);
}

void launch_p2p_sync(hipStream_t s, uint32_t *my_flags, const PeerSet &peer_flags, int rank, int world, uint32_t seq,
                     uint32_t *status, unsigned long long timeout_ticks) {
    hipLaunchKernelGGL(k_p2p_sync, dim3(1), dim3(64), 0, s, my_flags, peer_flags, rank, world, seq, status, timeout_ticks);
}
void launch_p2p_sync_gather(hipStream_t s, uint32_t *my_flags, const PeerSet &peer_flags, int rank, int world, uint32_t seq,
                            uint32_t *status, unsigned long long timeout_ticks, const PeerSet &occ, uint32_t *occ_all) {
    hipLaunchKernelGGL(k_p2p_sync_gather, dim3(1), dim3(64), 0, s, my_flags, peer_flags, rank, world, seq, status,
                       timeout_ticks, occ, 128, occ_all);
}
// tile_cnt == NULL: every tile counts as occupied (the frame did not come from the bins)
void launch_p2p_occupancy(hipStream_t s, const uint32_t *tile_cnt, int W, int H, uint32_t *occ) {
    hipLaunchKernelGGL(k_p2p_occupancy, dim3(1), dim3(kOccWords), 0, s, tile_cnt, tile_count(W, H), occ);
}
void launch_p2p_depth_reduce(hipStream_t s, const PeerSet &depth, const PeerSet &occ, uint32_t *red, size_t first,
                             size_t count, int world, int W, int H) {
    if (count == 0) return;
    const size_t quads = (count + 3) / 4;
    hipLaunchKernelGGL(k_p2p_depth_reduce, dim3((unsigned)((quads + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, depth, occ,
                       red, first, count, world, W, tile_geom(W, H));
}
// gathers `nbytes` of a buffer cut into slices of `chunk_bytes` (a multiple of 16) owned by ranks 0, 1, ...
void launch_p2p_gather(hipStream_t s, const PeerSet &src, void *dst, size_t chunk_bytes, size_t nbytes, int skip_owner) {
    const size_t n16 = nbytes / 16;
    if (n16)
        hipLaunchKernelGGL(k_p2p_gather, dim3((unsigned)((n16 + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, src, (uint4 *)dst,
                           chunk_bytes / 16, n16, skip_owner);
    if (nbytes % 16) {
        const int owner = (int)((nbytes - 1) / chunk_bytes);
        if (owner != skip_owner)
            hipLaunchKernelGGL(k_p2p_gather_tail, dim3(1), dim3(64), 0, s, src, (uint8_t *)dst, n16 * 16, nbytes, owner, -1);
    }
}
void launch_p2p_acc_resolve(hipStream_t s, const PeerSet &acc, const PeerSet &occ, uint8_t *img, size_t first,
                            size_t count, int world, int W, int H) {
    if (count == 0) return;
    const size_t quads = (count + 3) / 4;
    hipLaunchKernelGGL(k_p2p_acc_resolve, dim3((unsigned)((quads + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, acc, occ, img,
                       first, count, world, W, tile_geom(W, H));
}

// ---------------------------------------------------------------------------------
// synthetic scenes generated straight into HBM; op-for-op the same
// arithmetic as orc_generate so CPU, GPU shards and fixtures agree bit for bit.
__device__ __forceinline__ uint64_t mix64(uint64_t z) {
    z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull;
    z ^= z >> 27; z *= 0x94D049BB133111EBull;
    z ^= z >> 31; return z;
}
__device__ __forceinline__ uint64_t hsh(uint64_t seed, uint64_t i, uint64_t k) {
    return mix64(seed + (4ull * i + k) * 0x9E3779B97F4A7C15ull);
}
__device__ __forceinline__ float u01(uint64_t h) { return (float)(uint32_t)(h >> 40) * 0x1p-24f; }
__device__ __forceinline__ uint32_t compact1by1(uint64_t v) {
    v &= 0x5555555555555555ull;
    v = (v | (v >> 1)) & 0x3333333333333333ull;
    v = (v | (v >> 2)) & 0x0F0F0F0F0F0F0F0Full;
    v = (v | (v >> 4)) & 0x00FF00FF00FF00FFull;
    v = (v | (v >> 8)) & 0x0000FFFF0000FFFFull;
    v = (v | (v >> 16)) & 0x00000000FFFFFFFFull;
    return (uint32_t)v;
}

__device__ void room_shell_point(uint64_t seed, uint64_t i, uint64_t total, float &px, float &py, float &pz) {
    const uint32_t wts[14] = {24, 24, 64, 64, 24, 24, 3, 3, 3, 3, 3, 3, 3, 3};
    uint64_t start = 0, cnt = 0;
    int s = 0;
    for (s = 0; s < 14; ++s) {
        cnt = (s == 13) ? (total - start) : (total * wts[s]) / 248ull;
        if (i < start + cnt || s == 13) break;
        start += cnt;
    }
    uint64_t j = i - start;
    int b = 0;
    while (b < 15 && (1ull << (2 * (b + 1))) <= cnt) ++b;
    uint64_t M = 1ull << (2 * b);
    uint64_t cell = (cnt > 0) ? (j * M) / cnt : 0;
    float cx = (float)compact1by1(cell), cy = (float)compact1by1(cell >> 1);
    float scale = __uint_as_float((uint32_t)(127 - b) << 23);
    float sp = f_mul(f_add(cx, u01(hsh(seed, i, 0))), scale);
    float tp = f_mul(f_add(cy, u01(hsh(seed, i, 1))), scale);
    if (s < 6) {
        float a8 = f_add(-4.0f, f_mul(sp, 8.0f)), b8 = f_add(-4.0f, f_mul(tp, 8.0f));
        float a3 = f_add(-1.5f, f_mul(sp, 3.0f)), b3 = f_add(-1.5f, f_mul(tp, 3.0f));
        switch (s) {
            case 0: px = -4.0f; py = a3; pz = b8; break;
            case 1: px = 4.0f; py = a3; pz = b8; break;
            case 2: px = a8; py = -1.5f; pz = b8; break;
            case 3: px = a8; py = 1.5f; pz = b8; break;
            case 4: px = a8; py = b3; pz = -4.0f; break;
            default: px = a8; py = b3; pz = 4.0f; break;
        }
    } else {
        int q = s - 6;
        float cxs = (q & 1) ? 2.0f : -2.0f, cys = (q & 2) ? 0.75f : -0.75f, czs = (q & 4) ? 2.0f : -2.0f;
        float a = f_sub(f_mul(2.0f, sp), 1.0f), bb = f_sub(f_mul(2.0f, tp), 1.0f);
        float aa = fabsf(a), ab = fabsf(bb);
        float vz = f_sub(f_sub(1.0f, aa), ab);
        float vx = a, vy = bb;
        if (vz < 0.0f) {
            vx = copysignf(f_sub(1.0f, ab), a);
            vy = copysignf(f_sub(1.0f, aa), bb);
        }
        float len = __builtin_sqrtf(fmaf(vz, vz, fmaf(vy, vy, f_mul(vx, vx))));
        float k = 0.5f / len;
        px = fmaf(vx, k, cxs);
        py = fmaf(vy, k, cys);
        pz = fmaf(vz, k, czs);
    }
}

__global__ __launch_bounds__(kBlock) void k_generate(int scene, uint64_t seed, uint64_t first, uint64_t count,
                                                     uint64_t total, float *__restrict__ x, float *__restrict__ y,
                                                     float *__restrict__ z, uint32_t *__restrict__ rgba) {
    uint64_t stride = (uint64_t)gridDim.x * kBlock;
    for (uint64_t t = (uint64_t)blockIdx.x * kBlock + threadIdx.x; t < count; t += stride) {
        uint64_t i = first + t;
        float px, py, pz;
        if (scene == 0) {
            px = f_add(-4.0f, f_mul(u01(hsh(seed, i, 0)), 8.0f));
            py = f_add(-1.5f, f_mul(u01(hsh(seed, i, 1)), 3.0f));
            pz = f_add(-4.0f, f_mul(u01(hsh(seed, i, 2)), 8.0f));
        } else {
            room_shell_point(seed, i, total, px, py, pz);
        }
        x[t] = px;
        y[t] = py;
        z[t] = pz;
        rgba[t] = (uint32_t)(hsh(seed, i, 3) & 0xFFFFFFull) | 0xFF000000u;
    }
}

void launch_generate(hipStream_t s, int scene, uint64_t seed, uint64_t first, uint64_t count, uint64_t total, float *x,
                     float *y, float *z, uint32_t *rgba) {
    if (count == 0) return;
    uint64_t blocks = (count + kBlock - 1) / kBlock;
    int grid = (int)(blocks < 8192 ? blocks : 8192);
    hipLaunchKernelGGL(k_generate, dim3(grid), dim3(kBlock), 0, s, scene, seed, first, count, total, x, y, z, rgba);
}

// ---------------------------------------------------------------------------------
// layout conversion at the boundary: the reference hands over AoS float4 / uchar4
// (Octreegrid.h:162-180); the kernels stream SoA.
__global__ __launch_bounds__(kBlock) void k_aos_to_soa(const uint8_t *__restrict__ xyz, size_t xs,
                                                       const uint8_t *__restrict__ rgb, size_t rs, uint64_t count,
                                                       float *__restrict__ x, float *__restrict__ y,
                                                       float *__restrict__ z, uint32_t *__restrict__ rgba) {
    uint64_t stride = (uint64_t)gridDim.x * kBlock;
    for (uint64_t t = (uint64_t)blockIdx.x * kBlock + threadIdx.x; t < count; t += stride) {
        const float *p = (const float *)(xyz + t * xs);
        x[t] = p[0];
        y[t] = p[1];
        z[t] = p[2];
        const uint8_t *c = rgb + t * rs;
        rgba[t] = (uint32_t)c[0] | ((uint32_t)c[1] << 8) | ((uint32_t)c[2] << 16) | 0xFF000000u;
    }
}

void launch_aos_to_soa(hipStream_t s, const uint8_t *xyz, size_t xyz_stride, const uint8_t *rgb, size_t rgb_stride,
                       uint64_t count, float *x, float *y, float *z, uint32_t *rgba) {
    if (count == 0) return;
    uint64_t blocks = (count + kBlock - 1) / kBlock;
    int grid = (int)(blocks < 8192 ? blocks : 8192);
    hipLaunchKernelGGL(k_aos_to_soa, dim3(grid), dim3(kBlock), 0, s, xyz, xyz_stride, rgb, rgb_stride, count, x, y, z,
                       rgba);
}

__global__ __launch_bounds__(kBlock) void k_soa_to_aos(const float *__restrict__ x, const float *__restrict__ y,
                                                       const float *__restrict__ z, const uint32_t *__restrict__ rgba,
                                                       uint64_t count, float4 *__restrict__ xyzw,
                                                       uint32_t *__restrict__ out) {
    uint64_t stride = (uint64_t)gridDim.x * kBlock;
    for (uint64_t t = (uint64_t)blockIdx.x * kBlock + threadIdx.x; t < count; t += stride) {
        xyzw[t] = make_float4(x[t], y[t], z[t], 1.0f);
        out[t] = rgba[t];
    }
}

void launch_soa_to_aos(hipStream_t s, const float *x, const float *y, const float *z, const uint32_t *rgba,
                       uint64_t count, float *xyzw, uint8_t *rgba_out) {
    if (count == 0) return;
    uint64_t blocks = (count + kBlock - 1) / kBlock;
    int grid = (int)(blocks < 8192 ? blocks : 8192);
    hipLaunchKernelGGL(k_soa_to_aos, dim3(grid), dim3(kBlock), 0, s, x, y, z, rgba, count, (float4 *)xyzw,
                       (uint32_t *)rgba_out);
}

__global__ void k_pad_nan(float *x, float *y, float *z, uint32_t *rgba, uint64_t n, uint64_t n_pad) {
    uint64_t t = n + threadIdx.x;
    if (t < n_pad) {
        float q = __uint_as_float(0x7FC00000u);
        x[t] = q;
        y[t] = q;
        z[t] = q;
        rgba[t] = 0;
    }
}

// Device -> pinned host copy of a frame's outputs by a small grid of plain 16-byte stores over PCIe
// (rtr_project_async).  It runs on the copy stream beside the next frame's kernels whatever the runtime's copy path
// would have chosen: hipMemcpyAsync into the same buffers overlapped in some processes and serialised in others
// (0.34 .. 0.62 ms per frame for one and the same loop).  The grid is deliberately tiny: measured on C3 + filter,
// frames per second through the async pair peak at 8 workgroups (0.328 ms/frame; 4: 0.40, 16: 0.37, 64: 0.43,
// 512: 0.48) -- fewer cannot keep the link busy, more take dispatch slots from the next frame's T1.
constexpr int kCopyWgs = 8;
typedef uint32_t u32x4_n __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(kBlock) void k_copy_to_host(const u32x4_n *__restrict__ a, u32x4_n *__restrict__ ha, size_t na,
                                                        const u32x4_n *__restrict__ b, u32x4_n *__restrict__ hb, size_t nb) {
    const size_t stride = (size_t)gridDim.x * kBlock;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < na; i += stride) __builtin_nontemporal_store(a[i], ha + i);
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < nb; i += stride) __builtin_nontemporal_store(b[i], hb + i);
}
void launch_copy_to_host(hipStream_t s, const void *a, void *ha, size_t bytes_a, const void *b, void *hb, size_t bytes_b) {
    hipLaunchKernelGGL(k_copy_to_host, dim3(kCopyWgs), dim3(kBlock), 0, s, (const u32x4_n *)a, (u32x4_n *)ha, (bytes_a + 15) / 16,
                       (const u32x4_n *)b, (u32x4_n *)hb, (bytes_b + 15) / 16);
}

void launch_pad_nan(hipStream_t s, float *x, float *y, float *z, uint32_t *rgba, uint64_t n, uint64_t n_pad) {
    if (n_pad > n) hipLaunchKernelGGL(k_pad_nan, dim3(1), dim3(64), 0, s, x, y, z, rgba, n, n_pad);
}

}  // namespace rtr
