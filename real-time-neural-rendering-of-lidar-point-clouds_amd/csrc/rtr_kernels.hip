// rtr_kernels.hip -- hand-written gfx950 (MI355X, wave64) kernels of the projector.
//
// Arithmetic contract (same as oracle/rtr_oracle.c): every fp32 op is
// one IEEE RNE operation, fused only where fmaf() is written.  This file MUST be
// compiled with -ffp-contract=off and hipcc's default correctly rounded fp32
// divide / sqrt.  Nothing here is GEMM-shaped: the path is an HBM-bound stream plus
// a scatter, so the work goes into coalescing, atomic traffic and launch count.
#include "rtr_kernels.h"

#include <hip/hip_fp16.h>

namespace rtr {

#define RTR_EMPTY 0x7F7FFFFFu
constexpr int kBlock = 256;   // 4 waves
constexpr int kPtGrid = 2048; // 256 CUs x 8 resident blocks, grid-stride the rest

__device__ __forceinline__ float f_mul(float a, float b) { return __fmul_rn(a, b); }
__device__ __forceinline__ float f_add(float a, float b) { return __fadd_rn(a, b); }
__device__ __forceinline__ float f_sub(float a, float b) { return __fsub_rn(a, b); }

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
    if (grid > kPtGrid) grid = kPtGrid;
    hipLaunchKernelGGL(k_clear, dim3(grid), dim3(kBlock), 0, s, (uint4 *)depth, (uint4 *)acc, n_depth4, n_acc4, depth,
                       npix);
}

// ---------------------------------------------------------------------------------
// A4 minDepthPass (render.cu:53-83).  Semantics = "atomicMin of every surviving
// point"; the reference's __match_any_sync aggregation is only a contention trick.
// Here: SoA float4 loads (1 KiB per wave-instruction per coordinate), early-z (skip
// the atomic unless strictly closer than what an L1-bypassing load sees).  A stale
// early-z value can only cause a redundant atomic, never a wrong result.
__device__ __forceinline__ void zmin(uint32_t *depth, int pix, float d) {
    if (pix < 0) return;
    uint32_t b = __float_as_uint(d);
    uint32_t cur = __hip_atomic_load(&depth[pix], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (b < cur) atomicMin(&depth[pix], b);
}

__global__ __launch_bounds__(kBlock) void k_min_depth(const float4 *__restrict__ x4, const float4 *__restrict__ y4,
                                                      const float4 *__restrict__ z4, uint64_t n4, Proj P, int W, int H,
                                                      uint32_t *__restrict__ depth) {
    const float fW = (float)W, fH = (float)H;
    uint64_t stride = (uint64_t)gridDim.x * kBlock;
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n4; i += stride) {
        float4 X = x4[i], Y = y4[i], Z = z4[i];
        float d0, d1, d2, d3;
        int p0 = project_point(P, X.x, Y.x, Z.x, W, H, fW, fH, d0);
        int p1 = project_point(P, X.y, Y.y, Z.y, W, H, fW, fH, d1);
        int p2 = project_point(P, X.z, Y.z, Z.z, W, H, fW, fH, d2);
        int p3 = project_point(P, X.w, Y.w, Z.w, W, H, fW, fH, d3);
        zmin(depth, p0, d0);
        zmin(depth, p1, d1);
        zmin(depth, p2, d2);
        zmin(depth, p3, d3);
    }
}

void launch_min_depth(hipStream_t s, const Cloud &c, const Proj &P, int W, int H, uint32_t *depth) {
    uint64_t n4 = (c.n + 3) / 4;
    if (n4 == 0) return;
    uint64_t blocks = (n4 + kBlock - 1) / kBlock;
    int grid = (int)(blocks < (uint64_t)kPtGrid ? blocks : (uint64_t)kPtGrid);
    hipLaunchKernelGGL(k_min_depth, dim3(grid), dim3(kBlock), 0, s, (const float4 *)c.x, (const float4 *)c.y,
                       (const float4 *)c.z, n4, P, W, H, depth);
}

// ---------------------------------------------------------------------------------
// A5 accumulatePass (render.cu:85-130): re-project, depth-window test against the
// (global) minimum, integer colour sums.
__device__ __forceinline__ void zacc(const uint32_t *__restrict__ depth, uint32_t *__restrict__ acc,
                                     const uint32_t *__restrict__ rgba, uint64_t idx, int pix, float d, float window) {
    if (pix < 0) return;
    float m = __uint_as_float(depth[pix]);
    if (d > f_add(m, window)) return;  // render.cu:106
    uint32_t c = rgba[idx];
    uint32_t *a = acc + 4 * (size_t)pix;
    atomicAdd(a + 0, c & 0xFFu);
    atomicAdd(a + 1, (c >> 8) & 0xFFu);
    atomicAdd(a + 2, (c >> 16) & 0xFFu);
    atomicAdd(a + 3, 1u);
}

__global__ __launch_bounds__(kBlock) void k_accumulate(const float4 *__restrict__ x4, const float4 *__restrict__ y4,
                                                       const float4 *__restrict__ z4,
                                                       const uint32_t *__restrict__ rgba, uint64_t n4, Proj P, int W,
                                                       int H, const uint32_t *__restrict__ depth,
                                                       uint32_t *__restrict__ acc, float window) {
    const float fW = (float)W, fH = (float)H;
    uint64_t stride = (uint64_t)gridDim.x * kBlock;
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n4; i += stride) {
        float4 X = x4[i], Y = y4[i], Z = z4[i];
        float d0, d1, d2, d3;
        int p0 = project_point(P, X.x, Y.x, Z.x, W, H, fW, fH, d0);
        int p1 = project_point(P, X.y, Y.y, Z.y, W, H, fW, fH, d1);
        int p2 = project_point(P, X.z, Y.z, Z.z, W, H, fW, fH, d2);
        int p3 = project_point(P, X.w, Y.w, Z.w, W, H, fW, fH, d3);
        zacc(depth, acc, rgba, 4 * i + 0, p0, d0, window);
        zacc(depth, acc, rgba, 4 * i + 1, p1, d1, window);
        zacc(depth, acc, rgba, 4 * i + 2, p2, d2, window);
        zacc(depth, acc, rgba, 4 * i + 3, p3, d3, window);
    }
}

void launch_accumulate(hipStream_t s, const Cloud &c, const Proj &P, int W, int H, const uint32_t *depth,
                       uint32_t *acc, float window) {
    uint64_t n4 = (c.n + 3) / 4;
    if (n4 == 0) return;
    uint64_t blocks = (n4 + kBlock - 1) / kBlock;
    int grid = (int)(blocks < (uint64_t)kPtGrid ? blocks : (uint64_t)kPtGrid);
    hipLaunchKernelGGL(k_accumulate, dim3(grid), dim3(kBlock), 0, s, (const float4 *)c.x, (const float4 *)c.y,
                       (const float4 *)c.z, c.rgba, n4, P, W, H, depth, acc, window);
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
// depth-heuristic prefilter (project_cloud.cu:28-187)

// A8 reduce (project_cloud.cu:28-53): 2x2 min-pool; source row stride 2*w.
__global__ __launch_bounds__(kBlock) void k_reduce(const float *__restrict__ hi, float *__restrict__ lo, int w, int h) {
    int idx = blockIdx.x * kBlock + threadIdx.x;
    if (idx >= w * h) return;
    int x = idx % w, y = idx / w;
    const float2 *r0 = (const float2 *)(hi + (size_t)(2 * y) * (2 * w)) + x;
    const float2 *r1 = (const float2 *)(hi + (size_t)(2 * y + 1) * (2 * w)) + x;
    float2 a = *r0, b = *r1;
    float l0 = a.x < a.y ? a.x : a.y;
    float l1 = b.x < b.y ? b.x : b.y;
    lo[idx] = l0 < l1 ? l0 : l1;
}

// A9 laplacianKernel (project_cloud.cu:55-79): all nine taps, row-major, fmaf chain.
__global__ __launch_bounds__(kBlock) void k_laplacian(const float *__restrict__ in, uint8_t *__restrict__ out, int w,
                                                      int h, float thr) {
    int idx = blockIdx.x * kBlock + threadIdx.x;
    if (idx >= w * h) return;
    int x = idx % w, y = idx / w;
    if (x == 0 || x == w - 1 || y == 0 || y == h - 1) {
        out[idx] = 0;
        return;
    }
    const float k[9] = {0.f, 1.f, 0.f, 1.f, -4.f, 1.f, 0.f, 1.f, 0.f};  // project_cloud.cu:26
    float sum = 0.0f;
#pragma unroll
    for (int ky = -1; ky <= 1; ++ky)
#pragma unroll
        for (int kx = -1; kx <= 1; ++kx) sum = fmaf(in[(y + ky) * w + (x + kx)], k[(ky + 1) * 3 + (kx + 1)], sum);
    out[idx] = (sum > thr) ? 255 : 0;
}

__device__ __forceinline__ float lo_px(const float *__restrict__ lo, int x, int y, int w, int h) {
    return (x >= 0 && x < w && y >= 0 && y < h) ? lo[y * w + x] : -1.0f;  // project_cloud.cu:81-86
}

// A10 compareImgsKernel (project_cloud.cu:88-126)
__global__ __launch_bounds__(kBlock) void k_compare(const float *__restrict__ lo, const float *__restrict__ hi,
                                                    const uint8_t *__restrict__ grad, uint8_t *__restrict__ mask, int hw,
                                                    int hh, float strength) {
    int idx = blockIdx.x * kBlock + threadIdx.x;
    if (idx >= hw * hh) return;
    int x = idx % hw, y = idx / hw;
    float cur = hi[idx];
    if ((double)cur >= 3.4028e38) {  // MAX_FLOAT, project_cloud.cu:21,97
        mask[idx] = 0;
        return;
    }
    int lw = hw / 2, lh = hh / 2, lx = x / 2, ly = y / 2;
    bool keep = false;
    if (grad[ly * lw + lx] > 0) {
#pragma unroll
        for (int dx = -1; dx <= 1; ++dx)
#pragma unroll
            for (int dy = -1; dy <= 1; ++dy) keep = keep || (cur <= f_mul(lo_px(lo, lx + dx, ly + dy, lw, lh), strength));
    } else {
        keep = cur <= f_mul(lo_px(lo, lx, ly, lw, lh), strength);
    }
    mask[idx] = keep ? 255 : 0;
}

// A11 resizeKernel (project_cloud.cu:128-161): in-place bilinear x2 where mask == 0.
__global__ __launch_bounds__(kBlock) void k_resize(const float *__restrict__ lo, float *__restrict__ hi,
                                                   const uint8_t *__restrict__ mask, int ow, int oh) {
    int idx = blockIdx.x * kBlock + threadIdx.x;
    if (idx >= ow * oh) return;
    if (mask[idx] > 0) return;
    int x = idx % ow, y = idx / ow;
    int lw = ow / 2, lh = oh / 2;
    float inX = f_sub(f_add((float)x, 0.5f) / 2.0f, 0.5f);
    float inY = f_sub(f_add((float)y, 0.5f) / 2.0f, 0.5f);
    int x0 = (int)floorf(inX), x1 = x0 + 1, y0 = (int)floorf(inY), y1 = y0 + 1;
    x0 = x0 < 0 ? 0 : (x0 >= lw ? lw - 1 : x0);
    x1 = x1 < 0 ? 0 : (x1 >= lw ? lw - 1 : x1);
    y0 = y0 < 0 ? 0 : (y0 >= lh ? lh - 1 : y0);
    y1 = y1 < 0 ? 0 : (y1 >= lh ? lh - 1 : y1);
    float wx = f_sub(inX, (float)x0), wy = f_sub(inY, (float)y0);
    float v0 = fmaf(wx, lo[y0 * lw + x1], f_mul(f_sub(1.0f, wx), lo[y0 * lw + x0]));
    float v1 = fmaf(wx, lo[y1 * lw + x1], f_mul(f_sub(1.0f, wx), lo[y1 * lw + x0]));
    hi[idx] = fmaf(wy, v1, f_mul(f_sub(1.0f, wy), v0));
}

// A12 (render.cu:168-240): min / max of the depth bit patterns, sentinel skipped.
// wave64 shuffle reduction, one atomic pair per wave; minmax[] pre-set to {~0, 0}.
__global__ __launch_bounds__(kBlock) void k_minmax(const uint32_t *__restrict__ d, size_t n,
                                                   uint32_t *__restrict__ minmax) {
    uint32_t lo = 0xFFFFFFFFu, hi = 0u;
    size_t stride = (size_t)gridDim.x * kBlock;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        uint32_t v = d[i];
        if (v != RTR_EMPTY) {
            lo = v < lo ? v : lo;
            hi = v > hi ? v : hi;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        uint32_t ol = __shfl_xor(lo, off, 64), oh = __shfl_xor(hi, off, 64);
        lo = ol < lo ? ol : lo;
        hi = oh > hi ? oh : hi;
    }
    if ((threadIdx.x & 63) == 0) {
        if (lo != 0xFFFFFFFFu) atomicMin(&minmax[0], lo);
        if (hi != 0u) atomicMax(&minmax[1], hi);
    }
}

__global__ void k_minmax_init(uint32_t *minmax) {
    minmax[0] = 0xFFFFFFFFu;
    minmax[1] = 0u;
}

__device__ __forceinline__ uint16_t to_half_bits(float f) {
    if (f != f) return 0x7E00u;  // canonical NaN (oracle does the same)
    return __half_as_ushort(__float2half_rn(f));
}
__device__ __forceinline__ float half_round(float f) { return __half2float(__float2half_rn(f)); }

// A13 removeMask (project_cloud.cu:163-187) over ALL W*H pixels, tensor plane stride
// W*H (the reference strides by W*H_eff: quirk Q3).  Rows >= H_eff never saw the
// pyramid test: their mask is "non-empty".
__global__ __launch_bounds__(kBlock) void k_remove_mask(float *__restrict__ depth, uint8_t *__restrict__ img,
                                                        uint8_t *__restrict__ mask, uint16_t *__restrict__ tensor,
                                                        const uint32_t *__restrict__ minmax, size_t npix,
                                                        size_t n_eff) {
    size_t idx = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (idx >= npix) return;
    float d = depth[idx];
    uint8_t m;
    if (idx < n_eff) {
        m = mask[idx];
    } else {
        m = ((double)d >= 3.4028e38) ? 0 : 255;
        mask[idx] = m;
    }
    if (m == 0) {
        depth[idx] = -1.0f;
        img[3 * idx + 0] = 0;
        img[3 * idx + 1] = 0;
        img[3 * idx + 2] = 0;
        tensor[0 * npix + idx] = 0;
        tensor[1 * npix + idx] = 0;
        tensor[2 * npix + idx] = 0;
        tensor[3 * npix + idx] = 0;
        tensor[4 * npix + idx] = 0xBC00u;
        return;
    }
    float mn = __uint_as_float(minmax[0]), mx = __uint_as_float(minmax[1]);
    float range = f_sub(mx, mn);
#pragma unroll
    for (int k = 0; k < 3; ++k) tensor[k * npix + idx] = to_half_bits(half_round((float)img[3 * idx + k]) / 255.0f);
    tensor[3 * npix + idx] = to_half_bits(half_round((float)m) / 255.0f);
    tensor[4 * npix + idx] = to_half_bits(half_round(f_sub(d, mn)) / range);
}

// A14 applyDepthFilter (project_cloud.cu:331-392): same launch sequence, but on
// pre-allocated levels and without any host synchronisation or malloc per frame.
void launch_filter(hipStream_t s, const FilterLevels &L, uint32_t *depth_bits, uint8_t *img, uint8_t *mask,
                   uint8_t *grad, uint16_t *tensor, uint32_t *minmax, int W, int H, float strength, float thr) {
    auto blocks = [](size_t n) { return dim3((unsigned)((n + kBlock - 1) / kBlock)); };
    const int nl = L.levels;
    for (int i = 1; i <= nl; ++i)
        hipLaunchKernelGGL(k_reduce, blocks((size_t)L.w[i] * L.h[i]), dim3(kBlock), 0, s, L.lv[i - 1], L.lv[i], L.w[i],
                           L.h[i]);
    int cw = L.w[nl], ch = L.h[nl];
    size_t npix = (size_t)W * H;
    for (int i = nl; i >= 1; --i) {
        hipLaunchKernelGGL(k_laplacian, blocks((size_t)cw * ch), dim3(kBlock), 0, s, L.lv[i], grad, cw, ch, thr);
        cw *= 2;
        ch *= 2;
        hipLaunchKernelGGL(k_compare, blocks((size_t)cw * ch), dim3(kBlock), 0, s, L.lv[i], L.lv[i - 1], grad, mask, cw,
                           ch, strength);
        if (i == 1) {
            size_t n_eff = (size_t)cw * ch;
            hipLaunchKernelGGL(k_minmax_init, dim3(1), dim3(1), 0, s, minmax);
            int g = (int)((n_eff + kBlock - 1) / kBlock);
            if (g > 1024) g = 1024;
            hipLaunchKernelGGL(k_minmax, dim3(g), dim3(kBlock), 0, s, depth_bits, n_eff, minmax);
            hipLaunchKernelGGL(k_remove_mask, blocks(npix), dim3(kBlock), 0, s, (float *)depth_bits, img, mask, tensor,
                               minmax, npix, n_eff);
        } else {
            hipLaunchKernelGGL(k_resize, blocks((size_t)cw * ch), dim3(kBlock), 0, s, L.lv[i], L.lv[i - 1], mask, cw, ch);
        }
    }
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
        float len = __fsqrt_rn(fmaf(vz, vz, fmaf(vy, vy, f_mul(vx, vx))));
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

void launch_pad_nan(hipStream_t s, float *x, float *y, float *z, uint32_t *rgba, uint64_t n, uint64_t n_pad) {
    if (n_pad > n) hipLaunchKernelGGL(k_pad_nan, dim3(1), dim3(64), 0, s, x, y, z, rgba, n, n_pad);
}

}  // namespace rtr
