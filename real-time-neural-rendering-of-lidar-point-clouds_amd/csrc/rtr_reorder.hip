// rtr_reorder.hip -- one-off spatial reordering of the resident cloud (upload-time work, not
// part of a frame): Morton-sort the points so that consecutive indices are spatial
// neighbours.  This is what the reference's loader does at 0.25 m granularity with its block
// grid (cloudreader.cpp:8-82); the projector's output does not depend on point order
// (render.cu:81,125-128 commute), only its speed does: coherent order makes the tile sort
// append one run per wave and tile and lets whole 256-point chunks be frustum-culled.
// The radix sort is rocPRIM's (called directly); everything in the per-frame path is hand-written.
#include <hip/hip_runtime.h>

#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>

#include <cstdint>

namespace rtr {

namespace {

__device__ __forceinline__ uint32_t f2ord(float f) {  // order-preserving float -> uint
    uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__host__ __device__ inline float ord2f(uint32_t o) {
    uint32_t u = (o & 0x80000000u) ? (o & 0x7FFFFFFFu) : ~o;
    float f;
#if defined(__HIP_DEVICE_COMPILE__)
    f = __uint_as_float(u);
#else
    memcpy(&f, &u, 4);
#endif
    return f;
}

__global__ __launch_bounds__(256) void k_bbox(const float *__restrict__ x, const float *__restrict__ y,
                                              const float *__restrict__ z, uint64_t n, uint32_t *__restrict__ bb) {
    uint32_t lo[3] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}, hi[3] = {0u, 0u, 0u};
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) {
        const float v[3] = {x[i], y[i], z[i]};
#pragma unroll
        for (int k = 0; k < 3; ++k)
            if (v[k] == v[k] && fabsf(v[k]) < 3.0e38f) {  // finite only
                uint32_t o = f2ord(v[k]);
                lo[k] = o < lo[k] ? o : lo[k];
                hi[k] = o > hi[k] ? o : hi[k];
            }
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            uint32_t a = __shfl_xor(lo[k], off, 64), b = __shfl_xor(hi[k], off, 64);
            lo[k] = a < lo[k] ? a : lo[k];
            hi[k] = b > hi[k] ? b : hi[k];
        }
        if ((threadIdx.x & 63) == 0) {
            atomicMin(&bb[k], lo[k]);
            atomicMax(&bb[3 + k], hi[k]);
        }
    }
}

__device__ __forceinline__ uint64_t spread21(uint32_t v) {  // bits of v to every third position
    uint64_t x = v & 0x1FFFFFu;
    x = (x | (x << 32)) & 0x1F00000000FFFFull;
    x = (x | (x << 16)) & 0x1F0000FF0000FFull;
    x = (x | (x << 8)) & 0x100F00F00F00F00Full;
    x = (x | (x << 4)) & 0x10C30C30C30C30C3ull;
    x = (x | (x << 2)) & 0x1249249249249249ull;
    return x;
}

__global__ __launch_bounds__(256) void k_keys(const float *__restrict__ x, const float *__restrict__ y,
                                              const float *__restrict__ z, uint64_t n, float lx, float ly, float lz,
                                              float sx, float sy, float sz, uint64_t *__restrict__ keys,
                                              uint32_t *__restrict__ vals) {
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) {
        float fx = (x[i] - lx) * sx, fy = (y[i] - ly) * sy, fz = (z[i] - lz) * sz;
        uint32_t qx = (fx >= 0.0f) ? (fx < 2097151.0f ? (uint32_t)fx : 2097151u) : 0u;  // NaN -> 0
        uint32_t qy = (fy >= 0.0f) ? (fy < 2097151.0f ? (uint32_t)fy : 2097151u) : 0u;
        uint32_t qz = (fz >= 0.0f) ? (fz < 2097151.0f ? (uint32_t)fz : 2097151u) : 0u;
        keys[i] = spread21(qx) | (spread21(qy) << 1) | (spread21(qz) << 2);
        vals[i] = (uint32_t)i;
    }
}

__global__ __launch_bounds__(256) void k_gather(const uint32_t *__restrict__ perm, uint64_t n,
                                                const float *__restrict__ x, const float *__restrict__ y,
                                                const float *__restrict__ z, const uint32_t *__restrict__ c,
                                                float *__restrict__ ox, float *__restrict__ oy, float *__restrict__ oz,
                                                uint32_t *__restrict__ oc) {
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) {
        uint32_t j = perm[i];
        ox[i] = x[j];
        oy[i] = y[j];
        oz[i] = z[j];
        oc[i] = c[j];
    }
}

// How spatially compact the 256-point chunks are: sum of the chunk boxes' diagonals, number of
// chunks with a finite box, and the cloud's bounding box -- all from the chunk bounds
// (k_chunk_bounds), i.e. 24 bytes per 256 points.
__global__ __launch_bounds__(256) void k_order_quality(const float *__restrict__ bounds, uint64_t nchunks,
                                                       float *__restrict__ sum_diag, uint32_t *__restrict__ finite,
                                                       uint32_t *__restrict__ bb) {
    float sum = 0.f;
    uint32_t cnt = 0;
    uint32_t lo[3] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}, hi[3] = {0u, 0u, 0u};
    for (uint64_t c = (uint64_t)blockIdx.x * 256 + threadIdx.x; c < nchunks; c += (uint64_t)gridDim.x * 256) {
        const float *b = bounds + 6 * c;
        const float dx = b[3] - b[0], dy = b[4] - b[1], dz = b[5] - b[2];
        const float d = sqrtf(dx * dx + dy * dy + dz * dz);
        if (d == d && d < 3.0e38f) {  // finite (an empty or non-finite chunk has an infinite / NaN box)
            sum += d;
            cnt += 1;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const uint32_t a = f2ord(b[k]), e = f2ord(b[3 + k]);
                lo[k] = a < lo[k] ? a : lo[k];
                hi[k] = e > hi[k] ? e : hi[k];
            }
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        sum += __shfl_xor(sum, off, 64);
        cnt += __shfl_xor(cnt, off, 64);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const uint32_t a = __shfl_xor(lo[k], off, 64), e = __shfl_xor(hi[k], off, 64);
            lo[k] = a < lo[k] ? a : lo[k];
            hi[k] = e > hi[k] ? e : hi[k];
        }
    }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(sum_diag, sum);
        atomicAdd(finite, cnt);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            atomicMin(&bb[k], lo[k]);
            atomicMax(&bb[3 + k], hi[k]);
        }
    }
}

}  // namespace

// mean chunk diagonal / cloud diagonal (0 when undefined); returns a hipError_t as int
int order_quality(hipStream_t s, const float *bounds, uint64_t n, float *ratio, float absmax[3]) {
    *ratio = 0.f;
    absmax[0] = absmax[1] = absmax[2] = __builtin_inff();
    const uint64_t nchunks = ((n + 3) / 4 + 63) / 64;
    if (nchunks == 0) return 0;
    struct Out { float sum; uint32_t finite; uint32_t bb[6]; } h, *d = nullptr;
    const Out init{0.f, 0u, {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0u, 0u, 0u}};
    hipError_t e = hipMalloc((void **)&d, sizeof(Out));
    if (e != hipSuccess) return (int)e;
    e = hipMemcpyAsync(d, &init, sizeof init, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) {
        const unsigned grid = (unsigned)((nchunks + 255) / 256 < 128 ? (nchunks + 255) / 256 : 128);  // (few waves: eight same-address atomics each)
        hipLaunchKernelGGL(k_order_quality, dim3(grid), dim3(256), 0, s, bounds, nchunks, &d->sum, &d->finite, d->bb);
        e = hipMemcpyAsync(&h, d, sizeof h, hipMemcpyDeviceToHost, s);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    (void)hipFree(d);
    if (e != hipSuccess) return (int)e;
    if (h.finite == 0 || !(h.bb[0] <= h.bb[3])) return 0;
    float ext2 = 0.f;
    for (int k = 0; k < 3; ++k) {
        const float lo = ord2f(h.bb[k]), hi = ord2f(h.bb[3 + k]);
        const float ext = hi - lo;
        ext2 += ext * ext;
        absmax[k] = fabsf(lo) > fabsf(hi) ? fabsf(lo) : fabsf(hi);
    }
    const float cloud = sqrtf(ext2);
    if (cloud > 0.f) *ratio = (h.sum / (float)h.finite) / cloud;
    return 0;
}

// Sorts the n points in place (through scratch copies).  Returns a hipError_t as int.
int reorder_morton(hipStream_t s, float *x, float *y, float *z, uint32_t *rgba, uint64_t n) {
    if (n < 2) return 0;
    if (n >= (1ull << 32)) return (int)hipErrorInvalidValue;  // the permutation is 32-bit
    uint32_t *bb = nullptr;
    uint64_t *k0 = nullptr, *k1 = nullptr;
    uint32_t *v0 = nullptr, *v1 = nullptr;
    float *tx = nullptr, *ty = nullptr, *tz = nullptr;
    uint32_t *tc = nullptr;
    void *tmp = nullptr;
    size_t tmp_bytes = 0;
    hipError_t e = hipSuccess;
    auto ok = [&](hipError_t r) { if (e == hipSuccess && r != hipSuccess) e = r; return e == hipSuccess; };
    const uint32_t init[6] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0u, 0u, 0u};
    uint32_t hb[6];
    const int grid = 2048;
    if (ok(hipMalloc((void **)&bb, 24)) && ok(hipMemcpyAsync(bb, init, 24, hipMemcpyHostToDevice, s))) {
        hipLaunchKernelGGL(k_bbox, dim3(grid), dim3(256), 0, s, x, y, z, n, bb);
        ok(hipMemcpyAsync(hb, bb, 24, hipMemcpyDeviceToHost, s));
        ok(hipStreamSynchronize(s));
    }
    if (e == hipSuccess) {
        float lo[3], sc[3];
        for (int k = 0; k < 3; ++k) {
            float a = ord2f(hb[k]), b = ord2f(hb[3 + k]);
            if (!(hb[k] <= hb[3 + k])) { a = 0.f; b = 1.f; }  // no finite point at all
            float ext = b - a;
            lo[k] = a;
            sc[k] = ext > 0.f ? 2097151.0f / ext : 0.f;
        }
        if (ok(hipMalloc((void **)&k0, n * 8)) && ok(hipMalloc((void **)&k1, n * 8)) && ok(hipMalloc((void **)&v0, n * 4)) &&
            ok(hipMalloc((void **)&v1, n * 4))) {
            hipLaunchKernelGGL(k_keys, dim3(grid), dim3(256), 0, s, x, y, z, n, lo[0], lo[1], lo[2], sc[0], sc[1], sc[2],
                               k0, v0);
            ok(rocprim::radix_sort_pairs(nullptr, tmp_bytes, k0, k1, v0, v1, (size_t)n, 0u, 63u, s));
            if (ok(hipMalloc(&tmp, tmp_bytes ? tmp_bytes : 1)))
                ok(rocprim::radix_sort_pairs(tmp, tmp_bytes, k0, k1, v0, v1, (size_t)n, 0u, 63u, s));
        }
    }
    if (e == hipSuccess && ok(hipMalloc((void **)&tx, n * 4)) && ok(hipMalloc((void **)&ty, n * 4)) &&
        ok(hipMalloc((void **)&tz, n * 4)) && ok(hipMalloc((void **)&tc, n * 4))) {
        hipLaunchKernelGGL(k_gather, dim3(grid), dim3(256), 0, s, v1, n, x, y, z, rgba, tx, ty, tz, tc);
        ok(hipMemcpyAsync(x, tx, n * 4, hipMemcpyDeviceToDevice, s));
        ok(hipMemcpyAsync(y, ty, n * 4, hipMemcpyDeviceToDevice, s));
        ok(hipMemcpyAsync(z, tz, n * 4, hipMemcpyDeviceToDevice, s));
        ok(hipMemcpyAsync(rgba, tc, n * 4, hipMemcpyDeviceToDevice, s));
        ok(hipStreamSynchronize(s));
    }
    (void)hipFree(bb); (void)hipFree(k0); (void)hipFree(k1); (void)hipFree(v0); (void)hipFree(v1);
    (void)hipFree(tx); (void)hipFree(ty); (void)hipFree(tz); (void)hipFree(tc); (void)hipFree(tmp);
    if (e == hipSuccess) e = hipGetLastError();
    return (int)e;
}

}  // namespace rtr
