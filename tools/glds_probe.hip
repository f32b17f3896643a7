#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
// where do the bytes of `global_load_lds_dwordx4 voff, s[base] offset:K` land in LDS?  (M0 base + K + lane * 16?)
__global__ void k_probe(const uint8_t *src, uint32_t *out, int variant) {
    __shared__ __attribute__((aligned(16))) uint32_t lds[4096];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = 0xDEADBEEFu;
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63;
    uint32_t voff = lane * 16u;
    const uint32_t base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t *)lds;
    if (variant == 0) {
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %2\n\t" ::"v"(voff), "s"(base + 256u), "s"(src) : "memory");
    } else if (variant == 1) {
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %2 offset:1024\n\t" ::"v"(voff), "s"(base + 256u), "s"(src) : "memory");
    } else if (variant == 2) {
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %2 offset:2048 nt\n\t" ::"v"(voff), "s"(base), "s"(src) : "memory");
    } else {  // clamped source offsets: lanes >= 8 all read byte 112..127
        voff = voff < 112u ? voff : 112u;
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %2\n\t" ::"v"(voff), "s"(base), "s"(src) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) out[i] = lds[i];
}
int main() {
    std::vector<uint32_t> h(4096);
    for (int i = 0; i < 4096; ++i) h[i] = i;  // dword i holds i: byte offset = 4 i
    uint8_t *src; uint32_t *out;
    hipMalloc(&src, 16384); hipMalloc(&out, 16384);
    hipMemcpy(src, h.data(), 16384, hipMemcpyHostToDevice);
    for (int v = 0; v < 4; ++v) {
        k_probe<<<1, 64>>>(src, out, v);
        std::vector<uint32_t> r(4096);
        hipMemcpy(r.data(), out, 16384, hipMemcpyDeviceToHost);
        int first = -1, last = -1;
        for (int i = 0; i < 4096; ++i) if (r[i] != 0xDEADBEEFu) { if (first < 0) first = i; last = i; }
        printf("variant %d: LDS dwords [%d..%d] written; lds[first]=%u lds[first+1]=%u lds[first+4]=%u lds[last]=%u\n", v, first, last,
               first >= 0 ? r[first] : 0, first >= 0 ? r[first + 1] : 0, first >= 0 ? r[first + 4] : 0, last >= 0 ? r[last] : 0);
        if (v == 3) printf("   lane 7 -> %u, lane 8 -> %u, lane 63 -> %u (expect 28, 28, 28)\n", r[28], r[32], r[252]);
    }
    return 0;
}
