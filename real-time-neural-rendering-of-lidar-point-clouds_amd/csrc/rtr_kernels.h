// rtr_kernels.h -- launch wrappers of the gfx950 kernels (internal to librtr_hip.so).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rtr {

// Rows 0..2 of the row-major camera matrix (row 3 is never used: render.cu:33-40
// computes result.w but nothing reads it).  Passed BY VALUE so the twelve floats
// live in SGPRs -- the reference pays a synchronous 64-byte H2D copy per frame
// (project_cloud.cu:320).
struct Proj {
    float m[12];
};

struct Cloud {
    const float *x, *y, *z;  // SoA, padded to a multiple of 4 points with NaN
    const uint32_t *rgba;    // packed c0 | c1<<8 | c2<<16 | 255<<24
    uint64_t n;              // real point count
    int grid;                // workgroups of the grid-stride point kernels (lists are sized for it)
    int debug;               // timing experiments only (frames become wrong): bit1 T3 move, bit2/3/4 T4 min/acc/
                             // write-out, bit5 T4 no colour gather, bit6 T1 no list stores, bit7 T1 no LDS histogram
};

struct FilterLevels {
    float *lv[9];  // lv[0] = depth buffer (as float), lv[i] = level i
    int w[9], h[9];
    int levels;
};

// wave-private candidate lists (SoA) written by T1 and the tile-sorted copy made by T3
struct Lists {
    uint32_t *tiled, *depth, *idx;  // [num_waves * region_cap] each
    uint32_t *counts;               // [num_waves]
    uint64_t region_cap;
};
// peer-to-peer exchange (see the comment on k_p2p_sync): p[r] = rank r's buffer as mapped into this process
constexpr int kMaxPeers = 16;
struct PeerSet {
    void *p[kMaxPeers];
};
// A frame-sized buffer that still lies in per-rank slices of `chunk` elements (the reduced depth /
// the resolved image right after the slice kernels): element i is read from src.p[i / chunk].
// chunk == 0: not sliced, use the local buffer.
struct Sliced {
    PeerSet src;
    size_t chunk;
};
struct TilePyr {  // F1 folded into T4 (whole-frame calls with the default 4 levels)
    FilterLevels L;
    uint32_t n_eff_rows;
    uint32_t *part_min, *part_max;
    int enable;
};
struct Entry {  // 12 bytes, moved with one dwordx3 store
    uint32_t tiled, depth, idx;
};
struct Bins {
    Entry *entries;                             // entries counting-sorted by tile
    uint32_t *tile_hist, *tile_start, *cursor;  // [ntiles], [ntiles + 1], [ntiles]
    uint32_t *order;                            // [ntiles]: tile launch order of T4, heavy tiles first
    uint32_t *stats;                            // [2] in mapped HOST memory: entries of the frame, of its heaviest tile
    uint32_t *blk_hist;                         // [point-grid workgroups][ntiles]: T1's per-workgroup counts
};

void launch_clear(hipStream_t s, uint32_t *depth, uint32_t *acc, size_t npix);
// mode 0: the reference's structure (two full streams, global atomics)
void launch_min_depth(hipStream_t s, const Cloud &c, const Proj &P, int W, int H, uint32_t *depth);
void launch_accumulate(hipStream_t s, const Cloud &c, const Proj &P, int W, int H, const uint32_t *depth,
                       uint32_t *acc, float window);
// mode 1 (default): tile-binned pipeline -- T1 stream + lists + histogram, T2 scan + T3 scatter,
// T4 per-tile LDS z-buffer (tile mode 0 whole frame, 1 min only, 2 accumulate only)
constexpr int kDefaultPointGrid = 1024;
uint64_t list_region_cap(uint64_t n, int grid);  // entries per wave region
uint64_t list_num_waves(uint64_t n, int grid);   // number of wave regions
int tile_count(int W, int H);
// bounds != NULL enables per-chunk frustum culling (see k_project_bin)
void launch_project_bin(hipStream_t s, const Cloud &c, const Proj &P, int W, int H, const Lists &L,
                        uint32_t *tile_hist, uint32_t *blk_hist, const float *bounds);
void launch_chunk_bounds(hipStream_t s, const Cloud &c, float *bounds);  // 6 floats per 256 points
int reorder_morton(hipStream_t s, float *x, float *y, float *z, uint32_t *rgba, uint64_t n);  // rtr_reorder.hip
// occ (optional): 128 words, one bit per tile that has entries (peer-to-peer exchange)
void launch_bin_sort(hipStream_t s, const Cloud &c, int W, int H, const Lists &L, const Bins &B, uint32_t *occ = nullptr);
// depth_slices (mode 2 only): read the global depth from the ranks' reduced slices and store it to `depth`
void launch_tile(hipStream_t s, int mode, const Cloud &c, int W, int H, const Bins &B, float window, uint32_t *depth,
                 uint32_t *acc, uint8_t *img, int write_acc, const TilePyr *pyr, const Sliced *depth_slices = nullptr);
void launch_stream_probe(hipStream_t s, const Cloud &c, const Proj &P, int W, int H, uint32_t *sink, int variant);
void launch_resolve(hipStream_t s, const uint32_t *acc, uint8_t *img, size_t npix);
// img_slices (4 levels only): read the input image from the ranks' resolved slices (chunk in pixels)
void launch_filter(hipStream_t s, const FilterLevels &L, uint32_t *depth_bits, uint8_t *img, uint8_t *mask,
                   uint16_t *tensor, uint32_t *minmax, uint32_t *part_min, uint32_t *part_max, int W, int H,
                   float strength, float thr, int pyramid_parts, const Sliced *img_slices = nullptr);
void launch_p2p_sync(hipStream_t s, uint32_t *my_flags, const PeerSet &peer_flags, int rank, int world, uint32_t seq,
                     uint32_t *status, unsigned long long timeout_ticks);
constexpr int kP2POccBytes = 512;  // occupancy bitmap: one bit per screen tile (<= 4096)
void launch_p2p_occupancy(hipStream_t s, const uint32_t *tile_start, int W, int H, uint32_t *occ);
void launch_p2p_depth_reduce(hipStream_t s, const PeerSet &depth, const PeerSet &occ, uint32_t *red, size_t first,
                             size_t count, int world, int W, int H);
void launch_p2p_gather(hipStream_t s, const PeerSet &src, void *dst, size_t chunk_bytes, size_t nbytes,
                       int skip_owner);  // skip_owner < 0: copy every slice
void launch_p2p_acc_resolve(hipStream_t s, const PeerSet &acc, const PeerSet &occ, uint8_t *img, size_t first,
                            size_t count, int world, int W, int H);
void launch_generate(hipStream_t s, int scene, uint64_t seed, uint64_t first, uint64_t count, uint64_t total,
                     float *x, float *y, float *z, uint32_t *rgba);
void launch_aos_to_soa(hipStream_t s, const uint8_t *xyz, size_t xyz_stride, const uint8_t *rgb, size_t rgb_stride,
                       uint64_t count, float *x, float *y, float *z, uint32_t *rgba);
void launch_soa_to_aos(hipStream_t s, const float *x, const float *y, const float *z, const uint32_t *rgba,
                       uint64_t count, float *xyzw, uint8_t *rgba_out);
void launch_pad_nan(hipStream_t s, float *x, float *y, float *z, uint32_t *rgba, uint64_t n, uint64_t n_pad);

}  // namespace rtr
