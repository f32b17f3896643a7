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

// Lossless resident form of the coordinates for the tile-binned point kernel (option "pack").  A chunk =
// 256 consecutive points = what one wave handles per iteration (lane l: points 4 l .. 4 l + 3).  Per
// chunk and axis the fp32 BIT PATTERNS are stored as base | low bits: base = the leading bits all 256
// patterns share (low b bits clear), b = 0..25 or 32 the number of bits below that common prefix (0: the axis
// is constant in the chunk; 32: nothing is shared -- NaNs, mixed signs -- or more than 25 bits differ).  An
// axis block is the 256 values' low b bits as TWO little-endian bit streams: the A stream holds the FIRST value of
// every lane (lane l's b bits at bit b l: 8 b bytes), the B stream its other three (3 b bits at bit 3 b l: 24 b bytes) --
// the point kernel's lane test needs one point per lane, so nine chunks in ten are read through their A streams
// alone, a quarter of the bytes (round 4; before: one stream, lane l's four values at bit 4 b l).  A lane reads ONE
// dword-aligned 8-byte (16-byte) load per axis and stream and shifts its data down by the 0..31 bits its first value
// starts into that dword.  All A streams lie in one array, chunk after chunk (x, y, z), all B streams in another.
// hdr[2 c] = {base x, base y, base z, bx | by << 6 | bz << 12 | kPackWideFlag if some b is 32},
// hdr[2 c + 1] = {first 32-byte unit (lo, hi) -- the chunk's A streams start 8 bytes x that into the A array, its B
// streams 24 bytes x that into the B array --, lane spread of the chunk (fp32 bits, see Cloud::spread), 0}.
// Both arrays end with spare bytes (the last lanes' loads run past their values).
// Spatially ordered clouds need 16-21 bits per coordinate (neighbours share sign, exponent and leading
// mantissa bits): 5-8 B/pt instead of 12 (round 2 stored whole bytes: 6.5-9.2 B/pt).
constexpr uint32_t kPackMaxBits = 25;  // shift (<= 31) + b <= 64, + 3 b <= 128 bits of a lane's two loads
constexpr uint32_t kPackWideFlag = 1u << 18;
struct PackedXyz {
    const uint4 *hdr;        // null: not packed
    const uint32_t *planes;  // the A streams (every lane's first value); one allocation with ...
    const uint32_t *planes_b;  // ... the B streams (the other three values), pack_b_dwords(units) dwords behind
};
// where the B streams start inside the allocation, in dwords (units: 32-byte units of both streams together, the sum of
// the chunks' axis widths; the A streams end with ~1 KB of spare bytes), and the allocation's size
constexpr uint64_t pack_b_dwords(uint64_t units) { return (units * 2 + 256 + 16 + 15) & ~15ull; }  // (the point kernel requests 1 KB per chunk)
constexpr uint64_t pack_total_dwords(uint64_t units) { return pack_b_dwords(units) + units * 6 + 16; }

struct Cloud {
    const float *x, *y, *z;  // SoA, padded to a multiple of 4 points with NaN
    const uint32_t *rgba;    // packed c0 | c1<<8 | c2<<16 | 255<<24
    uint64_t n;              // real point count
    int grid;                // workgroups of the grid-stride point kernels
    int incoherent;          // consecutive points are unrelated (measured at upload): no wave-level claim groups
    PackedXyz pk;            // the same coordinates, packed (only read by launch_project_bin)
    // Lane spread (k_chunk_bounds): per 256-point chunk the largest |coordinate difference| between a lane's FIRST
    // point (4 l) and its other three (4 l + 1 .. 4 l + 3), over lanes and axes -- +inf when the chunk holds a
    // non-finite or huge (> 1e30) coordinate.  T1 tests one point per lane and bounds the other three by it (see
    // k_project_bin, "lane test").  null: no lane test.  absmax: largest finite |x|, |y|, |z| of the cloud.
    const float *spread;
    float absmax[3];
};

struct FilterLevels {
    float *lv[9];  // lv[0] = depth buffer (as float), lv[i] = level i
    int w[9], h[9];
    int levels;
};

// peer-to-peer exchange (see the comment on k_p2p_sync): p[r] = rank r's buffer as mapped into this process
constexpr int kMaxPeers = 16;
struct PeerSet {
    void *p[kMaxPeers];
};
// A frame-sized buffer that still lies in per-rank slices of `chunk` elements (the reduced depth /
// the resolved image right after the slice kernels): element i is read from src.p[i / chunk].
// chunk == 0: not sliced, use the local buffer.
// peers > 0 (accumulate pass of a sharded whole frame): src.p[r] is rank r's LOCAL depth buffer and the
// element is the MIN over the ranks whose bit for the tile is set in occ_all[r * 128 + (tile >> 5)] (every
// rank's occupancy bitmap, gathered into local memory by the barrier launch); the completed depth goes to `out`,
// a buffer no peer reads in that launch (NOT into the local depth buffer the peers are reading at that moment).
struct OwnedTab;
struct Sliced {
    PeerSet src;
    size_t chunk;
    const uint32_t *occ_all;
    int peers;
    uint32_t *out;
    int rank;             // tile mode 4 (owner-computes sharded frames): this rank, ...
    const OwnedTab *tab;  // ... and where every rank's tile store / frame buffers are mapped (device memory)
};
// Owner-computes form of the sharded frame (rtr_p2p_render_owned): every screen tile is produced by ONE of the ranks
// that have points in it (tile_owner below), which reads the other occupying ranks' entries straight out of their tile
// stores over xGMI -- no MIN / SUM exchange, no second pass.  The table lives in device memory (one pointer in the
// kernel arguments instead of half a kilobyte).
struct OwnedTab {
    const uint32_t *meta[kMaxPeers];   // rank r's TileStore::meta (stream lengths per tile: ts_cnt4, extent directory)
    const uint64_t *ext0[kMaxPeers];   // ... static extents
    const uint64_t *dyn[kMaxPeers];    // ... dynamic extent pool
    const uint32_t *depth[kMaxPeers];  // ... depth buffer (tiles it owns are final after the tile launch)
    const uint8_t *ximg[kMaxPeers];    // ... image exchange copy (ditto)
};
struct TilePyr {  // F1 folded into T4 (whole-frame calls with the default 4 levels)
    FilterLevels L;
    uint32_t n_eff_rows;
    uint32_t *part_min, *part_max;
    int enable;
};
// Tile store of the binned form.  Every 32x16 screen "storage tile" owns a stream of 8-byte
// entries (depth bits << 33 | in-tile pixel << 24 | colour), appended by T1 straight in tile order:
// the stream's first kS0 entries live in a static extent, later ones in extents of doubling size
// handed out on demand from `dyn` (extent k >= 1 holds stream positions [kS0 << (k-1), kS0 << k)).
constexpr int kS0Shift = 12;             // static extent: 4096 entries = 32 KB per storage tile
constexpr uint32_t kS0 = 1u << kS0Shift;
constexpr int kDirK = 21;                // extents 1..20 reach 2^32 entries per storage tile
constexpr int kHeavyExtra = 512;         // extra tile-kernel workgroups available for split tiles
struct TileStore {   // what travels as kernel argument (the point kernel is short of scalar registers)
    uint64_t *ext0;  // [nst * kS0]
    uint32_t *meta;  // every small array of the store in ONE allocation (see the ts_* accessors)
    uint32_t seq;    // 24-bit frame stamp, never 0
    int nst, ntiles; // 32x16 storage tiles, processing tiles
    int fill_shift;  // stream counter of storage tile st = fill[st << fill_shift] (one counter per 2^shift words)
};
constexpr int kFillShiftMax = 6;
struct StoreConsts {  // rarely needed, rarely changing: lives in the store's header (hdr[kHdrConsts ..])
    uint32_t *depth, *acc, *occ;  // frame buffers / occupancy bitmap T1's last workgroup writes to
    uint64_t *dyn;                // [dyn_cap] pool of the dynamic extents
    uint64_t dyn_cap;
    uint32_t heavy, slice;        // split tiles with more entries than `heavy` into slices of >= `slice`
    uint32_t *err_host;           // mapped host word: T1's epilogue ORs a frame's tile-store error code into it
    uint32_t *split_host;         // mapped host word: tiles above the split threshold in the frame T1 has just binned
    uint32_t *entries_host;       // mapped host word: in-frustum entries of the last frame whose statistics are complete
};
// meta, in 32-bit words:
//   fill[nst << kFillShiftMax]  stream length while T1 runs; zero between frames
//   (nst words, unused)
//   tile_cnt[nt]   entries per processing tile (32x32 or 64x32 pixels)
//   hctr[nt]       arrival counters of split tiles
//   items[(2 nt + kHeavyExtra + 1) * 8]  work list of the tile kernel, 32-byte records: word 0 = tile |
//                  slice << 12 | (slices - 1) << 22, words 1..4 = the lengths of the tile's two (four)
//                  streams.  Records [0, nt): one per tile, at position perm[tile] (kItemSkip when the
//                  tile is split); records [nt, nt + hdr[kHdrSplitItems]): the slices of split tiles
//   perm[nt]       tile -> record position: tiles that held more than twice the mean entry count in
//                  the PREVIOUS frame first (written by an extra workgroup of the tile launch)
//   hdr[32]        kHdr* below; hdr[kHdrConsts ..] = StoreConsts
//   ticket[2]      (u64) low word: groups of T1 workgroups that have finished (see sub[] below); high word: 256-point
//                  chunks whose colours were loaded
//   pool_next[2]   (u64) entries of `dyn` handed out in this frame
//   dir[nst * kDirK * 2]  (u64) extent base << 24 | frame stamp (valid iff stamp == seq)
// (every sub-array starts on a 16-byte boundary: the work list is read and written as uint4 records, the
// ticket, the pool cursor and the directory as 8-byte words)
__host__ __device__ constexpr size_t ts_align4(size_t x) { return (x + 3) & ~(size_t)3; }
__host__ __device__ constexpr size_t ts_off_count(int nst, int nt) { return ts_align4((size_t)nst << kFillShiftMax); }
__host__ __device__ constexpr size_t ts_off_tile_cnt(int nst, int nt) { return ts_off_count(nst, nt) + ts_align4((size_t)nst); }
__host__ __device__ constexpr size_t ts_off_hctr(int nst, int nt) { return ts_off_tile_cnt(nst, nt) + ts_align4((size_t)nt); }
__host__ __device__ constexpr size_t ts_off_items(int nst, int nt) { return ts_off_hctr(nst, nt) + ts_align4((size_t)nt); }
__host__ __device__ constexpr size_t ts_off_hdr(int nst, int nt) { return ts_off_items(nst, nt) + ((size_t)2 * nt + kHeavyExtra + 1) * 8; }
__host__ __device__ constexpr size_t ts_off_ticket(int nst, int nt) { return ts_off_hdr(nst, nt) + 32; }
__host__ __device__ constexpr size_t ts_off_pool(int nst, int nt) { return ts_off_ticket(nst, nt) + 2; }
__host__ __device__ constexpr size_t ts_off_dir(int nst, int nt) { return ts_off_pool(nst, nt) + 2; }
__host__ __device__ constexpr size_t ts_off_perm(int nst, int nt) { return ts_off_dir(nst, nt) + (size_t)nst * kDirK * 2; }
__host__ __device__ constexpr size_t ts_off_dbg(int nst, int nt) { return ts_off_perm(nst, nt) + ts_align4((size_t)nt); }
// cnt4[nt * 4]   the lengths of a tile's two (four) streams, indexed by TILE (the work list is in launch order): what a
//                peer reads of this rank's store in the owner-computes sharded form
__host__ __device__ constexpr size_t ts_off_cnt4(int nst, int nt) { return ts_align4(ts_off_dbg(nst, nt) + 128); }  // (dbg: 64 u64 time stamps, RTR_EXPERIMENT builds)
// sub[kSubTickets * 32]  (u64 at the start of each 128-byte line) first level of T1's ticket: the workgroups of the
//                point kernel finish within a few microseconds of each other, and 1024 returning adds on ONE word
//                serialise at ~11 ns each (the last workgroup learnt that it was last ~10 us after it had finished);
//                32 words on lines of their own take 32 adds each, and the last arrival of each adds to ticket[]
constexpr int kSubTickets = 32;
__host__ __device__ constexpr size_t ts_off_sub(int nst, int nt) { return ts_off_cnt4(nst, nt) + (size_t)nt * 4; }
// LEAN frames (whole single-GPU frames without split tiles: the usual case): T1 ends without ticket and epilogue; the
// tile kernel's workgroup at launch position b takes tile order[b], reads that tile's stream counters ITSELF and resets
// them; the frame's book-keeping (statistics, error word, pool cursor, next launch order) is done by the tile launch's
// one extra workgroup, off every critical path.
// order[2][nt]   launch position -> tile (the inverse of perm[]), per frame parity: the tile workgroups of a lean frame of
//                parity p read order[p] while that launch's extra workgroup writes order[p ^ 1] for the next frame
// lcnt[2][nt]    per frame parity: entries per tile of a lean frame, stored by its tile workgroups (plain stores: 2040
//                adds on a handful of statistics words would queue up on one memory channel and hold every tile
//                workgroup's first wave back -- measured: +13 us on the tile kernel); summed up into hdr[] by the NEXT lean
//                frame's extra workgroup (or by rtr_frame_stats), which sees them complete and stable
// lflag[4]       lflag[p] = 1 while lcnt[p] holds a frame that has not been folded into hdr[] yet
__host__ __device__ constexpr size_t ts_off_order(int nst, int nt) { return ts_off_sub(nst, nt) + (size_t)kSubTickets * 32; }
__host__ __device__ constexpr size_t ts_off_lcnt(int nst, int nt) { return ts_off_order(nst, nt) + 2 * ts_align4((size_t)nt); }
__host__ __device__ constexpr size_t ts_off_lflag(int nst, int nt) { return ts_off_lcnt(nst, nt) + 2 * ts_align4((size_t)nt); }
__host__ __device__ constexpr size_t ts_meta_words(int nst, int nt) { return ts_off_lflag(nst, nt) + 4; }
static_assert(ts_off_items(150, 75) % 4 == 0 && ts_off_hdr(150, 75) % 4 == 0 && ts_off_ticket(150, 75) % 2 == 0 &&
              ts_off_pool(150, 75) % 2 == 0 && ts_off_dir(150, 75) % 2 == 0 && ts_off_dbg(150, 75) % 2 == 0 &&
              ts_off_cnt4(150, 75) % 4 == 0,
              "tile store sub-arrays: 16-byte records / 8-byte words must be aligned (320x240: nst = 150)");
__host__ __device__ inline uint32_t *ts_fill(const TileStore &S) { return S.meta; }
__host__ __device__ inline uint32_t *ts_count(const TileStore &S) { return S.meta + ts_off_count(S.nst, S.ntiles); }
__host__ __device__ inline uint32_t *ts_tile_cnt(const TileStore &S) { return S.meta + ts_off_tile_cnt(S.nst, S.ntiles); }
__host__ __device__ inline uint32_t *ts_hctr(const TileStore &S) { return S.meta + ts_off_hctr(S.nst, S.ntiles); }
__host__ __device__ inline uint32_t *ts_items(const TileStore &S) { return S.meta + ts_off_items(S.nst, S.ntiles); }
__host__ __device__ inline uint32_t *ts_hdr(const TileStore &S) { return S.meta + ts_off_hdr(S.nst, S.ntiles); }
__host__ __device__ inline uint32_t *ts_ticket(const TileStore &S) { return S.meta + ts_off_ticket(S.nst, S.ntiles); }
__host__ __device__ inline unsigned long long *ts_pool(const TileStore &S) {
    return reinterpret_cast<unsigned long long *>(S.meta + ts_off_pool(S.nst, S.ntiles));
}
__host__ __device__ inline uint32_t *ts_perm(const TileStore &S) { return S.meta + ts_off_perm(S.nst, S.ntiles); }
constexpr uint32_t kItemSkip = 0xFFFFFFFEu;
__host__ __device__ inline unsigned long long *ts_dbg(const TileStore &S) {
    return reinterpret_cast<unsigned long long *>(S.meta + ts_off_dbg(S.nst, S.ntiles));
}
__host__ __device__ inline unsigned long long *ts_sub(const TileStore &S, uint32_t g) {
    return reinterpret_cast<unsigned long long *>(S.meta + ts_off_sub(S.nst, S.ntiles) + (size_t)g * 32);
}
__host__ __device__ inline uint32_t *ts_order(const TileStore &S, int parity) {
    return S.meta + ts_off_order(S.nst, S.ntiles) + (size_t)parity * ts_align4((size_t)S.ntiles);
}
__host__ __device__ inline uint32_t *ts_lcnt(const TileStore &S, int parity) {
    return S.meta + ts_off_lcnt(S.nst, S.ntiles) + (size_t)parity * ts_align4((size_t)S.ntiles);
}
__host__ __device__ inline uint32_t *ts_lflag(const TileStore &S) { return S.meta + ts_off_lflag(S.nst, S.ntiles); }
// (lean frames: T1's workgroups add their colour-chunk counts to the second 8-byte word of the sub-ticket lines)
__host__ __device__ inline unsigned long long *ts_sub_colour(const TileStore &S, uint32_t g) { return ts_sub(S, g) + 1; }
__host__ __device__ inline uint4 *ts_cnt4(const TileStore &S) { return reinterpret_cast<uint4 *>(S.meta + ts_off_cnt4(S.nst, S.ntiles)); }
__host__ __device__ inline unsigned long long *ts_dir(const TileStore &S) {
    return reinterpret_cast<unsigned long long *>(S.meta + ts_off_dir(S.nst, S.ntiles));
}
// kHdrError: the tile-store error code of the LAST frame (0: none; 1: an extent never appeared, 2: the extent pool
// overflowed -- entries were dropped), published by T1's epilogue from kHdrErrLive, which store_error() ORs into
// while T1 runs; kHdrColourChunks: 256-point chunks with at least one in-frustum point (each loads 1 KiB of colours)
enum { kHdrItems = 0, kHdrSplitItems = 1, kHdrEntries = 2, kHdrHeaviest = 3, kHdrSlice = 4, kHdrError = 5,
       kHdrSplitTiles = 6, kHdrColourChunks = 7, kHdrConsts = 8, kHdrErrLive = 28,
       // k_tile_split: slice records taken in its min phase, slices whose minima are in the depth buffer, records taken
       // in its second phase
       kHdrSplitQ1 = 29, kHdrSplitDone = 30, kHdrSplitQ2 = 31 };
static_assert(kHdrConsts + sizeof(StoreConsts) / 4 <= kHdrErrLive, "StoreConsts overlaps the header words behind it");
__host__ __device__ inline const StoreConsts *ts_consts(const TileStore &S) {
    return reinterpret_cast<const StoreConsts *>(ts_hdr(S) + kHdrConsts);
}

#ifdef RTR_EXPERIMENT
void read_filter_stamps(hipStream_t s, unsigned long long *out16);
#endif
void launch_clear(hipStream_t s, uint32_t *depth, uint32_t *acc, size_t npix);
// mode 0: the reference's structure (two full streams, global atomics)
void launch_min_depth(hipStream_t s, const Cloud &c, const Proj &P, int W, int H, uint32_t *depth);
void launch_accumulate(hipStream_t s, const Cloud &c, const Proj &P, int W, int H, const uint32_t *depth,
                       uint32_t *acc, float window);
// mode 1 (default): tile-binned pipeline -- T1 stream + lists + histogram, T2 scan + T3 scatter,
// T4 per-tile LDS z-buffer (tile mode 0 whole frame, 1 min only, 2 accumulate only)
constexpr int kDefaultPointGrid = 1024;
int tile_count(int W, int H);          // processing tiles (32x32, or 64x32 above 4096 of them)
int storage_tile_count(int W, int H);  // 32x16 storage tiles
// T1: stream the cloud once, append every in-frustum point to its storage tile's stream; the last
// workgroup to finish turns the stream lengths into the tile kernel's work list (and, with
// `clear_split`, resets depth / accumulators of the tiles that will be split; the frame buffers and
// the occupancy bitmap of the peer-to-peer exchange are named by StoreConsts in the store's header).
// bounds != NULL enables per-chunk frustum culling (see k_project_bin)
void launch_project_bin(hipStream_t s, const Cloud &c, const Proj &P, int W, int H, const TileStore &S,
                        const float *bounds, int clear_split, int phases, int xp = 0, hipEvent_t ev_start = nullptr,
                        hipEvent_t ev_stop = nullptr);  // ev_*: time stamps taken by the dispatch itself (timing on)
void launch_chunk_bounds(hipStream_t s, const Cloud &c, float *bounds, float *spread);  // 6 floats (+ 1) per 256 points
// packing (see PackedXyz): pack_measure fills hdr[2 nchunks] (bases, widths, block offsets by an exclusive
// scan) and *total_planes (device; in 32-byte units); pack_write fills the blocks; pack_verify counts the points whose decoded
// coordinates differ from the raw ones (must be 0) into *mismatches (device).  nchunks = ceil(ceil(n / 4) / 64).
void pack_measure(hipStream_t s, const Cloud &c, uint4 *hdr, uint32_t *chunk_planes, uint64_t *total_planes);  // (c.spread -> hdr[2 c + 1].z)
void pack_write(hipStream_t s, const Cloud &c, const uint4 *hdr, uint32_t *planes, uint32_t *planes_b);
void pack_verify(hipStream_t s, const Cloud &c, const uint4 *hdr, const uint32_t *planes, const uint32_t *planes_b, uint64_t *mismatches);
// x, y, z (padded to a multiple of 4 points) back from the packed form, bit for bit
void unpack_to_soa(hipStream_t s, const PackedXyz &pk, uint64_t n, float *x, float *y, float *z);
int reorder_morton(hipStream_t s, float *x, float *y, float *z, uint32_t *rgba, uint64_t n);  // rtr_reorder.hip
// mean diagonal of the 256-point chunk boxes / diagonal of the cloud's box, from launch_chunk_bounds' output
// absmax[3]: the largest finite |x|, |y|, |z| over the chunk boxes (+inf when no chunk has a finite box)
int order_quality(hipStream_t s, const float *bounds, uint64_t n, float *ratio, float absmax[3]);
// T4: per-tile LDS z-buffer over the tile store.  mode 0 = whole frame (min + accumulate + resolve
// of every unsplit tile, min phase of split tiles), 3 = second phase of the split tiles of a whole
// frame, 1 = min only, 2 = accumulate only, 4 = owner-computes sharded frame: like 0, but only the tiles
// tile_owner() gives to this rank, over the entries of EVERY occupying rank (depth_slices: occ_all, peers, rank, tab).
// write_acc: bit 0 (mode 0) also write the accumulators; bit 1 (modes 1, 2) this launch is the only writer of
// the frame buffer: store instead of folding into what memory holds; bit 2 (modes 1, 2; sharded frames)
// tiles without entries are not written at all (the peers only read tiles of the occupancy bitmap).
// depth_slices (mode 2 only): read the global depth from the ranks' reduced slices and store it to `depth`
// resets depth / accumulators under the tiles T1's epilogue decided to split (what the epilogue itself does when
// launch_project_bin's clear_split is set); for frames whose T1 overlapped the previous frame's tail
void launch_reset_split(hipStream_t s, int W, int H, const TileStore &S, uint32_t *depth, uint32_t *acc);
// write_acc bit 3 (mode 0): a LEAN frame (see ts_off_order) -- T1 ran without epilogue (launch_project_bin's flag 8);
// bit 4 (modes 0, 1): the frame's parity (which lcnt[] half its workgroups write, which order[] they read; the
// launch's extra workgroup writes the OTHER order[])
void launch_tile(hipStream_t s, int mode, int W, int H, const TileStore &S, float window, uint32_t *depth,
                 uint32_t *acc, uint8_t *img, int write_acc, const TilePyr *pyr, const Sliced *depth_slices = nullptr);
// folds the statistics of the lean frame of parity `parity` into the store's header now (rtr_frame_stats)
void launch_lean_fold(hipStream_t s, int W, int H, const TileStore &S, int parity);
void launch_stream_probe(hipStream_t s, const Cloud &c, const Proj &P, int W, int H, uint32_t *sink, int variant);
void launch_resolve(hipStream_t s, const uint32_t *acc, uint8_t *img, size_t npix);
// img_slices (4 levels only): read the input image from the ranks' resolved slices (chunk in pixels);
// depth_src (4 levels only): read the unfiltered depth from this buffer instead of depth_bits (which is written)
void launch_filter(hipStream_t s, const FilterLevels &L, uint32_t *depth_bits, uint8_t *img, uint8_t *mask,
                   uint16_t *tensor, uint32_t *minmax, uint32_t *part_min, uint32_t *part_max, int W, int H,
                   float strength, float thr, int pyramid_parts, const Sliced *img_slices = nullptr,
                   const uint32_t *depth_src = nullptr);
void launch_p2p_sync(hipStream_t s, uint32_t *my_flags, const PeerSet &peer_flags, int rank, int world, uint32_t seq,
                     uint32_t *status, unsigned long long timeout_ticks);
// the same barrier + every rank's occupancy bitmap gathered into occ_all[world * 128] (local memory)
void launch_p2p_sync_gather(hipStream_t s, uint32_t *my_flags, const PeerSet &peer_flags, int rank, int world, uint32_t seq,
                            uint32_t *status, unsigned long long timeout_ticks, const PeerSet &occ, uint32_t *occ_all);
constexpr int kP2POccBytes = 512;  // occupancy bitmap: one bit per screen tile (<= 4096)
// the frame owner's last step: every tile another rank produced is copied over xGMI into the local depth buffer /
// image (tiles nobody has points in are cleared) and the prefilter's pyramid levels / min-max partials emitted
void launch_p2p_collect(hipStream_t s, int W, int H, const OwnedTab *tab, const uint32_t *occ_all, int world, int rank,
                        uint32_t *depth, uint8_t *img, const TilePyr *pyr);
void launch_p2p_occupancy(hipStream_t s, const uint32_t *tile_cnt, int W, int H, uint32_t *occ);
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
// (a, b: device buffers padded to 16 bytes; ha, hb: device pointers of mapped pinned host buffers, padded likewise)
void launch_copy_to_host(hipStream_t s, const void *a, void *ha, size_t bytes_a, const void *b, void *hb, size_t bytes_b);
void launch_pad_nan(hipStream_t s, float *x, float *y, float *z, uint32_t *rgba, uint64_t n, uint64_t n_pad);

}  // namespace rtr
