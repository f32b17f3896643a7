// rtr_api.hip -- the C ABI of include/rtr.h on top of the gfx950 kernels.
//
// Host orchestration that replaces project_cloud.cu:189-434: the cloud stays
// resident in HBM as SoA, frame buffers and all pyramid levels are allocated once
// per resolution (the reference mallocs/frees 12 buffers per filtered frame,
// project_cloud.cu:346-390), every launch goes to one stream with no device-wide
// synchronisation in between (the reference calls cudaDeviceSynchronize ten times
// per frame), and the camera matrix travels as a kernel argument.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/rtr.h"
#include "rtr_kernels.h"

constexpr int kSplitCooldown = 8;  // whole frames keep launching k_tile_split this long after the last report of a tile above the threshold

struct rtr_ctx {
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    rtr_params prm{};
    std::string err;

    // resident cloud (SoA)
    float *x = nullptr, *y = nullptr, *z = nullptr;
    uint32_t *rgba = nullptr;
    float *bounds = nullptr;    // bounding box per 256-point chunk (frustum culling option)
    float *spread = nullptr;    // lane spread per 256-point chunk (rtr::Cloud::spread: T1's lane test)
    float absmax[3] = {0.f, 0.f, 0.f};  // largest finite |x|, |y|, |z| of the resident cloud
    int opt_lane_test = 1;      // T1 tests one point per lane first (option "lane_test")
    int opt_keep_soa = 0;       // 1: the fp32 SoA arrays stay resident beside the packed form (option "keep_soa")
    int opt_pool_worst = 0;     // 1: the extent pool is always sized for the worst case, 2 n entries (option "pool_worst_case")
    bool pool_worst = false;    // ... for this cloud: a frame overflowed the adaptive pool, or the peers map it
    uint32_t *entries_host = nullptr, *entries_dev = nullptr;  // mapped host word: entries of the last frame whose statistics are complete
    uint64_t entries_max = 0;   // the most entries a completed frame of this cloud has had
    float last_P[16] = {0};     // the last whole frame (rtr_render): what a synchronising call repeats when it learns that
    int last_filter = 0;        // the adaptive extent pool was too small for it
    bool last_valid = false;
    int opt_lean = 1;           // whole single-GPU frames without split tiles end T1 without its epilogue (option "lean")
    int opt_lean_identity = 1;  // lean frames: tile workgroup b takes tile b when the whole launch is resident (option "lean_identity")
    int opt_lean_early = -1;    // lean frames: first batch of entries requested before the stream counters are known: 0 never,
                                // 1 always, -1 when the previous frame's tiles were full (option "lean_early")
    bool last_lean = false;     // the last binned frame was a lean one (its statistics are folded on demand) ...
    int lean_parity = 0;        // ... and this was its parity
    uint64_t n = 0, cap = 0;
    uint4 *pk_hdr = nullptr;        // rtr::PackedXyz of the resident cloud (option "pack"); null: not in use
    uint32_t *pk_planes = nullptr, *pk_planes_b = nullptr;  // (one allocation: A streams, then B streams)
    uint64_t pk_bytes = 0;          // headers + planes
    int opt_pack = 1;               // 0 never, 1 when it saves >= 1/8 of the coordinate stream, 2 always + verified after packing

    // frame buffers
    int W = 0, H = 0;
    uint32_t *depth = nullptr, *acc = nullptr, *minmax = nullptr;
    uint8_t *img = nullptr, *mask = nullptr;
    uint32_t *part_min = nullptr, *part_max = nullptr;  // per-tile min / max partials of the prefilter
    uint16_t *tensor = nullptr;
    rtr::FilterLevels lv{};
    int lv_levels = 0;  // levels the pyramid was allocated for

    // tile-binned pipeline: the tile store T1 appends to and T4 reads (rtr_kernels.h)
    // (two sets: with option "overlap" T1 of frame k+1 fills one set on the front stream while the
    // tail of frame k still reads the other)
    struct FrontSet {
        rtr::TileStore store{};
        uint64_t pool_n = 0;        // point count the dynamic extent pool was sized for
        int nst = 0, ntiles = 0;    // tile counts the per-tile arrays were sized for
        rtr::StoreConsts consts{};  // host copy of the store's header constants
        uint64_t *dyn = nullptr;    // dynamic extent pool
        uint64_t dyn_cap = 0;
        hipEvent_t binned = nullptr, consumed = nullptr;  // T1 done (front stream) / T4 done (tail stream)
        bool consumed_valid = false;
    } fs[2];
    int cur = 0;
    FrontSet &F() { return fs[cur]; }
    int opt_overlap = 0;        // whole-frame renders run T1 on `front`, everything else on `stream`
    int opt_tail_cus = 0;       // CUs per XCD reserved for the tail stream when overlapping (0 = no CU masks)
    hipStream_t front = nullptr;
    hipStream_t masked_tail = nullptr;
    bool list_valid = false;    // bins match list_P / current cloud / resolution / window
    float list_P[12] = {0};
    int opt_mode = 1;           // 0 = two-pass global atomics (the reference's structure),
                                // 1 = tile-binned LDS z-buffer (default)
    int opt_keep_accum = 0;     // whole-frame calls also materialise RTR_BUF_ACCUM
    bool force_atomic = false;       // set around a whole frame that takes the atomic form (> 4096 tiles)
    int opt_heavy = 32768;           // tiles with more entries are split over several workgroups in T4 ...
    int opt_slice = 16384;           // ... into slices of at least this many entries
    int opt_p2p_timeout_ms = 2000;   // peer-to-peer flag barriers give up after this long (option "p2p_timeout_ms")
    int opt_fill_shift = -1;         // spacing of the stream counters, 4 << value bytes; -1: by the frames seen (see "fill_shift")
    int opt_debug_dyn_cap = -1;      // test aid: cap the dynamic extent pool at this many entries (-1: off)
    uint32_t *err_host = nullptr;    // mapped host word: tile-store error bits of frames since it was last read
    uint32_t *err_dev = nullptr;     // ... as the device sees it (StoreConsts::err_host)
    // whole frames launch k_tile_split (an empty launch costs ~5 us) only while tiles above the split threshold have
    // been seen: T1's epilogue stores their number here (mapped host word, read without a sync -- it describes the
    // last frame whose T1 has COMPLETED, the host may be frames ahead), and the launch stays on for kSplitCooldown
    // frames after the last such report, after an upload, a new resolution or new split options.  A frame that
    // turns out to need it while it is off is still exact (bin_epilogue, `no_split`).
    uint32_t *split_host = nullptr, *split_dev = nullptr;
    int split_cooldown = 0;
    int opt_xp = 0;                  // RTR_EXPERIMENT builds only (tools/kbench.py)
    int opt_phases = 0;         // T1: phase groups of the grid stride (option "phases", see k_project_bin); 0 = automatic
    int opt_probe = 0;          // rtr_stream_probe variant (experiments)
    int opt_cull = 0;           // per-chunk frustum culling in T1
    int opt_auto_reorder = 2;   // Morton-sort a cloud right after upload / generation: 0 never, 1 always, 2 when its
                                // 256-point chunks are not spatially compact (default)
    bool reordered = false;     // the resident cloud was sorted by the library
    float order_ratio = 0.f;    // mean chunk diagonal / cloud diagonal as uploaded
    int opt_grid = rtr::kDefaultPointGrid;  // workgroups of the point kernels

    // peer-to-peer exchange (rtr_p2p_*): own exchange buffers, the peers' mappings, barrier state
    struct P2P {
        uint32_t *red = nullptr;          // [npix] reduced depth; this rank writes its slice, peers read it
        uint8_t *ximg = nullptr;          // [3 * npix] resolved image; same (the prefilter rewrites RTR_BUF_IMAGE in place)
        uint32_t *flags = nullptr;        // [RTR_P2P_MAX_RANKS] uncached: barrier counters written by the peers
        uint32_t *occ = nullptr;          // [128] one bit per screen tile: this rank's frame has entries there
        uint32_t *occ_all = nullptr;      // [kMaxPeers * 128] every rank's bitmap, gathered by the first barrier of a frame
        bool depth_peers = false;         // rtr_p2p_render: the accumulate launch takes the MIN over the peers' depth itself
        bool depth_in_red = false;        // ... and has left the completed depth in `red` (the peers were reading `depth`)
        bool occ_current = false;         // occ was computed from the bins that are valid now
        bool occ_from_scan = false;       // ... by the epilogue of this frame's T1 (no separate launch)
        bool whole_frame = false;         // inside rtr_p2p_render: the tile launches are the only writers of depth /
                                          // accumulators (no clear, no read-modify-write) and T4<2> emits the pyramid
        bool pyramid_done = false;
        bool depth_sliced = false;        // ... RTR_BUF_DEPTH is completed by the accumulate launch (no depth gather)
        bool image_sliced = false;        // ... the prefilter reads the image from the ranks' slices (no image gather)
        bool acc_from_bins = false;       // the last accumulate pass used exactly those bins
        uint32_t *status_host = nullptr;  // mapped host word: barrier timeouts
        uint32_t *status_dev = nullptr;
        rtr::PeerSet depth{}, accum{}, image{}, reduced{}, flags_of{}, occ_of{}, meta_of{}, ext0_of{}, dyn_of{};
        rtr::OwnedTab *tab = nullptr;     // device copy of the peers' tile-store / frame-buffer mappings (owner-computes form)
        void *opened[9][RTR_P2P_MAX_RANKS] = {};
        int rank = 0, world = 0;
        uint32_t seq = 0;
        bool open = false;
    } p2p;

    // asynchronous host outputs (rtr_project_async): per slot a device snapshot of depth + image (so the next
    // frame's kernels may overwrite the frame buffers), pinned host buffers, and the events that order the two
    // streams -- the device-to-host copies run on `copy_stream` beside the next frame's kernels
    struct HostOut {
        uint8_t *img = nullptr, *dimg = nullptr;
        float *depth = nullptr;
        uint32_t *ddepth = nullptr;
        void *img_map = nullptr, *depth_map = nullptr;  // the pinned buffers as the device addresses them
        hipEvent_t snap = nullptr, done = nullptr;  // snapshot taken (frame stream) / copies finished (copy stream)
        bool busy = false;
    } ho[RTR_ASYNC_SLOTS];
    hipStream_t copy_stream = nullptr;

    // timing
    int timing = 0;  // 0 off, 1 every phase, 2 only the streaming kernel (RTR_K_MIN_DEPTH / ACCUMULATE), 3 = 2 on every 4th launch
    uint32_t timing_tick = 0;
    struct Span { hipEvent_t a, b; int k; };
    std::vector<Span> pending;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pool;
    double total_ms[RTR_K_COUNT] = {0};
    uint64_t launches[RTR_K_COUNT] = {0};
};

static thread_local std::string g_create_err;

static int fail(rtr_ctx *c, int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (c) c->err = buf; else g_create_err = buf;
    return code;
}

#define HIP_TRY(c, expr)                                                                         \
    do {                                                                                         \
        hipError_t e_ = (expr);                                                                  \
        if (e_ != hipSuccess)                                                                    \
            return fail((c), RTR_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                        __FILE__, __LINE__);                                                     \
    } while (0)

#define NEED(c, cond, msg) \
    do { if (!(cond)) return fail((c), RTR_ERR_INVALID, "%s", msg); } while (0)

namespace {

struct DevGuard {  // contexts pin their device for the duration of a call and hand the caller's back
    int prev = -1;
    explicit DevGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) (void)hipSetDevice(dev); else prev = -1;
    }
    ~DevGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
    DevGuard(const DevGuard &) = delete;
    DevGuard &operator=(const DevGuard &) = delete;
};

template <class T>
void dfree(T *&p) {
    if (p) (void)hipFree(p);
    p = nullptr;
}

void p2p_release(rtr_ctx *c) {  // the peers' mappings and this rank's exchange buffers
    auto &q = c->p2p;
    for (auto &kind : q.opened)
        for (auto &ptr : kind) {
            if (ptr) (void)hipIpcCloseMemHandle(ptr);
            ptr = nullptr;
        }
    dfree(q.red);
    dfree(q.ximg);
    dfree(q.flags);
    dfree(q.occ);
    dfree(q.occ_all);
    dfree(q.tab);
    q.open = false;
    q.world = 0;
    q.seq = 0;
}

struct Slice { size_t chunk, first, count; };
Slice p2p_slice(const rtr_ctx *c) {  // pixels owned by this rank: slices are multiples of 16 pixels
    const size_t npix = (size_t)c->W * c->H, w = (size_t)c->p2p.world;
    Slice s;
    s.chunk = (((npix + w - 1) / w) + 15) & ~(size_t)15;  // 16 pixels: 64 B of depth, 48 B of image
    s.first = s.chunk * (size_t)c->p2p.rank;
    if (s.first > npix) s.first = npix;
    s.count = (npix - s.first) < s.chunk ? npix - s.first : s.chunk;
    return s;
}
void free_host_out(rtr_ctx *c) {
    if (c->copy_stream) (void)hipStreamSynchronize(c->copy_stream);
    for (auto &h : c->ho) {
        if (h.img) (void)hipHostFree(h.img);
        if (h.depth) (void)hipHostFree(h.depth);
        dfree(h.dimg); dfree(h.ddepth);
        if (h.snap) (void)hipEventDestroy(h.snap);
        if (h.done) (void)hipEventDestroy(h.done);
        h = rtr_ctx::HostOut{};
    }
}

void free_frame(rtr_ctx *c) {
    c->list_valid = false;
    c->split_cooldown = kSplitCooldown;  // (a new resolution: nothing is known about its frames)
    free_host_out(c);
    p2p_release(c);
    dfree(c->depth); dfree(c->acc); dfree(c->img); dfree(c->mask); dfree(c->part_min); dfree(c->part_max); dfree(c->tensor);
    for (int i = 1; i <= 8; ++i) dfree(c->lv.lv[i]);
    for (auto &f : c->fs) {
        dfree(f.store.ext0); dfree(f.store.meta);
        f.nst = f.ntiles = 0;
        f.consts = rtr::StoreConsts{};
    }
    c->lv.lv[0] = nullptr;
    c->W = c->H = 0;
    c->lv_levels = 0;
}

void free_lists(rtr_ctx *c) {  // the dynamic extent pools (sized by the point count)
    if (c->p2p.open || c->p2p.red) p2p_release(c);  // (the peers map this rank's pool: they must re-open after a new cloud)
    for (auto &f : c->fs) {
        dfree(f.dyn);
        f.dyn_cap = 0;
        f.pool_n = 0;
    }
    c->list_valid = false;
    c->split_cooldown = kSplitCooldown;  // (a new cloud)
    c->pool_worst = false;
    c->entries_max = 0;
    if (c->entries_host) *c->entries_host = 0u;
    c->last_valid = false;
}

void free_pack(rtr_ctx *c) {
    dfree(c->pk_hdr); dfree(c->pk_planes);
    c->pk_planes_b = nullptr;
    c->pk_bytes = 0;
}

void free_cloud(rtr_ctx *c) {
    dfree(c->x); dfree(c->y); dfree(c->z); dfree(c->rgba); dfree(c->bounds); dfree(c->spread);
    free_pack(c);
    free_lists(c);
    c->n = c->cap = 0;
}

// Per-resolution part of the tile store: a static 32 KB extent, a stream length, an extent directory
// per 32x16 storage tile, and the tile kernel's work list.  s: the stream T1 will run on.
int ensure_tiles(rtr_ctx *c, hipStream_t s) {
    const int nt = rtr::tile_count(c->W, c->H), nst = rtr::storage_tile_count(c->W, c->H);
    auto &f = c->F();
    auto &t = f.store;
    if (!(t.ext0 && f.nst == nst && f.ntiles == nt)) {
        dfree(t.ext0); dfree(t.meta);
        f.nst = f.ntiles = 0;
        f.consts = rtr::StoreConsts{};
        c->list_valid = false;
        const size_t meta_bytes = rtr::ts_meta_words(nst, nt) * sizeof(uint32_t);
        // (+ 16 entries of slack: the tile kernel's sweeps read a few entries past the piece they are masking)
        HIP_TRY(c, hipMalloc((void **)&t.ext0, ((size_t)nst * rtr::kS0 + 16) * sizeof(uint64_t)));
        HIP_TRY(c, hipMalloc((void **)&t.meta, meta_bytes));
        HIP_TRY(c, hipMemsetAsync(t.meta, 0, meta_bytes, s));  // stream lengths 0, directory stamps 0 (never current)
        t.seq = 0;
        t.nst = f.nst = nst;
        t.ntiles = f.ntiles = nt;
        {   // launch order of the tile kernel: identity until a frame has been rendered
            std::vector<uint32_t> ident((size_t)nt);
            for (int i = 0; i < nt; ++i) ident[(size_t)i] = (uint32_t)i;
            HIP_TRY(c, hipMemcpyAsync(rtr::ts_perm(t), ident.data(), (size_t)nt * 4, hipMemcpyHostToDevice, s));
            HIP_TRY(c, hipMemcpyAsync(rtr::ts_order(t, 0), ident.data(), (size_t)nt * 4, hipMemcpyHostToDevice, s));
            HIP_TRY(c, hipMemcpyAsync(rtr::ts_order(t, 1), ident.data(), (size_t)nt * 4, hipMemcpyHostToDevice, s));
            HIP_TRY(c, hipStreamSynchronize(s));  // `ident` goes out of scope
        }
    }
    // header constants: the buffers T1's last workgroup resets for split tiles / writes the occupancy
    // bitmap to, the dynamic extent pool, the split parameters (uploaded only when one of them changes)
    rtr::StoreConsts want{};
    want.depth = c->depth; want.acc = c->acc; want.occ = c->p2p.open ? c->p2p.occ : nullptr;
    want.dyn = f.dyn; want.dyn_cap = f.dyn_cap;
    if (c->opt_debug_dyn_cap >= 0 && (uint64_t)c->opt_debug_dyn_cap < want.dyn_cap) want.dyn_cap = (uint64_t)c->opt_debug_dyn_cap;
    want.err_host = c->err_dev;
    want.split_host = c->split_dev;
    want.entries_host = c->entries_dev;
    want.heavy = c->opt_heavy > 0 ? (uint32_t)c->opt_heavy : 0xFFFFFFFFu;
    want.slice = (uint32_t)c->opt_slice;
    if (memcmp(&want, &f.consts, sizeof want) != 0) {
        f.consts = want;
        HIP_TRY(c, hipMemcpyAsync(rtr::ts_hdr(t) + rtr::kHdrConsts, &f.consts, sizeof f.consts, hipMemcpyHostToDevice, s));
    }
    return RTR_OK;
}

// The dynamic extents of one frame sum to less than twice its entries (every extent doubles its stream,
// rtr_kernels.h), and a frame has at most n entries: 2 n + 64 entries = 16 B per point is the worst case (round 1's
// wave lists + sorted copy: 24).  An ordinary view has a few per cent of the cloud inside the frustum, so the pool is
// sized by the frames this cloud has had: 8 x the most entries a completed frame reported (a mapped host word, read
// without a sync), at least n / 2 and 2^20 -- 4 B per point (n / 4 re-allocated in the middle of BASELINE C2's frames:
// a sync, a free and a malloc cost more than the memory is worth).  A frame whose entries jump past that (the camera suddenly
// sees four times more of the cloud than ever before) overflows the pool, reports it (tile-store error 2), and the next
// synchronising call grows the pool to the worst case and renders the frame again (finish_sync) -- the caller never
// sees it, unless it consumes frames on the stream without ever synchronising: option "pool_worst_case" is for that.
hipError_t sync_streams(rtr_ctx *c);
uint64_t pool_worst_cap(const rtr_ctx *c) { return 2 * c->n + 64; }
uint64_t pool_want_cap(rtr_ctx *c, uint64_t have) {
    const uint64_t worst = pool_worst_cap(c);
    if (c->opt_pool_worst || c->pool_worst || c->p2p.open) return worst;
    if (c->entries_host) {
        const uint64_t e = __atomic_load_n(c->entries_host, __ATOMIC_RELAXED);
        if (e > c->entries_max) c->entries_max = e;
    }
    uint64_t floor_ = c->n / 2 > (1ull << 20) ? c->n / 2 : (1ull << 20);
    // (hysteresis: grown to 8 x when the head-room over the densest frame seen falls under 4 x)
    uint64_t want = have >= 4 * c->entries_max && have >= floor_ ? have : (8 * c->entries_max > floor_ ? 8 * c->entries_max : floor_);
    return want < worst ? want : worst;
}
int ensure_lists(rtr_ctx *c) {
    auto &f = c->F();
    const uint64_t want = pool_want_cap(c, f.pool_n == c->n ? f.dyn_cap : 0);
    if (f.dyn && f.pool_n == c->n && f.dyn_cap >= want) return RTR_OK;
    // (the peers map the pool that was EXPORTED -- set 0's: export / open again after a new cloud.  The second set's
    // pool, first allocated by a frame with option "overlap", is nobody else's business)
    if (&f == &c->fs[0] && (c->p2p.open || c->p2p.red)) p2p_release(c);
    HIP_TRY(c, sync_streams(c));  // (frames in flight may still read the old pool)
    dfree(f.dyn);
    c->list_valid = false;
    f.dyn_cap = want;
    f.pool_n = c->n;
    HIP_TRY(c, hipMalloc((void **)&f.dyn, f.dyn_cap * sizeof(uint64_t)));
    return RTR_OK;
}

// The fp32 SoA arrays of a cloud that is resident in packed form only (option "keep_soa" = 0, the default): decoded
// from the packed form -- bit for bit, it is lossless -- for the calls that read fp32 coordinates (the atomic form, the
// sort, rtr_download_points, option "pack" = 0, the stream probe).  They stay until the cloud is packed again.
int ensure_soa(rtr_ctx *c) {
    if (c->x) return RTR_OK;
    if (!c->pk_hdr || c->cap == 0) return fail(c, RTR_ERR_INTERNAL, "no resident coordinates");
    HIP_TRY(c, hipMalloc((void **)&c->x, c->cap * 4));
    HIP_TRY(c, hipMalloc((void **)&c->y, c->cap * 4));
    HIP_TRY(c, hipMalloc((void **)&c->z, c->cap * 4));
    rtr::unpack_to_soa(c->stream, rtr::PackedXyz{c->pk_hdr, c->pk_planes, c->pk_planes_b}, c->n, c->x, c->y, c->z);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(c, RTR_ERR_HIP, "unpack launch failed: %s", hipGetErrorString(e));
    return RTR_OK;
}
void drop_soa(rtr_ctx *c) {  // after the cloud has been packed: 12 B per point back
    if (!c->pk_hdr || c->opt_keep_soa || !c->x) return;
    (void)sync_streams(c);
    dfree(c->x); dfree(c->y); dfree(c->z);
}

int alloc_cloud(rtr_ctx *c, uint64_t n) {
    uint64_t n_pad = (n + 3) & ~3ull;
    if (n_pad == 0) n_pad = 4;
    if (n_pad > c->cap) {
        free_cloud(c);
        HIP_TRY(c, hipMalloc((void **)&c->x, n_pad * 4));
        HIP_TRY(c, hipMalloc((void **)&c->y, n_pad * 4));
        HIP_TRY(c, hipMalloc((void **)&c->z, n_pad * 4));
        HIP_TRY(c, hipMalloc((void **)&c->rgba, n_pad * 4));
        HIP_TRY(c, hipMalloc((void **)&c->bounds, ((n_pad / 4 + 63) / 64) * 6 * sizeof(float)));
        HIP_TRY(c, hipMalloc((void **)&c->spread, ((n_pad / 4 + 63) / 64) * sizeof(float)));
        c->cap = n_pad;
    }
    if (!c->x) {  // (the previous cloud was resident in packed form only)
        HIP_TRY(c, hipMalloc((void **)&c->x, c->cap * 4));
        HIP_TRY(c, hipMalloc((void **)&c->y, c->cap * 4));
        HIP_TRY(c, hipMalloc((void **)&c->z, c->cap * 4));
    }
    if (n != c->n) {  // (the adaptive extent pool starts over; the peers of a sharded frame must map the new one)
        if (c->p2p.open || c->p2p.red) p2p_release(c);
        c->pool_worst = false;
        c->entries_max = 0;
        if (c->entries_host) *c->entries_host = 0u;
    }
    c->n = n;
    c->list_valid = false;
    c->last_valid = false;
    c->split_cooldown = kSplitCooldown;  // (a new cloud: nothing is known about its frames)
    return RTR_OK;
}

rtr::Proj make_proj(const float P[16]) {
    rtr::Proj p;
    for (int i = 0; i < 12; ++i) p.m[i] = P[i];
    return p;
}

rtr::Cloud cloud_of(const rtr_ctx *c) {
    // (a chunk of 256 points spanning more than half of the cloud: consecutive points are unrelated.  A hash-ordered
    // cloud measures ~1.0; the reference loader's 0.25 m blocks in hash-map order, unordered inside, measure 0.28 for a
    // 10 m room and must keep the wave-level claim groups: 0.33 ms instead of 0.66 ms per frame without them)
    return rtr::Cloud{c->x, c->y, c->z, c->rgba, c->n, c->opt_grid, (!c->reordered && c->order_ratio > 0.5f) ? 1 : 0,
                      rtr::PackedXyz{c->pk_hdr, c->pk_planes, c->pk_planes_b}, c->spread, {c->absmax[0], c->absmax[1], c->absmax[2]}};
}

struct Timed {  // brackets one phase with hipEvents on the stream it is launched on
    rtr_ctx *c; int k; hipStream_t s; hipEvent_t a = nullptr, b = nullptr;
    bool in_dispatch;  // the launch itself carries the two events (hipExtLaunchKernelGGL): nothing to record here
    Timed(rtr_ctx *c_, int k_, hipStream_t s_ = nullptr, bool in_dispatch_ = false)
        : c(c_), k(k_), s(s_ ? s_ : c_->stream), in_dispatch(in_dispatch_) {
        if (!c->timing) return;
        if (c->timing >= 2 && k != RTR_K_MIN_DEPTH && k != RTR_K_ACCUMULATE) return;
        if (c->timing == 3 && (c->timing_tick++ & 3u) != 0u) return;  // a bracket costs ~8-10 us of stream time
        if (c->timing == 4 && (c->timing_tick++ & 1u) != 0u) return;
        if (c->pool.empty()) {
            (void)hipEventCreate(&a); (void)hipEventCreate(&b);
        } else {
            a = c->pool.back().first; b = c->pool.back().second; c->pool.pop_back();
        }
        if (!in_dispatch) (void)hipEventRecord(a, s);
    }
    ~Timed() {
        if (!a) return;
        if (!in_dispatch) (void)hipEventRecord(b, s);
        c->pending.push_back({a, b, k});
    }
};

// everything queued by this context is finished (the tail stream waits for the front stream's
// T1 of every frame it completes, so the order below drains both)
hipError_t sync_streams(rtr_ctx *c) {
    if (c->front) {
        hipError_t e = hipStreamSynchronize(c->front);
        if (e != hipSuccess) return e;
    }
    return hipStreamSynchronize(c->stream);
}

// Tile-store errors (entries dropped by T1: an extent that never appeared, an exhausted extent pool) reach the
// host through a mapped word that T1's epilogue writes; every call that has just synchronised reports and
// clears it -- a wrong frame is never returned as RTR_OK.
// retry (out): the only error is an overflow of the ADAPTIVE extent pool (ensure_lists) -- the pool is worst-case sized
// from now on and the caller renders the frame again instead of failing.
int check_store_error(rtr_ctx *c, bool *retry = nullptr) {
    if (retry) *retry = false;
    if (!c->err_host) return RTR_OK;
    const uint32_t e = __atomic_exchange_n(c->err_host, 0u, __ATOMIC_ACQUIRE);
    if (e == 0u) return RTR_OK;
    if (e == 2u && c->opt_debug_dyn_cap < 0 && c->F().dyn_cap < pool_worst_cap(c)) {
        c->pool_worst = true;  // (ensure_lists re-allocates before the next T1)
        if (retry) {
            *retry = true;
            return RTR_OK;
        }
        return fail(c, RTR_ERR_INTERNAL, "tile store error 0x2: the extent pool, sized by the frames seen so far, was too small "
                    "for a frame rendered since the last synchronising call -- entries were dropped; the pool is worst-case "
                    "sized from now on: render the frame again (option pool_worst_case = 1 sizes it so from the start)");
    }
    return fail(c, RTR_ERR_INTERNAL, "tile store error 0x%x: %s%s%s%s-- entries were dropped, frames rendered since the last "
                "synchronising call are incomplete", e, (e & 1u) ? "a stream extent never appeared " : "",
                (e & 2u) ? "the dynamic extent pool overflowed " : "",
                (e & 4u) ? "a contested tile of a sharded frame had more stream pieces than its table holds " : "",
                (e & 8u) ? "the split tiles' second phase gave up waiting for the first " : "");
}

// after the last reader of the active list / bin set has been queued on the tail stream
void mark_consumed(rtr_ctx *c) {
    if (!c->front) return;
    auto &f = c->F();
    if (hipEventRecord(f.consumed, c->stream) == hipSuccess) f.consumed_valid = true;
}

int collect_timing(rtr_ctx *c) {
    if (c->pending.empty()) return RTR_OK;
    HIP_TRY(c, sync_streams(c));
    for (auto &s : c->pending) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, s.a, s.b) == hipSuccess) {
            c->total_ms[s.k] += ms;
            c->launches[s.k] += 1;
        }
        c->pool.emplace_back(s.a, s.b);
    }
    c->pending.clear();
    return RTR_OK;
}

int check_frame(rtr_ctx *c) {
    NEED(c, c->W > 0 && c->H > 0, "rtr_set_resolution has not been called");
    return RTR_OK;
}

int launch_check(rtr_ctx *c, const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(c, RTR_ERR_HIP, "%s launch failed: %s", what, hipGetErrorString(e));
    return RTR_OK;
}

int ensure_pyramid(rtr_ctx *c) {
    const int L = c->prm.levels;
    if (L < 1 || L > 8) return fail(c, RTR_ERR_UNSUPPORTED, "levels must be in 1..8");
    if ((c->W % (1 << L)) != 0 || (c->H >> L) < 1)
        return fail(c, RTR_ERR_UNSUPPORTED,
                    "prefilter needs W %% 2^levels == 0 and H >= 2^levels (got %dx%d, levels %d): the reference's "
                    "pyramid strides are only defined then (project_cloud.cu:39,336-362)", c->W, c->H, L);
    if (c->lv_levels == L) return RTR_OK;
    for (int i = 1; i <= 8; ++i) dfree(c->lv.lv[i]);
    c->lv.levels = L;
    c->lv.lv[0] = reinterpret_cast<float *>(c->depth);
    c->lv.w[0] = c->W; c->lv.h[0] = c->H;
    for (int i = 1; i <= L; ++i) {
        c->lv.w[i] = c->lv.w[i - 1] / 2;
        c->lv.h[i] = c->lv.h[i - 1] / 2;
        HIP_TRY(c, hipMalloc((void **)&c->lv.lv[i], sizeof(float) * (size_t)c->lv.w[i] * c->lv.h[i]));
    }
    c->lv_levels = L;
    return RTR_OK;
}

}  // namespace

static int set_overlap(rtr_ctx *c, bool on);

extern "C" {

int rtr_abi_version(void) { return RTR_ABI_VERSION; }

void rtr_default_params(rtr_params *p) {
    if (!p) return;
    p->depth_window = 0.02f;        // render.cu:106
    p->filter_strength = 1.025f;    // project_cloud.cu:24
    p->gradient_threshold = 0.03f;  // project_cloud.cu:25
    p->levels = 4;                  // project_cloud.cu:23
}

int rtr_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

int rtr_create(rtr_ctx **out, int device) {
    if (!out) return fail(nullptr, RTR_ERR_INVALID, "out is NULL");
    *out = nullptr;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev == 0)
        return fail(nullptr, RTR_ERR_HIP, "no HIP device available (%s): this library has no CPU fallback",
                    e != hipSuccess ? hipGetErrorString(e) : "device count 0");
    if (device < 0 || device >= ndev) return fail(nullptr, RTR_ERR_INVALID, "device %d out of range [0,%d)", device, ndev);
    rtr_ctx *c = new rtr_ctx();
    c->device = device;
    rtr_default_params(&c->prm);
    DevGuard g(device);
    e = hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        int rc = fail(nullptr, RTR_ERR_HIP, "hipStreamCreate failed: %s", hipGetErrorString(e));
        delete c;
        return rc;
    }
    c->stream = c->own_stream;
    e = hipMalloc((void **)&c->minmax, 2 * sizeof(uint32_t));
    if (e == hipSuccess) e = hipHostMalloc((void **)&c->err_host, sizeof(uint32_t), hipHostMallocMapped);
    if (e == hipSuccess) {
        *c->err_host = 0u;
        void *d = nullptr;
        e = hipHostGetDevicePointer(&d, c->err_host, 0);
        c->err_dev = static_cast<uint32_t *>(d);
    }
    if (e == hipSuccess) e = hipHostMalloc((void **)&c->split_host, sizeof(uint32_t), hipHostMallocMapped);
    if (e == hipSuccess) {
        *c->split_host = 0u;
        void *d = nullptr;
        e = hipHostGetDevicePointer(&d, c->split_host, 0);
        c->split_dev = static_cast<uint32_t *>(d);
        c->split_cooldown = kSplitCooldown;
    }
    if (e == hipSuccess) e = hipHostMalloc((void **)&c->entries_host, sizeof(uint32_t), hipHostMallocMapped);
    if (e == hipSuccess) {
        *c->entries_host = 0u;
        void *d = nullptr;
        e = hipHostGetDevicePointer(&d, c->entries_host, 0);
        c->entries_dev = static_cast<uint32_t *>(d);
    }
    if (e != hipSuccess) {
        int rc = fail(nullptr, RTR_ERR_HIP, "hipMalloc failed: %s", hipGetErrorString(e));
        if (c->minmax) (void)hipFree(c->minmax);
        if (c->err_host) (void)hipHostFree(c->err_host);
        if (c->split_host) (void)hipHostFree(c->split_host);
        if (c->entries_host) (void)hipHostFree(c->entries_host);
        (void)hipStreamDestroy(c->own_stream);
        delete c;
        return rc;
    }
    *out = c;
    return RTR_OK;
}

int rtr_destroy(rtr_ctx *c) {
    if (!c) return RTR_OK;
    DevGuard g(c->device);
    (void)sync_streams(c);
    (void)collect_timing(c);
    (void)set_overlap(c, false);
    for (auto &p : c->pool) { (void)hipEventDestroy(p.first); (void)hipEventDestroy(p.second); }
    free_frame(c);
    free_cloud(c);
    dfree(c->minmax);
    if (c->p2p.status_host) (void)hipHostFree(c->p2p.status_host);
    if (c->split_host) (void)hipHostFree(c->split_host);
    if (c->err_host) (void)hipHostFree(c->err_host);
    if (c->entries_host) (void)hipHostFree(c->entries_host);
    if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    delete c;
    return RTR_OK;
}

const char *rtr_last_error(const rtr_ctx *c) { return c ? c->err.c_str() : g_create_err.c_str(); }

int rtr_set_params(rtr_ctx *c, const rtr_params *p) {
    if (!c) return RTR_ERR_INVALID;
    NEED(c, p != nullptr, "params is NULL");
    NEED(c, p->levels >= 1 && p->levels <= 8, "levels must be in 1..8");
    c->prm = *p;
    c->list_valid = false;
    return RTR_OK;
}

// Option "overlap": whole-frame renders put T1 on a second stream and alternate between two
// list / bin sets, so the HBM-bound stream of frame k+1 runs beside the latency-bound tail of
// frame k.  With "tail_cus" = t > 0 the two streams get disjoint CU masks (t CUs of every XCD for
// the tail, the rest for T1) so neither takes the other's wave slots.  The bit -> CU mapping of a
// mask is not documented for 8-XCD parts (XCD-interleaved or XCD-blocked); the pattern below
// gives every XCD exactly t tail CUs under both numberings.
static int set_overlap(rtr_ctx *c, bool on) {
    DevGuard g(c->device);
    HIP_TRY(c, sync_streams(c));
    (void)collect_timing(c);
    if (!on) {
        if (c->masked_tail && c->stream == c->masked_tail) c->stream = c->own_stream;
        if (c->front) (void)hipStreamDestroy(c->front);
        if (c->masked_tail) (void)hipStreamDestroy(c->masked_tail);
        c->front = c->masked_tail = nullptr;
        for (auto &f : c->fs) {
            if (f.binned) (void)hipEventDestroy(f.binned);
            if (f.consumed) (void)hipEventDestroy(f.consumed);
            f.binned = f.consumed = nullptr;
            f.consumed_valid = false;
        }
        c->opt_overlap = 0;
        return RTR_OK;
    }
    if (c->opt_overlap) return RTR_OK;
    // (the peers of a sharded frame read THE exported tile store; alternating between two of them is for single-GPU frames)
    if (c->p2p.open) return fail(c, RTR_ERR_INVALID, "option overlap cannot be switched on while the peer-to-peer exchange is open (rtr_p2p_close first)");
    int ncu = 0;
    HIP_TRY(c, hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, c->device));
    if (c->opt_tail_cus > 0) {
        if (ncu != 256) return fail(c, RTR_ERR_UNSUPPORTED, "tail_cus needs the 256-CU / 8-XCD part (device has %d CUs)", ncu);
        uint32_t tail[8] = {0}, head[8] = {0};
        for (int i = 0; i < 256; ++i) {
            int a = i % 8, b = (i / 8) % 4, x = i / 32;
            bool is_tail = b * 8 + (a + x) % 8 < c->opt_tail_cus;
            (is_tail ? tail : head)[i / 32] |= 1u << (i % 32);
        }
        HIP_TRY(c, hipExtStreamCreateWithCUMask(&c->front, 8, head));
        hipError_t e = hipExtStreamCreateWithCUMask(&c->masked_tail, 8, tail);
        if (e != hipSuccess) {
            (void)hipStreamDestroy(c->front);
            c->front = nullptr;
            return fail(c, RTR_ERR_HIP, "hipExtStreamCreateWithCUMask failed: %s", hipGetErrorString(e));
        }
        if (c->stream == c->own_stream) c->stream = c->masked_tail;
    } else {
        HIP_TRY(c, hipStreamCreateWithFlags(&c->front, hipStreamNonBlocking));
    }
    for (auto &f : c->fs) {
        hipError_t e = hipEventCreateWithFlags(&f.binned, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&f.consumed, hipEventDisableTiming);
        if (e != hipSuccess) {
            int rc = fail(c, RTR_ERR_HIP, "hipEventCreate failed: %s", hipGetErrorString(e));
            c->opt_overlap = 1;  // so that the teardown below runs
            (void)set_overlap(c, false);
            return rc;
        }
        f.consumed_valid = false;
    }
    c->opt_overlap = 1;
    return RTR_OK;
}

static int pack_cloud(rtr_ctx *c);

int rtr_set_option(rtr_ctx *c, const char *key, int value) {
    if (!c) return RTR_ERR_INVALID;
    NEED(c, key != nullptr, "key is NULL");
    if (!strcmp(key, "mode")) {
        NEED(c, value == 0 || value == 1, "mode must be 0 or 1");
        c->opt_mode = value;
        c->list_valid = false;
        return RTR_OK;
    }
    if (!strcmp(key, "point_grid")) {
        NEED(c, value >= 1 && value <= 65535, "point_grid out of range");
        c->opt_grid = value;
        c->list_valid = false;
        return RTR_OK;
    }
    if (!strcmp(key, "split_threshold")) {  // 0: never split
        NEED(c, value >= 0, "split_threshold must be >= 0");
        c->opt_heavy = value;
        c->split_cooldown = kSplitCooldown;
        c->list_valid = false;
        return RTR_OK;
    }
#ifdef RTR_EXPERIMENT
    if (!strcmp(key, "xp")) {  // timing experiments: parts of T1 switched off, frames become wrong
        c->opt_xp = value;
        return RTR_OK;
    }
#endif
    if (!strcmp(key, "fill_shift")) {  // spacing of the tile stream counters: 4 << value bytes
        NEED(c, value >= -1 && value <= rtr::kFillShiftMax, "fill_shift must be in -1..6 (-1: automatic)");
        DevGuard g(c->device);
        HIP_TRY(c, sync_streams(c));
        c->opt_fill_shift = value;  // (the counters are all zero between frames: any spacing can follow any other)
        c->list_valid = false;
        return RTR_OK;
    }
    if (!strcmp(key, "debug_dyn_cap")) {  // test aid: a pool this small makes heavy tiles overflow it (error code 2)
        NEED(c, value >= -1, "debug_dyn_cap must be >= -1 (-1: off)");
        c->opt_debug_dyn_cap = value;
        c->list_valid = false;
        return RTR_OK;
    }
    if (!strcmp(key, "p2p_timeout_ms")) {
        NEED(c, value >= 1 && value <= 60000, "p2p_timeout_ms must be in 1..60000");
        c->opt_p2p_timeout_ms = value;
        return RTR_OK;
    }
    if (!strcmp(key, "phases")) {  // T1: wave groups that start at different places of the cloud (k_project_bin)
        NEED(c, value >= 0 && value <= 65535, "phases must be in 0..65535 (0: automatic)");
        c->opt_phases = value;
        c->list_valid = false;
        return RTR_OK;
    }
    if (!strcmp(key, "split_slice")) {
        NEED(c, value >= 1, "split_slice must be >= 1");
        c->opt_slice = value;
        c->split_cooldown = kSplitCooldown;
        c->list_valid = false;
        return RTR_OK;
    }
    if (!strcmp(key, "auto_reorder")) {
        NEED(c, value >= 0 && value <= 2, "auto_reorder must be 0 (never), 1 (always) or 2 (when the order is incoherent)");
        c->opt_auto_reorder = value;
        return RTR_OK;
    }
    if (!strcmp(key, "cull")) {
        c->opt_cull = value != 0;
        return RTR_OK;
    }
    if (!strcmp(key, "lean")) {  // whole frames without T1's epilogue when nothing needs it (rtr_render)
        c->opt_lean = value != 0;
        return RTR_OK;
    }
    if (!strcmp(key, "lean_identity")) {  // lean frames: workgroup b = tile b when every tile workgroup is resident at once
        c->opt_lean_identity = value != 0;
        return RTR_OK;
    }
    if (!strcmp(key, "lean_early")) {  // lean frames: entries requested before the counters (tile_body); -1 = by the last frame
        NEED(c, value >= -1 && value <= 1, "lean_early: -1, 0 or 1");
        c->opt_lean_early = value;
        return RTR_OK;
    }
    if (!strcmp(key, "lane_test")) {  // T1: one point per lane first (k_project_bin); 0 = every point, as in round 3
        c->opt_lane_test = value != 0;
        c->list_valid = false;
        return RTR_OK;
    }
    if (!strcmp(key, "pack")) {  // applies to the resident cloud at once, and to every later one
        NEED(c, value >= 0 && value <= 2, "pack must be 0 (never), 1 (when it pays) or 2 (always, verified)");
        c->opt_pack = value;
        DevGuard g(c->device);
        HIP_TRY(c, sync_streams(c));
        if (c->n == 0 || c->cap == 0) return RTR_OK;
        if (int rc = ensure_soa(c)) return rc;  // (packing reads the fp32 arrays; "pack" = 0 leaves them as the resident form)
        if (int rc = pack_cloud(c)) return rc;
        drop_soa(c);
        return RTR_OK;
    }
    if (!strcmp(key, "keep_soa")) {  // 1: the fp32 SoA arrays stay resident beside the packed form (12 B per point)
        c->opt_keep_soa = value != 0;
        DevGuard g(c->device);
        if (c->opt_keep_soa) return (c->cap && c->n) ? ensure_soa(c) : RTR_OK;
        drop_soa(c);
        return RTR_OK;
    }
    if (!strcmp(key, "pool_worst_case")) {  // 1: the extent pool is sized for the worst case (2 n entries) at once
        c->opt_pool_worst = value != 0;
        return RTR_OK;
    }
    if (!strcmp(key, "probe_variant")) {
        c->opt_probe = value;
        return RTR_OK;
    }
    if (!strcmp(key, "tail_cus")) {  // takes effect when "overlap" is switched on
        NEED(c, value >= 0 && value < 32, "tail_cus must be in 0..31 (CUs per XCD)");
        NEED(c, !c->opt_overlap, "set tail_cus before overlap");
        c->opt_tail_cus = value;
        return RTR_OK;
    }
    if (!strcmp(key, "overlap")) return set_overlap(c, value != 0);
    if (!strcmp(key, "keep_accum")) {
        c->opt_keep_accum = value != 0;
        return RTR_OK;
    }
    return fail(c, RTR_ERR_INVALID, "unknown option '%s'", key);
}

int rtr_get_option(rtr_ctx *c, const char *key, int *value) {
    if (!c) return RTR_ERR_INVALID;
    NEED(c, key != nullptr && value != nullptr, "key / value is NULL");
    if (!strcmp(key, "mode")) *value = c->opt_mode;
    else if (!strcmp(key, "auto_reorder")) *value = c->opt_auto_reorder;
    else if (!strcmp(key, "reordered")) *value = c->reordered ? 1 : 0;  // the resident cloud was sorted by the library
    else if (!strcmp(key, "order_ratio_ppm")) *value = (int)(c->order_ratio * 1e6f);  // chunk / cloud diagonal as uploaded
    else if (!strcmp(key, "cull")) *value = c->opt_cull;
    else if (!strcmp(key, "lane_test")) *value = c->opt_lane_test;
    else if (!strcmp(key, "lean")) *value = c->opt_lean;
    else if (!strcmp(key, "lean_identity")) *value = c->opt_lean_identity;
    else if (!strcmp(key, "lean_early")) *value = c->opt_lean_early;
    else if (!strcmp(key, "p2p_open")) *value = c->p2p.open ? 1 : 0;  // the peers' buffers are mapped (rtr_p2p_open)
    else if (!strcmp(key, "keep_soa")) *value = c->opt_keep_soa;
    else if (!strcmp(key, "pool_worst_case")) *value = c->opt_pool_worst;
    else if (!strcmp(key, "resident_millibytes_per_point")) {
        // device memory this context holds for the cloud and its frames, per point: coordinates (fp32 SoA and / or packed
        // form), colours, chunk boxes and lane spreads, tile stores and extent pools, frame buffers
        const uint64_t nchunks = ((c->cap / 4) + 63) / 64;
        uint64_t b = (c->x ? 12 * c->cap : 0) + (c->rgba ? 4 * c->cap : 0) + nchunks * 28 + (c->pk_hdr ? c->pk_bytes + 64 : 0);
        for (const auto &f : c->fs) {
            b += f.dyn ? f.dyn_cap * 8 : 0;
            b += f.store.ext0 ? ((uint64_t)f.nst * rtr::kS0 + 16) * 8 + rtr::ts_meta_words(f.nst, f.ntiles) * 4 : 0;
        }
        const uint64_t npix = (uint64_t)c->W * c->H;
        b += c->depth ? npix * (4 + 16 + 3 + 1 + 10) : 0;
        *value = c->n ? (int)((b * 1000) / c->n > 0x7FFFFFFFull ? 0x7FFFFFFF : (b * 1000) / c->n) : 0;
    }
    else if (!strcmp(key, "pack")) *value = c->opt_pack;
    else if (!strcmp(key, "packed")) *value = c->pk_hdr ? 1 : 0;  // the point kernel reads the packed coordinates
    else if (!strcmp(key, "packed_millibytes_per_point"))         // its coordinate stream, headers included (12000 = raw)
        *value = c->pk_hdr && c->n ? (int)(c->pk_bytes * 1000 / c->n) : 12000;
    else if (!strcmp(key, "keep_accum")) *value = c->opt_keep_accum;
    else if (!strcmp(key, "split_threshold")) *value = c->opt_heavy;
    else if (!strcmp(key, "split_slice")) *value = c->opt_slice;
    else if (!strcmp(key, "point_grid")) *value = c->opt_grid;
    else return fail(c, RTR_ERR_INVALID, "unknown option '%s'", key);
    return RTR_OK;
}

int rtr_stream_probe(rtr_ctx *c, const float P[16]) {
    if (!c) return RTR_ERR_INVALID;
    NEED(c, P != nullptr, "P is NULL");
    if (int rc = check_frame(c)) return rc;
    DevGuard g(c->device);
    if (int rc = ensure_soa(c)) return rc;
    { Timed t(c, RTR_K_PROBE); rtr::launch_stream_probe(c->stream, cloud_of(c), make_proj(P), c->W, c->H, c->minmax, c->opt_probe); }
    return launch_check(c, "stream_probe");
}

int rtr_get_params(const rtr_ctx *c, rtr_params *p) {
    if (!c || !p) return RTR_ERR_INVALID;
    *p = c->prm;
    return RTR_OK;
}

static int switch_stream(rtr_ctx *c, hipStream_t s) {
    DevGuard g(c->device);
    HIP_TRY(c, sync_streams(c));
    (void)collect_timing(c);
    c->stream = s;
    return RTR_OK;
}

int rtr_set_stream(rtr_ctx *c, void *s) {  // NULL is HIP's default stream, a valid choice
    if (!c) return RTR_ERR_INVALID;
    return switch_stream(c, reinterpret_cast<hipStream_t>(s));
}

int rtr_reset_stream(rtr_ctx *c) {
    if (!c) return RTR_ERR_INVALID;
    return switch_stream(c, c->masked_tail ? c->masked_tail : c->own_stream);
}

// After a synchronisation: tile-store errors.  When the adaptive extent pool overflowed and the last frame was a whole
// one (rtr_render and what is built on it), that frame is rendered again with the grown pool -- the device buffers hold
// the right frame when the call returns.
static int finish_sync_rerender(rtr_ctx *c) {
    float P[16];
    memcpy(P, c->last_P, sizeof P);
    return rtr_render(c, P, c->last_filter);
}
static int finish_sync(rtr_ctx *c) {
    bool retry = false;
    int rc = check_store_error(c, c->last_valid ? &retry : nullptr);
    if (rc || !retry) return rc;
    if ((rc = finish_sync_rerender(c))) return rc;
    HIP_TRY(c, sync_streams(c));
    return check_store_error(c);
}

int rtr_synchronize(rtr_ctx *c) {
    if (!c) return RTR_ERR_INVALID;
    DevGuard g(c->device);
    HIP_TRY(c, sync_streams(c));
    return finish_sync(c);
}

// ---- cloud -------------------------------------------------------------------------

// Option "auto_reorder" (after every upload / generation; the chunk bounds are current).  2 (default):
// sort when the 256-point chunks are not spatially compact -- mean chunk diagonal more than twice what
// an ideally ordered VOLUME cloud of this size would have, (256 / n)^(1/3) of the cloud's diagonal.
// A scanner's sweep order or the reference loader's Morton-like surfaces pass; a hash-ordered cloud
// (ratio ~1) and the reference's 0.25 m blocks that are unordered inside do not.  Frames never depend
// on the point order; rtr_download_points returns the resident (possibly sorted) order.
// Best effort: the sort works on scratch copies and only writes the cloud back at the very end, so a
// cloud too large for the scratch simply stays in the order it was uploaded in.
// Option "pack" (after every upload / generation / reorder): the tile-binned point kernel reads the
// coordinates from the lossless PackedXyz form when that is at least 1/8 smaller than the 12 B/pt SoA
// stream (spatially ordered clouds: 6-9 B/pt; a hash-ordered one stays raw).  The SoA arrays stay resident
// -- the atomic form, the phase calls with another matrix, rtr_download_points and the sort use them.
// Best effort: without memory for it the cloud simply stays unpacked.
static int pack_cloud(rtr_ctx *c) {
    free_pack(c);
    if (c->opt_pack == 0 || c->n == 0) return RTR_OK;
    const uint64_t n4 = (c->n + 3) / 4, nchunks = (n4 + 63) / 64;
    struct Scratch {  // freed on every exit path
        void *p = nullptr;
        ~Scratch() { if (p) (void)hipFree(p); }
    } cnt, tot;
    uint4 *hdr = nullptr;
    uint32_t *planes = nullptr;
    auto give_up = [&]() {
        (void)hipGetLastError();
        if (hdr) (void)hipFree(hdr);
        if (planes) (void)hipFree(planes);
        return RTR_OK;
    };
    // (+ one zero header: the point kernel reads headers in pairs)
    if (hipMalloc((void **)&hdr, (nchunks + 1) * 2 * sizeof(uint4)) != hipSuccess) return give_up();
    if (hipMemsetAsync(hdr + nchunks * 2, 0, 2 * sizeof(uint4), c->stream) != hipSuccess) return give_up();
    if (hipMalloc(&cnt.p, nchunks * sizeof(uint32_t)) != hipSuccess) return give_up();
    if (hipMalloc(&tot.p, 2 * sizeof(uint64_t)) != hipSuccess) return give_up();
    if (hipMemsetAsync(tot.p, 0, 2 * sizeof(uint64_t), c->stream) != hipSuccess) return give_up();
    const rtr::Cloud cl = cloud_of(c);
    rtr::pack_measure(c->stream, cl, hdr, (uint32_t *)cnt.p, (uint64_t *)tot.p);
    uint64_t host[2] = {0, 0};
    if (hipMemcpyAsync(host, tot.p, sizeof host, hipMemcpyDeviceToHost, c->stream) != hipSuccess) return give_up();
    if (hipStreamSynchronize(c->stream) != hipSuccess) return give_up();
    const uint64_t bytes = host[0] * 32 + nchunks * 32;  // blocks (32-byte units) + headers
    if (c->opt_pack == 1 && bytes * 8 > n4 * 48 * 7) return give_up();  // saves less than 1/8 of the 12 B/pt stream
    // (one allocation: the A streams, 64 spare bytes, the B streams, 64 spare bytes -- the last lanes' loads run up to
    // 12 bytes past the last value of their stream)
    const uint64_t b_dw = rtr::pack_b_dwords(host[0]);
    if (hipMalloc((void **)&planes, rtr::pack_total_dwords(host[0]) * 4) != hipSuccess) return give_up();
    if (hipMemsetAsync(planes + host[0] * 2, 0, (b_dw - host[0] * 2) * 4, c->stream) != hipSuccess) return give_up();
    if (hipMemsetAsync(planes + b_dw + host[0] * 6, 0, 64, c->stream) != hipSuccess) return give_up();
    rtr::pack_write(c->stream, cl, hdr, planes, planes + b_dw);
    if (c->opt_pack == 2) {
        rtr::pack_verify(c->stream, cl, hdr, planes, planes + b_dw, (uint64_t *)tot.p + 1);
        if (hipMemcpyAsync(host, tot.p, sizeof host, hipMemcpyDeviceToHost, c->stream) != hipSuccess) return give_up();
    }
    if (hipStreamSynchronize(c->stream) != hipSuccess) return give_up();
    if (c->opt_pack == 2 && host[1] != 0) {
        (void)give_up();
        return fail(c, RTR_ERR_HIP, "pack: %llu points decode to other coordinates", (unsigned long long)host[1]);
    }
    c->pk_hdr = hdr;
    c->pk_planes = planes;
    c->pk_planes_b = planes + b_dw;
    c->pk_bytes = bytes;
    return RTR_OK;
}

static int auto_reorder(rtr_ctx *c) {
    c->reordered = false;
    c->order_ratio = 0.f;
    c->absmax[0] = c->absmax[1] = c->absmax[2] = __builtin_inff();  // (unknown: the lane test's margin step stays off)
    if (c->n < 1) return RTR_OK;
    float ratio = 0.f;  // measured under every policy: the point kernel has a form for incoherent clouds
    if (rtr::order_quality(c->stream, c->bounds, c->n, &ratio, c->absmax) != 0) {
        (void)hipGetLastError();
        c->absmax[0] = c->absmax[1] = c->absmax[2] = __builtin_inff();
        return RTR_OK;
    }
    bool want = c->opt_auto_reorder == 1;
    if (c->n >= (1u << 16)) {  // (tiny clouds render in microseconds whatever their order)
        c->order_ratio = ratio;
        if (c->opt_auto_reorder == 2) want = ratio > 2.0f * cbrtf(256.0f / (float)c->n);
    }
    if (c->n < 2 || !want) return RTR_OK;
    if (rtr_reorder_points(c) != RTR_OK) (void)hipGetLastError();
    return RTR_OK;
}

int rtr_upload_points(rtr_ctx *c, const float *xyz, size_t xs, const uint8_t *rgb, size_t rs, size_t n) {
    if (!c) return RTR_ERR_INVALID;
    NEED(c, n == 0 || (xyz && rgb), "xyz / rgb is NULL");
    NEED(c, xs >= 12 && xs % 4 == 0, "xyz_stride_bytes must be >= 12 and a multiple of 4");
    NEED(c, rs >= 3, "rgb_stride_bytes must be >= 3");
    NEED(c, n < (1ull << 32), "too many points for one context (point indices are 32-bit): shard the cloud");
    DevGuard g(c->device);
    HIP_TRY(c, sync_streams(c));
    int rc = alloc_cloud(c, n);
    if (rc) return rc;
    // stage through device chunks; AoS -> SoA on the GPU
    const uint64_t chunk = 1ull << 24;
    struct Staging {  // freed on every exit path
        uint8_t *p = nullptr;
        ~Staging() { if (p) (void)hipFree(p); }
    } stx, stc;
    uint64_t m = n < chunk ? n : chunk;
    if (m) {
        HIP_TRY(c, hipMalloc((void **)&stx.p, m * xs));
        HIP_TRY(c, hipMalloc((void **)&stc.p, m * rs));
    }
    uint8_t *sx = stx.p, *sc = stc.p;
    for (uint64_t off = 0; off < n; off += chunk) {
        uint64_t cnt = (n - off) < chunk ? (n - off) : chunk;
        HIP_TRY(c, hipMemcpyAsync(sx, (const uint8_t *)xyz + off * xs, cnt * xs, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(c, hipMemcpyAsync(sc, rgb + off * rs, cnt * rs, hipMemcpyHostToDevice, c->stream));
        rtr::launch_aos_to_soa(c->stream, sx, xs, sc, rs, cnt, c->x + off, c->y + off, c->z + off, c->rgba + off);
        HIP_TRY(c, sync_streams(c));
    }
    rtr::launch_pad_nan(c->stream, c->x, c->y, c->z, c->rgba, n, (n + 3) & ~3ull);
    rtr::launch_chunk_bounds(c->stream, cloud_of(c), c->bounds, c->spread);
    HIP_TRY(c, sync_streams(c));
    if (int rc2 = launch_check(c, "aos_to_soa")) return rc2;
    free_pack(c);
    if (int rc2 = auto_reorder(c)) return rc2;
    if (!c->pk_hdr)  // (a sort has packed already)
        if (int rc2 = pack_cloud(c)) return rc2;
    drop_soa(c);
    return RTR_OK;
}

int rtr_generate_synthetic(rtr_ctx *c, int scene, uint64_t seed, uint64_t first, uint64_t count, uint64_t total) {
    if (!c) return RTR_ERR_INVALID;
    NEED(c, scene == RTR_SCENE_UNIFORM_BOX || scene == RTR_SCENE_ROOM_SHELL, "unknown scene");
    NEED(c, first + count <= total, "first + count exceeds total");
    NEED(c, total < (1ull << 33), "total too large");
    NEED(c, count < (1ull << 32), "too many points for one context (point indices are 32-bit): shard the cloud");
    DevGuard g(c->device);
    HIP_TRY(c, sync_streams(c));
    int rc = alloc_cloud(c, count);
    if (rc) return rc;
    rtr::launch_generate(c->stream, scene, seed, first, count, total, c->x, c->y, c->z, c->rgba);
    rtr::launch_pad_nan(c->stream, c->x, c->y, c->z, c->rgba, count, (count + 3) & ~3ull);
    rtr::launch_chunk_bounds(c->stream, cloud_of(c), c->bounds, c->spread);
    HIP_TRY(c, sync_streams(c));
    if (int rc2 = launch_check(c, "generate")) return rc2;
    free_pack(c);
    if (int rc2 = auto_reorder(c)) return rc2;
    if (!c->pk_hdr)
        if (int rc2 = pack_cloud(c)) return rc2;
    drop_soa(c);
    return RTR_OK;
}

int rtr_reorder_points(rtr_ctx *c) {
    if (!c) return RTR_ERR_INVALID;
    DevGuard g(c->device);
    HIP_TRY(c, sync_streams(c));
    c->list_valid = false;
    if (int rc = ensure_soa(c)) return rc;
    int e = rtr::reorder_morton(c->stream, c->x, c->y, c->z, c->rgba, c->n);
    if (e != 0) return fail(c, RTR_ERR_HIP, "reorder failed: %s", hipGetErrorString((hipError_t)e));
    c->reordered = true;
    free_pack(c);
    rtr::launch_chunk_bounds(c->stream, cloud_of(c), c->bounds, c->spread);
    HIP_TRY(c, sync_streams(c));
    if (int rc = launch_check(c, "reorder")) return rc;
    if (int rc = pack_cloud(c)) return rc;
    drop_soa(c);
    return RTR_OK;
}

int rtr_num_points(const rtr_ctx *c, uint64_t *n) {
    if (!c || !n) return RTR_ERR_INVALID;
    *n = c->n;
    return RTR_OK;
}

int rtr_download_points(rtr_ctx *c, float *xyzw, uint8_t *rgba, uint64_t first, uint64_t count) {
    if (!c) return RTR_ERR_INVALID;
    NEED(c, xyzw && rgba, "output is NULL");
    NEED(c, first + count <= c->n, "range exceeds the resident cloud");
    if (count == 0) return RTR_OK;
    DevGuard g(c->device);
    if (int rc = ensure_soa(c)) return rc;  // (a cloud resident in packed form only is decoded for the copy, bit for bit)
    const uint64_t chunk = 1ull << 24;
    uint64_t m = count < chunk ? count : chunk;
    struct Staging {  // freed on every exit path
        void *p = nullptr;
        ~Staging() { if (p) (void)hipFree(p); }
    } stx, stc;
    HIP_TRY(c, hipMalloc(&stx.p, m * 16));
    HIP_TRY(c, hipMalloc(&stc.p, m * 4));
    float *dx = static_cast<float *>(stx.p);
    uint8_t *dc = static_cast<uint8_t *>(stc.p);
    for (uint64_t off = 0; off < count; off += chunk) {
        uint64_t cnt = (count - off) < chunk ? (count - off) : chunk;
        uint64_t s0 = first + off;
        rtr::launch_soa_to_aos(c->stream, c->x + s0, c->y + s0, c->z + s0, c->rgba + s0, cnt, dx, dc);
        HIP_TRY(c, hipMemcpyAsync(xyzw + off * 4, dx, cnt * 16, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipMemcpyAsync(rgba + off * 4, dc, cnt * 4, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, sync_streams(c));
    }
    drop_soa(c);
    return launch_check(c, "soa_to_aos");
}

// ---- camera ------------------------------------------------------------------------

// project_cloud.cu:318 with project_cloud.h:50-59 and CameraCalibration.cpp:17-27:
// both glm transposes cancel, leaving P = K4 * E evaluated in fp32 (each entry the
// left-to-right sum of four separately rounded products), stored row-major.
int rtr_compose_projection(const double K[9], const double E[16], float P[16]) {
    if (!K || !E || !P) return RTR_ERR_INVALID;
    float K4[16] = {0}, Ef[16];
    for (int r = 0; r < 3; ++r)
        for (int q = 0; q < 3; ++q) K4[4 * r + q] = static_cast<float>(K[3 * r + q]);
    K4[15] = 1.0f;
    for (int i = 0; i < 16; ++i) Ef[i] = static_cast<float>(E[i]);
    for (int r = 0; r < 4; ++r)
        for (int q = 0; q < 4; ++q) {
            volatile float s = K4[4 * r + 0] * Ef[q];  // volatile: one rounding per op, no contraction
            volatile float t = K4[4 * r + 1] * Ef[4 + q];
            s = s + t;
            t = K4[4 * r + 2] * Ef[8 + q];
            s = s + t;
            t = K4[4 * r + 3] * Ef[12 + q];
            s = s + t;
            P[4 * r + q] = s;
        }
    return RTR_OK;
}

int rtr_set_resolution(rtr_ctx *c, int W, int H) {
    if (!c) return RTR_ERR_INVALID;
    NEED(c, W > 0 && H > 0 && (int64_t)W * H < (1ll << 31), "bad resolution");
    if (W == c->W && H == c->H) return RTR_OK;
    DevGuard g(c->device);
    HIP_TRY(c, sync_streams(c));
    free_frame(c);
    size_t npix = (size_t)W * H;
    HIP_TRY(c, hipMalloc((void **)&c->depth, npix * 4));
    HIP_TRY(c, hipMalloc((void **)&c->acc, npix * 16));
    HIP_TRY(c, hipMalloc((void **)&c->img, (npix * 3 + 15) & ~(size_t)15));
    HIP_TRY(c, hipMalloc((void **)&c->mask, npix));
    size_t nparts = (size_t)((W + 31) / 32) * ((H + 31) / 32);
    HIP_TRY(c, hipMalloc((void **)&c->part_min, nparts * 4));
    HIP_TRY(c, hipMalloc((void **)&c->part_max, nparts * 4));
    HIP_TRY(c, hipMalloc((void **)&c->tensor, npix * 5 * sizeof(uint16_t)));
    c->W = W; c->H = H;
    return RTR_OK;
}

// ---- phases ------------------------------------------------------------------------

int rtr_clear(rtr_ctx *c) {
    if (!c) return RTR_ERR_INVALID;
    if (int rc = check_frame(c)) return rc;
    DevGuard g(c->device);
    { Timed t(c, RTR_K_CLEAR); rtr::launch_clear(c->stream, c->depth, c->acc, (size_t)c->W * c->H); }
    return launch_check(c, "clear");
}

// The binned form keeps one LDS counter per screen tile (<= 4096 tiles: up to 3840 x 2160);
// larger frames fall back to the atomic form.
static bool use_tiles(const rtr_ctx *c) {
    return c->opt_mode == 1 && !c->force_atomic && rtr::tile_count(c->W, c->H) <= 4096;
}

// T1 of the tile-binned form: stream the cloud, append the in-frustum points to the tile store; its
// last workgroup writes the tile kernel's work list (and resets the tiles that will be split when
// the tile launches are the frame buffers' only writers: `clear_split`).
// With `overlapped` T1 goes to the front stream and fills the set the tail is NOT reading, so it
// runs beside T4 / the prefilter of the previous frame.
static int bin_points(rtr_ctx *c, const float P[16], bool overlapped, bool clear_split, bool no_split = false,
                      bool lean = false) {
    c->list_valid = false;
    c->last_lean = false;
    c->p2p.occ_current = false;
    c->p2p.occ_from_scan = false;
    hipStream_t s1 = c->stream;
    if (overlapped) {
        c->cur ^= 1;
        s1 = c->front;
        if (c->F().consumed_valid) HIP_TRY(c, hipStreamWaitEvent(c->front, c->F().consumed, 0));
    }
    if (int rc = ensure_lists(c)) return rc;
    if (int rc = ensure_tiles(c, s1)) return rc;
    auto &t = c->F().store;
    // (automatic: 8 bytes apart -- fewer cache lines for T1's epilogue to read and reset: -2 us on C3, -2.5 us on C2 --
    // unless tiles above the split threshold have been seen lately, where the claims of all waves queue on a dozen
    // counters and those want lines of their own; any spacing can follow any other, the counters are zero between frames)
    const bool heavy_seen = c->split_cooldown > 0 || __atomic_load_n(c->split_host, __ATOMIC_RELAXED) != 0u;
    t.fill_shift = c->opt_fill_shift >= 0 ? c->opt_fill_shift : (heavy_seen ? 4 : 1);
    t.seq = (t.seq + 1u) & 0xFFFFFFu;
    if (t.seq == 0u) {  // the 24-bit stamp wrapped: forget every directory entry once
        HIP_TRY(c, hipMemsetAsync(rtr::ts_dir(t), 0, (size_t)c->F().nst * rtr::kDirK * sizeof(unsigned long long), s1));
        t.seq = 1u;
    }
    {
        Timed tm(c, RTR_K_MIN_DEPTH, s1, true);
        rtr::launch_project_bin(s1, cloud_of(c), make_proj(P), c->W, c->H, t, c->opt_cull ? c->bounds : nullptr,
                                (clear_split ? 1 : 0) | (no_split ? 2 : 0) | (c->opt_lane_test ? 0 : 4) | (lean ? 8 : 0),
                                c->opt_phases, c->opt_xp, tm.a, tm.b);
        c->p2p.occ_from_scan = c->p2p.open;
    }
    if (overlapped) {
        HIP_TRY(c, hipEventRecord(c->F().binned, c->front));
        HIP_TRY(c, hipStreamWaitEvent(c->stream, c->F().binned, 0));
    }
    memcpy(c->list_P, P, sizeof c->list_P);
    c->list_valid = true;
    return RTR_OK;
}

int rtr_min_depth_pass(rtr_ctx *c, const float P[16]) {
    if (!c) return RTR_ERR_INVALID;
    NEED(c, P != nullptr, "P is NULL");
    if (int rc = check_frame(c)) return rc;
    DevGuard g(c->device);
    c->list_valid = false;
    c->last_valid = false;
    if (use_tiles(c)) {
        if (int rc = bin_points(c, P, false, c->p2p.whole_frame)) return rc;
        Timed t(c, RTR_K_TILE);
        rtr::launch_tile(c->stream, 1, c->W, c->H, c->F().store, c->prm.depth_window, c->depth, c->acc, c->img,
                         (c->p2p.whole_frame ? 6 : 0) | (c->lean_parity << 4), nullptr);  // 2: only writer, 4: tiles without entries are not written
        mark_consumed(c);
    } else {
        if (int rc = ensure_soa(c)) return rc;
        Timed t(c, RTR_K_MIN_DEPTH);
        rtr::launch_min_depth(c->stream, cloud_of(c), make_proj(P), c->W, c->H, c->depth);
    }
    return launch_check(c, "min_depth_pass");
}

int rtr_accumulate_pass(rtr_ctx *c, const float P[16]) {
    if (!c) return RTR_ERR_INVALID;
    NEED(c, P != nullptr, "P is NULL");
    if (int rc = check_frame(c)) return rc;
    DevGuard g(c->device);
    // the bins are only usable for the matrix they were built with; otherwise re-project
    // the cloud like the reference does (render.cu:90-98)
    const bool use_bins = use_tiles(c) && c->list_valid && memcmp(c->list_P, P, sizeof c->list_P) == 0;
    c->p2p.acc_from_bins = use_bins && c->p2p.open && c->p2p.occ_current;
    if (c->p2p.whole_frame && !use_bins) return fail(c, RTR_ERR_INVALID, "rtr_p2p_render: the bins are not valid");
    if (use_bins) {
        // inside rtr_p2p_render with the default pyramid depth the tile kernel also emits the prefilter's
        // levels and min / max partials from the GLOBAL depth tile it has just loaded
        rtr::TilePyr pyr{};
        pyr.enable = (c->p2p.whole_frame && c->p2p.pyramid_done) ? 1 : 0;
        if (pyr.enable) {
            pyr.L = c->lv;
            pyr.n_eff_rows = (uint32_t)((c->H >> 4) << 4);
            pyr.part_min = c->part_min;
            pyr.part_max = c->part_max;
        }
        rtr::Sliced dsl{};
        if (c->p2p.whole_frame && c->p2p.depth_peers) {  // MIN over the occupying ranks' local depth, tile by tile
            dsl.src = c->p2p.depth;
            dsl.peers = c->p2p.world;
            dsl.occ_all = c->p2p.occ_all;
            dsl.out = c->p2p.red;  // (nobody reads `red` in this form; RTR_BUF_DEPTH is completed from it below)
            c->p2p.depth_in_red = true;
        } else if (c->p2p.whole_frame && c->p2p.depth_sliced) {
            dsl.src = c->p2p.reduced;
            dsl.chunk = p2p_slice(c).chunk;
        }
        c->p2p.depth_sliced = c->p2p.depth_peers = false;
        Timed t(c, RTR_K_TILE);
        rtr::launch_tile(c->stream, 2, c->W, c->H, c->F().store, c->prm.depth_window, c->depth, c->acc, c->img,
                         c->p2p.whole_frame ? 6 : 0, pyr.enable ? &pyr : nullptr, (dsl.chunk || dsl.peers) ? &dsl : nullptr);
        mark_consumed(c);
    } else {
        if (int rc = ensure_soa(c)) return rc;
        Timed t(c, RTR_K_ACCUMULATE);
        rtr::launch_accumulate(c->stream, cloud_of(c), make_proj(P), c->W, c->H, c->depth, c->acc, c->prm.depth_window);
    }
    return launch_check(c, "accumulate_pass");
}

int rtr_resolve(rtr_ctx *c) {
    if (!c) return RTR_ERR_INVALID;
    if (int rc = check_frame(c)) return rc;
    DevGuard g(c->device);
    { Timed t(c, RTR_K_RESOLVE); rtr::launch_resolve(c->stream, c->acc, c->img, (size_t)c->W * c->H); }
    return launch_check(c, "resolve");
}

int rtr_resolve_range(rtr_ctx *c, const void *acc_dev, uint64_t first_pixel, uint64_t count) {
    if (!c) return RTR_ERR_INVALID;
    if (int rc = check_frame(c)) return rc;
    const uint64_t npix = (uint64_t)c->W * c->H;
    NEED(c, first_pixel % 4 == 0 && first_pixel + count <= npix, "bad pixel range (first must be a multiple of 4)");
    if (count == 0) return RTR_OK;
    DevGuard g(c->device);
    const uint32_t *src = acc_dev ? static_cast<const uint32_t *>(acc_dev) : c->acc + first_pixel * 4;
    { Timed t(c, RTR_K_RESOLVE); rtr::launch_resolve(c->stream, src, c->img + first_pixel * 3, (size_t)count); }
    return launch_check(c, "resolve_range");
}

static int filter_impl(rtr_ctx *c, int pyramid_parts, const rtr::Sliced *img_slices = nullptr,
                       const uint32_t *depth_src = nullptr) {
    if (int rc = check_frame(c)) return rc;
    DevGuard g(c->device);
    if (int rc = ensure_pyramid(c)) return rc;
    {
        Timed t(c, RTR_K_FILTER);
        rtr::launch_filter(c->stream, c->lv, c->depth, c->img, c->mask, c->tensor, c->minmax, c->part_min, c->part_max,
                           c->W, c->H, c->prm.filter_strength, c->prm.gradient_threshold, pyramid_parts, img_slices,
                           depth_src);
    }
    return launch_check(c, "filter");
}

int rtr_filter(rtr_ctx *c) {
    if (!c) return RTR_ERR_INVALID;
    return filter_impl(c, 0);
}

// ---- whole frames ------------------------------------------------------------------

int rtr_render(rtr_ctx *c, const float P[16], int with_filter) {
    if (!c) return RTR_ERR_INVALID;
    NEED(c, P != nullptr, "P is NULL");
    if (int rc = check_frame(c)) return rc;
    if (with_filter) {  // fail before touching the frame buffers
        DevGuard g(c->device);
        if (int rc = ensure_pyramid(c)) return rc;
    }
    int rc;
    if (use_tiles(c)) {  // one launch does clear + min + accumulate + resolve per tile
        DevGuard g(c->device);
        // (T1 beside the previous frame's tail must not touch the frame buffers: the split tiles' pixels are then
        // reset by a launch of their own on the tail's stream)
        const bool overlapped = c->opt_overlap && c->front;
        bool split_launch = c->opt_heavy > 0;  // (split_threshold 0: nothing is ever split)
        if (split_launch) {
            if (__atomic_load_n(c->split_host, __ATOMIC_RELAXED) != 0u) c->split_cooldown = kSplitCooldown;
            split_launch = c->split_cooldown > 0;
            if (c->split_cooldown > 0) --c->split_cooldown;
        }
        // LEAN frames (rtr_kernels.h, ts_off_order): no split launch pending, one stream, no peers -- T1 ends without
        // ticket and epilogue, the tile workgroups read and reset their stream counters themselves
        const bool lean = c->opt_lean && !overlapped && !split_launch && !c->p2p.open;
        if ((rc = bin_points(c, P, overlapped, !overlapped, !split_launch, lean))) return rc;
        if (lean) {
            c->lean_parity ^= 1;
            c->last_lean = true;
        }
        if (overlapped && split_launch) rtr::launch_reset_split(c->stream, c->W, c->H, c->F().store, c->depth, c->acc);
        // with the default four levels the tile kernel also emits the prefilter's pyramid and
        // min / max partials (F1) while the finished depth tile is still in LDS
        rtr::TilePyr pyr{};
        pyr.enable = (with_filter && c->prm.levels == 4) ? 1 : 0;
        if (pyr.enable) {
            pyr.L = c->lv;
            pyr.n_eff_rows = (uint32_t)((c->H >> 4) << 4);
            pyr.part_min = c->part_min;
            pyr.part_max = c->part_max;
        }
        {
            Timed t(c, RTR_K_TILE);
            // (lean frames, tile_body: bit 5 = no launch order when the launch is resident at once; bit 6 = the first batch
            // of entries before the counters, when the tiles are expected full: the last frame's entry count -- a mapped
            // word, no sync -- is at least half a batch, 1024 entries, per tile)
            int lean_bits = 0;
            if (lean) {
                lean_bits = 8 | (c->opt_lean_identity ? 32 : 0);
                const uint64_t e_last = c->entries_host ? __atomic_load_n(c->entries_host, __ATOMIC_RELAXED) : 0u;
                if (c->opt_lean_early > 0 || (c->opt_lean_early < 0 && e_last >= 1024ull * (uint64_t)rtr::tile_count(c->W, c->H)))
                    lean_bits |= 64;
            }
            rtr::launch_tile(c->stream, 0, c->W, c->H, c->F().store, c->prm.depth_window, c->depth, c->acc, c->img,
                             c->opt_keep_accum | lean_bits | (c->lean_parity << 4), pyr.enable ? &pyr : nullptr);
            if (lean) c->list_valid = false;  // (the tile launch has consumed and reset the stream counters)
            // tiles heavier than option "split_threshold" are split over several workgroups: a second launch takes
            // the minimum over each slice (they meet in the depth buffer), then -- behind a barrier over its 256
            // workgroups -- accumulates the slices against that minimum, and the last slice of each tile resolves it
            // (no work item on ordinary frames: its workgroups leave at once)
            if (split_launch)  // (skipped while no frame has had a tile above the threshold: see rtr_ctx::split_host)
                rtr::launch_tile(c->stream, 3, c->W, c->H, c->F().store, c->prm.depth_window, c->depth, c->acc, c->img,
                                 c->opt_keep_accum, pyr.enable ? &pyr : nullptr);
        }
        mark_consumed(c);
        if ((rc = launch_check(c, "tile frame"))) return rc;
        memcpy(c->last_P, P, sizeof c->last_P);  // (what a synchronising call repeats if the adaptive pool overflowed)
        c->last_filter = with_filter;
        c->last_valid = true;
        if (with_filter) return filter_impl(c, pyr.enable ? rtr::tile_count(c->W, c->H) : 0);
        return RTR_OK;
    } else {
        c->force_atomic = true;  // the phase calls below must not take the binned form either
        rc = rtr_clear(c);
        if (!rc) rc = rtr_min_depth_pass(c, P);
        if (!rc) rc = rtr_accumulate_pass(c, P);
        if (!rc) rc = rtr_resolve(c);
        c->force_atomic = false;
        if (rc) return rc;
    }
    if (with_filter && (rc = rtr_filter(c))) return rc;
    return RTR_OK;
}

static int frame_to_host(rtr_ctx *c, const float P[16], uint8_t *host_img, float *host_depth, int with_filter) {
    if (!c) return RTR_ERR_INVALID;
    if (!host_img && !host_depth) return fail(c, RTR_ERR_NO_OUTPUT, "both outputs are NULL (project_cloud.cu:270-273)");
    for (int attempt = 0;; ++attempt) {  // (a second time only after the adaptive extent pool overflowed: check_store_error)
        int rc = rtr_render(c, P, with_filter);
        if (rc) return rc;
        DevGuard g(c->device);
        size_t npix = (size_t)c->W * c->H;
        if (host_depth) HIP_TRY(c, hipMemcpyAsync(host_depth, c->depth, npix * 4, hipMemcpyDeviceToHost, c->stream));
        if (host_img) HIP_TRY(c, hipMemcpyAsync(host_img, c->img, npix * 3, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, sync_streams(c));
        bool retry = false;
        rc = check_store_error(c, attempt == 0 ? &retry : nullptr);
        if (rc || !retry) return rc;
    }
}

int rtr_project(rtr_ctx *c, const float P[16], uint8_t *host_img, float *host_depth) {
    return frame_to_host(c, P, host_img, host_depth, 0);
}

// ---- asynchronous host outputs ------------------------------------------------------------
// The reference's call shape pays kernels + 14.5 MB over PCIe per 1080p frame, one after the other
// (project_cloud.cu:302-309,424-431).  Here frame k's copies run on a second stream (the SDMA engines) beside
// frame k + 1's kernels: depth and image are snapshotted on the device first (two device copies, ~10 us), so the
// frame buffers are free again at once.

static int ensure_host_out(rtr_ctx *c) {
    if (!c->copy_stream) HIP_TRY(c, hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
    const size_t npix = (size_t)c->W * c->H;
    for (auto &h : c->ho) {
        if (h.img) continue;
        HIP_TRY(c, hipHostMalloc((void **)&h.img, (npix * 3 + 15) & ~(size_t)15, hipHostMallocMapped));
        HIP_TRY(c, hipHostMalloc((void **)&h.depth, (npix * 4 + 15) & ~(size_t)15, hipHostMallocMapped));
        HIP_TRY(c, hipHostGetDevicePointer(&h.img_map, h.img, 0));
        HIP_TRY(c, hipHostGetDevicePointer(&h.depth_map, h.depth, 0));
        HIP_TRY(c, hipMalloc((void **)&h.dimg, (npix * 3 + 15) & ~(size_t)15));
        HIP_TRY(c, hipMalloc((void **)&h.ddepth, (npix * 4 + 15) & ~(size_t)15));
        HIP_TRY(c, hipEventCreateWithFlags(&h.snap, hipEventDisableTiming));
        HIP_TRY(c, hipEventCreateWithFlags(&h.done, hipEventDisableTiming));
    }
    return RTR_OK;
}

int rtr_host_output_buffers(rtr_ctx *c, int slot, uint8_t **img, float **depth) {
    if (!c) return RTR_ERR_INVALID;
    NEED(c, slot >= 0 && slot < RTR_ASYNC_SLOTS, "slot out of range");
    if (int rc = check_frame(c)) return rc;
    DevGuard g(c->device);
    if (int rc = ensure_host_out(c)) return rc;
    if (img) *img = c->ho[slot].img;
    if (depth) *depth = c->ho[slot].depth;
    return RTR_OK;
}

int rtr_project_async(rtr_ctx *c, const float P[16], int slot, int with_filter) {
    if (!c) return RTR_ERR_INVALID;
    NEED(c, P != nullptr, "P is NULL");
    NEED(c, slot >= 0 && slot < RTR_ASYNC_SLOTS, "slot out of range");
    if (int rc = check_frame(c)) return rc;
    DevGuard g(c->device);
    if (int rc = ensure_host_out(c)) return rc;
    auto &h = c->ho[slot];
    // (the slot's previous frame may still be on its way to the host: its snapshot must not be overwritten yet)
    if (h.busy) HIP_TRY(c, hipStreamWaitEvent(c->stream, h.done, 0));
    if (int rc = rtr_render(c, P, with_filter)) return rc;
    const size_t npix = (size_t)c->W * c->H;
    HIP_TRY(c, hipMemcpyAsync(h.ddepth, c->depth, npix * 4, hipMemcpyDeviceToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(h.dimg, c->img, npix * 3, hipMemcpyDeviceToDevice, c->stream));
    HIP_TRY(c, hipEventRecord(h.snap, c->stream));
    HIP_TRY(c, hipStreamWaitEvent(c->copy_stream, h.snap, 0));
    rtr::launch_copy_to_host(c->copy_stream, h.ddepth, h.depth_map, npix * 4, h.dimg, h.img_map, npix * 3);
    HIP_TRY(c, hipEventRecord(h.done, c->copy_stream));
    h.busy = true;
    return RTR_OK;
}

int rtr_wait(rtr_ctx *c, int slot) {
    if (!c) return RTR_ERR_INVALID;
    NEED(c, slot >= -1 && slot < RTR_ASYNC_SLOTS, "slot out of range (-1: every slot)");
    DevGuard g(c->device);
    for (int k = 0; k < RTR_ASYNC_SLOTS; ++k) {
        if ((slot >= 0 && k != slot) || !c->ho[k].busy) continue;
        HIP_TRY(c, hipEventSynchronize(c->ho[k].done));
        c->ho[k].busy = false;
    }
    return check_store_error(c);
}

int rtr_project_filtered(rtr_ctx *c, const float P[16], uint8_t *host_img, float *host_depth) {
    return frame_to_host(c, P, host_img, host_depth, 1);
}

// ---- peer-to-peer exchange ------------------------------------------------------------

static int p2p_alloc(rtr_ctx *c) {  // this rank's exchange buffers (per resolution)
    auto &q = c->p2p;
    const size_t npix = (size_t)c->W * c->H;
    if (!q.red) HIP_TRY(c, hipMalloc((void **)&q.red, ((npix + 3) & ~(size_t)3) * sizeof(uint32_t)));
    if (!q.ximg) HIP_TRY(c, hipMalloc((void **)&q.ximg, (npix * 3 + 15) & ~(size_t)15));
    if (!q.occ) HIP_TRY(c, hipMalloc((void **)&q.occ, rtr::kP2POccBytes));
    if (!q.occ_all) HIP_TRY(c, hipMalloc((void **)&q.occ_all, (size_t)rtr::kMaxPeers * rtr::kP2POccBytes));
    if (!q.flags) {
        HIP_TRY(c, hipExtMallocWithFlags((void **)&q.flags, 4096, hipDeviceMallocUncached));
        HIP_TRY(c, hipMemsetAsync(q.flags, 0, 4096, c->stream));
        HIP_TRY(c, sync_streams(c));
    }
    if (!q.status_host) {
        HIP_TRY(c, hipHostMalloc((void **)&q.status_host, sizeof(uint32_t), hipHostMallocMapped));
        *q.status_host = 0;
        void *d = nullptr;
        HIP_TRY(c, hipHostGetDevicePointer(&d, q.status_host, 0));
        q.status_dev = static_cast<uint32_t *>(d);
    }
    return RTR_OK;
}

int rtr_p2p_export(rtr_ctx *c, rtr_p2p_handles *mine) {
    if (!c) return RTR_ERR_INVALID;
    NEED(c, mine != nullptr, "handles is NULL");
    if (int rc = check_frame(c)) return rc;
    static_assert(sizeof(hipIpcMemHandle_t) <= 64, "handle block too small");
    NEED(c, !c->opt_overlap, "the peer-to-peer exchange needs option overlap off (the peers map ONE tile store)");
    DevGuard g(c->device);
    HIP_TRY(c, sync_streams(c));
    // (the owner-computes form reads the peers' tile stores: allocate this rank's now -- it is sized by the cloud, for
    // the worst case: a pool the peers have mapped must never move)
    c->cur = 0;
    c->pool_worst = true;
    if (int rc = ensure_lists(c)) return rc;
    if (int rc = ensure_tiles(c, c->stream)) return rc;
    if (int rc = p2p_alloc(c)) return rc;
    HIP_TRY(c, sync_streams(c));
    memset(mine, 0, sizeof *mine);
    void *bufs[9] = {c->depth, c->acc, c->p2p.ximg, c->p2p.red, c->p2p.flags, c->p2p.occ,
                     c->F().store.meta, c->F().store.ext0, c->F().dyn};
    unsigned char *dst[9] = {mine->depth, mine->accum, mine->image, mine->reduced, mine->flags, mine->tiles,
                             mine->store_meta, mine->store_ext0, mine->store_dyn};
    for (int k = 0; k < 9; ++k) {
        hipIpcMemHandle_t h;
        HIP_TRY(c, hipIpcGetMemHandle(&h, bufs[k]));
        memcpy(dst[k], &h, sizeof h);
    }
    return RTR_OK;
}

int rtr_p2p_open(rtr_ctx *c, int rank, int world, const rtr_p2p_handles *all) {
    if (!c) return RTR_ERR_INVALID;
    NEED(c, all != nullptr, "handles is NULL");
    NEED(c, world >= 1 && world <= RTR_P2P_MAX_RANKS && rank >= 0 && rank < world, "bad rank / world");
    if (int rc = check_frame(c)) return rc;
    DevGuard g(c->device);
    HIP_TRY(c, sync_streams(c));
    auto &q = c->p2p;
    NEED(c, q.red && q.ximg && q.flags && q.occ, "rtr_p2p_export has not been called for this resolution");
    NEED(c, !q.open, "already open (rtr_p2p_close first)");
    NEED(c, c->F().store.meta && c->F().store.ext0 && c->F().dyn, "the tile store changed since rtr_p2p_export (export again)");
    rtr::PeerSet *sets[9] = {&q.depth, &q.accum, &q.image, &q.reduced, &q.flags_of, &q.occ_of, &q.meta_of, &q.ext0_of, &q.dyn_of};
    void *own[9] = {c->depth, c->acc, q.ximg, q.red, q.flags, q.occ, c->F().store.meta, c->F().store.ext0, c->F().dyn};
    for (int r = 0; r < world; ++r) {
        const unsigned char *src[9] = {all[r].depth, all[r].accum, all[r].image, all[r].reduced, all[r].flags, all[r].tiles,
                                       all[r].store_meta, all[r].store_ext0, all[r].store_dyn};
        for (int k = 0; k < 9; ++k) {
            if (r == rank) {
                sets[k]->p[r] = own[k];
                continue;
            }
            hipIpcMemHandle_t h;
            memcpy(&h, src[k], sizeof h);
            void *ptr = nullptr;
            hipError_t e = hipIpcOpenMemHandle(&ptr, h, hipIpcMemLazyEnablePeerAccess);
            if (e != hipSuccess) {
                int rc = fail(c, RTR_ERR_HIP, "hipIpcOpenMemHandle (rank %d, buffer %d) failed: %s", r, k, hipGetErrorString(e));
                p2p_release(c);
                return rc;
            }
            q.opened[k][r] = ptr;
            sets[k]->p[r] = ptr;
        }
    }
    {   // the owner-computes form's pointer table, in device memory
        rtr::OwnedTab tab{};
        for (int r = 0; r < world; ++r) {
            tab.meta[r] = static_cast<const uint32_t *>(q.meta_of.p[r]);
            tab.ext0[r] = static_cast<const uint64_t *>(q.ext0_of.p[r]);
            tab.dyn[r] = static_cast<const uint64_t *>(q.dyn_of.p[r]);
            tab.depth[r] = static_cast<const uint32_t *>(q.depth.p[r]);
            tab.ximg[r] = static_cast<const uint8_t *>(q.image.p[r]);
        }
        if (!q.tab) HIP_TRY(c, hipMalloc((void **)&q.tab, sizeof tab));
        HIP_TRY(c, hipMemcpy(q.tab, &tab, sizeof tab, hipMemcpyHostToDevice));
    }
    q.rank = rank;
    q.world = world;
    q.seq = 0;
    q.open = true;
    *q.status_host = 0;
    return RTR_OK;
}

int rtr_p2p_close(rtr_ctx *c) {
    if (!c) return RTR_ERR_INVALID;
    DevGuard g(c->device);
    HIP_TRY(c, sync_streams(c));
    p2p_release(c);
    return RTR_OK;
}

int rtr_p2p_status(rtr_ctx *c, uint32_t *barrier_timeouts) {
    if (!c || !barrier_timeouts) return RTR_ERR_INVALID;
    *barrier_timeouts = c->p2p.status_host ? *c->p2p.status_host : 0u;
    return RTR_OK;
}

namespace {
void p2p_barrier(rtr_ctx *c) {
    auto &q = c->p2p;
    const unsigned long long ticks = 100000ull * (unsigned long long)c->opt_p2p_timeout_ms;  // 100 MHz wall clock
    rtr::launch_p2p_sync(c->stream, q.flags, q.flags_of, q.rank, q.world, ++q.seq, q.status_dev, ticks);
}
}  // namespace

int rtr_p2p_min_depth(rtr_ctx *c) {
    if (!c) return RTR_ERR_INVALID;
    if (int rc = check_frame(c)) return rc;
    NEED(c, c->p2p.open, "rtr_p2p_open has not been called");
    DevGuard g(c->device);
    auto &q = c->p2p;
    const Slice s = p2p_slice(c);
    const size_t npix = (size_t)c->W * c->H;
    // which screen tiles this rank's frame touches at all (only known when it came from the bins)
    const bool binned = use_tiles(c) && c->list_valid;
    if (!(binned && q.occ_from_scan))  // otherwise the scan kernel of the tile sort has already written it
        rtr::launch_p2p_occupancy(c->stream, binned ? rtr::ts_tile_cnt(c->F().store) : nullptr, c->W, c->H, q.occ);
    q.occ_current = binned;
    q.acc_from_bins = false;
    p2p_barrier(c);  // every rank's local depth (and occupancy) is complete
    rtr::launch_p2p_depth_reduce(c->stream, q.depth, q.occ_of, q.red, s.first, s.count, q.world, c->W, c->H);
    p2p_barrier(c);  // every slice is reduced; nobody reads the local depth buffers any more
    if (q.whole_frame)
        q.depth_sliced = true;  // the accumulate launch reads the slices tile by tile and stores the result
    else
        rtr::launch_p2p_gather(c->stream, q.reduced, c->depth, s.chunk * 4, npix * 4, -1);
    return launch_check(c, "p2p_min_depth");
}

int rtr_p2p_sum_resolve(rtr_ctx *c) {
    if (!c) return RTR_ERR_INVALID;
    if (int rc = check_frame(c)) return rc;
    NEED(c, c->p2p.open, "rtr_p2p_open has not been called");
    DevGuard g(c->device);
    auto &q = c->p2p;
    const Slice s = p2p_slice(c);
    const size_t npix = (size_t)c->W * c->H, nbytes = npix * 3;
    if (!q.acc_from_bins)  // accumulated by the atomic form (or no depth exchange before): nothing is known
        rtr::launch_p2p_occupancy(c->stream, nullptr, c->W, c->H, q.occ);
    p2p_barrier(c);  // every rank's accumulators are complete (and its reduced-depth slice has been read)
    rtr::launch_p2p_acc_resolve(c->stream, q.accum, q.occ_of, q.ximg, s.first, s.count, q.world, c->W, c->H);
    p2p_barrier(c);  // every image slice is resolved; nobody reads the accumulators any more
    if (q.whole_frame && q.pyramid_done)
        q.image_sliced = true;  // the fused prefilter reads the slices itself and writes RTR_BUF_IMAGE
    else
        rtr::launch_p2p_gather(c->stream, q.image, c->img, s.chunk * 3, nbytes, -1);
    return launch_check(c, "p2p_sum_resolve");
}

int rtr_p2p_render(rtr_ctx *c, const float P[16], int with_filter) {
    if (!c) return RTR_ERR_INVALID;
    NEED(c, P != nullptr, "P is NULL");
    NEED(c, c->p2p.open, "rtr_p2p_open has not been called");
    if (with_filter) {  // fail before any rank enters a barrier the others would wait in
        DevGuard g(c->device);
        if (int rc = ensure_pyramid(c)) return rc;
    }
    // In the tile-binned form both tile launches of the frame visit every pixel, so they can be the
    // only writers of the depth buffer / accumulators (no clear, no read-modify-write), and the
    // accumulate launch -- which loads the GLOBAL depth tile -- can emit the prefilter's pyramid.
    auto &q = c->p2p;
    q.whole_frame = use_tiles(c);
    q.pyramid_done = q.whole_frame && with_filter && c->prm.levels == 4;
    int rc = q.whole_frame ? RTR_OK : rtr_clear(c);
    if (!rc) rc = rtr_min_depth_pass(c, P);
    if (!rc && q.whole_frame && c->list_valid && q.occ_from_scan) {
        // Tile-binned frame: ONE barrier (every rank's min pass and occupancy bitmap are complete; the same
        // launch gathers the bitmaps), then the accumulate launch takes the MIN over the occupying ranks' local
        // depth tile by tile -- no slice reduction, no second barrier (the barrier in front of the colour
        // exchange is what tells a rank that nobody reads its depth buffer any more).
        DevGuard g(c->device);
        q.occ_current = true;
        q.acc_from_bins = false;
        const unsigned long long ticks = 100000ull * (unsigned long long)c->opt_p2p_timeout_ms;
        rtr::launch_p2p_sync_gather(c->stream, q.flags, q.flags_of, q.rank, q.world, ++q.seq, q.status_dev, ticks, q.occ_of,
                                    q.occ_all);
        q.depth_peers = true;
        rc = launch_check(c, "p2p barrier");
    } else if (!rc) {
        rc = rtr_p2p_min_depth(c);
    }
    if (!rc) rc = rtr_accumulate_pass(c, P);
    if (!rc) rc = rtr_p2p_sum_resolve(c);
    const int parts = q.pyramid_done ? rtr::tile_count(c->W, c->H) : 0;
    rtr::Sliced isl{};
    if (q.image_sliced) {
        isl.src = q.image;
        isl.chunk = p2p_slice(c).chunk;
    }
    // One-barrier form: the completed depth lies in `red`.  Every rank is past its accumulate launch (the barriers
    // of the colour exchange), so nobody reads this rank's depth buffer any more: the fused prefilter reads `red`
    // and writes RTR_BUF_DEPTH; without a prefilter (or with another pyramid depth) a device copy completes it.
    const bool from_red = q.depth_in_red;
    const bool fused = with_filter && c->prm.levels == 4;
    q.whole_frame = q.pyramid_done = q.depth_sliced = q.depth_peers = q.image_sliced = q.depth_in_red = false;
    if (!rc && from_red && !fused) {
        DevGuard g(c->device);
        HIP_TRY(c, hipMemcpyAsync(c->depth, q.red, (size_t)c->W * c->H * sizeof(uint32_t), hipMemcpyDeviceToDevice, c->stream));
    }
    if (!rc && with_filter) rc = filter_impl(c, parts, isl.chunk ? &isl : nullptr, (from_red && fused) ? q.red : nullptr);
    return rc;
}

// Owner-computes form of the sharded frame (rtr.h 5b): T1 -> barrier (+ every rank's occupancy bitmap) -> ONE fused tile
// launch over the tiles tile_owner() gives to this rank, reading the other occupying ranks' entries out of their tile
// stores -> barrier -> on the frame's owner only: collect the other ranks' tiles (+ pyramid) -> prefilter.  Six launches,
// two barriers, no MIN / SUM exchange; the next frame's first barrier is what keeps a rank from overwriting tiles the
// previous frame's owner is still collecting (that owner arrives at it only behind its collect and prefilter).
int rtr_p2p_render_owned(rtr_ctx *c, const float P[16], int with_filter, int frame_owner) {
    if (!c) return RTR_ERR_INVALID;
    NEED(c, P != nullptr, "P is NULL");
    NEED(c, c->p2p.open, "rtr_p2p_open has not been called");
    auto &q = c->p2p;
    NEED(c, frame_owner >= 0 && frame_owner < q.world, "frame_owner out of range");
    NEED(c, use_tiles(c), "the owner-computes form needs the tile-binned mode (option mode = 1, <= 4096 tiles)");
    NEED(c, !c->opt_overlap, "the owner-computes form does not combine with option overlap");
    const bool mine = frame_owner == q.rank;
    DevGuard g(c->device);
    if (with_filter && mine)  // (fail before any rank enters a barrier the others would wait in)
        if (int rc = ensure_pyramid(c)) return rc;
    const bool fused = with_filter && c->prm.levels == 4;
    if (int rc = bin_points(c, P, false, false)) return rc;
    NEED(c, c->F().store.meta == q.meta_of.p[q.rank] && c->F().dyn == q.dyn_of.p[q.rank],
         "the tile store changed since rtr_p2p_export (export and open again)");
    const unsigned long long ticks = 100000ull * (unsigned long long)c->opt_p2p_timeout_ms;
    // every rank's stream lengths and occupancy bitmap are final (and gathered into local memory)
    rtr::launch_p2p_sync_gather(c->stream, q.flags, q.flags_of, q.rank, q.world, ++q.seq, q.status_dev, ticks, q.occ_of, q.occ_all);
    rtr::TilePyr pyr{};
    pyr.enable = (mine && fused) ? 1 : 0;
    if (pyr.enable) {
        pyr.L = c->lv;
        pyr.n_eff_rows = (uint32_t)((c->H >> 4) << 4);
        pyr.part_min = c->part_min;
        pyr.part_max = c->part_max;
    }
    rtr::Sliced dsl{};
    dsl.occ_all = q.occ_all;
    dsl.peers = q.world;
    dsl.rank = q.rank;
    dsl.tab = q.tab;
    {   // the frame's owner writes its tiles where the frame ends up; everybody else into the buffers the owner reads
        Timed t(c, RTR_K_TILE);
        rtr::launch_tile(c->stream, 4, c->W, c->H, c->F().store, c->prm.depth_window, c->depth, c->acc, mine ? c->img : q.ximg,
                         c->opt_keep_accum, pyr.enable ? &pyr : nullptr, &dsl);
    }
    mark_consumed(c);
    p2p_barrier(c);  // every tile of the frame is final on the rank that produced it; nobody reads a tile store any more
    if (int rc = launch_check(c, "owned tile frame")) return rc;
    if (!mine) return RTR_OK;
    rtr::launch_p2p_collect(c->stream, c->W, c->H, q.tab, q.occ_all, q.world, q.rank, c->depth, c->img, pyr.enable ? &pyr : nullptr);
    if (int rc = launch_check(c, "collect")) return rc;
    if (with_filter) return filter_impl(c, pyr.enable ? rtr::tile_count(c->W, c->H) : 0);
    return RTR_OK;
}

// ---- buffers -----------------------------------------------------------------------

int rtr_device_buffer(rtr_ctx *c, int which, void **ptr, size_t *bytes) {
    if (!c) return RTR_ERR_INVALID;
    NEED(c, ptr != nullptr, "dev_ptr is NULL");
    if (which != RTR_BUF_MINMAX)
        if (int rc = check_frame(c)) return rc;
    size_t npix = (size_t)c->W * c->H, b = 0;
    void *p = nullptr;
    switch (which) {
        case RTR_BUF_DEPTH: p = c->depth; b = npix * 4; break;
        case RTR_BUF_ACCUM: p = c->acc; b = npix * 16; break;
        case RTR_BUF_IMAGE: p = c->img; b = npix * 3; break;
        case RTR_BUF_TENSOR: p = c->tensor; b = npix * 10; break;
        case RTR_BUF_MASK: p = c->mask; b = npix; break;
        case RTR_BUF_MINMAX: p = c->minmax; b = 8; break;
        default: return fail(c, RTR_ERR_INVALID, "unknown buffer id %d", which);
    }
    *ptr = p;
    if (bytes) *bytes = b;
    return RTR_OK;
}

int rtr_download_buffer(rtr_ctx *c, int which, void *host, size_t bytes) {
    if (!c) return RTR_ERR_INVALID;
    NEED(c, host != nullptr, "host is NULL");
    void *p = nullptr; size_t b = 0;
    int rc = rtr_device_buffer(c, which, &p, &b);
    if (rc) return rc;
    NEED(c, bytes == b, "size mismatch");
    DevGuard g(c->device);
    HIP_TRY(c, hipMemcpyAsync(host, p, b, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, sync_streams(c));
    bool retry = false;
    rc = check_store_error(c, c->last_valid ? &retry : nullptr);
    if (rc || !retry) return rc;
    if ((rc = finish_sync_rerender(c))) return rc;  // (the adaptive extent pool overflowed: the frame again, then the copy)
    HIP_TRY(c, hipMemcpyAsync(host, p, b, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, sync_streams(c));
    return check_store_error(c);
}

// ---- measurement -------------------------------------------------------------------

#ifdef RTR_EXPERIMENT
extern "C" int rtr_debug_stamps(rtr_ctx *c, unsigned long long out[64]) {  // timing-experiment builds only
    if (!c || !out || !c->F().store.meta) return RTR_ERR_INVALID;
    DevGuard g(c->device);
    HIP_TRY(c, hipMemcpyAsync(out, rtr::ts_dbg(c->F().store), 64 * 8, hipMemcpyDeviceToHost, c->stream));
    rtr::read_filter_stamps(c->stream, out + 24);  // (words 24..39: two workgroups of k_filter4; the second tile workgroup's stamps give way)
    HIP_TRY(c, hipMemsetAsync(rtr::ts_dbg(c->F().store) + 40, 0, 8 * 8, c->stream));
    HIP_TRY(c, hipMemsetAsync(rtr::ts_dbg(c->F().store) + 56, 0, 8 * 8, c->stream));  // (the per-wave maxima / sums of T1)
    HIP_TRY(c, sync_streams(c));
    return RTR_OK;
}
#endif

int rtr_frame_stats(rtr_ctx *c, uint32_t out[8]) {
    if (!c) return RTR_ERR_INVALID;
    NEED(c, out != nullptr, "out is NULL");
    NEED(c, c->F().store.meta != nullptr, "no binned frame yet");
    DevGuard g(c->device);
    if (c->last_lean) rtr::launch_lean_fold(c->stream, c->W, c->H, c->F().store, c->lean_parity);  // (lean frames fold lazily)
    HIP_TRY(c, hipMemcpyAsync(out, rtr::ts_hdr(c->F().store), 8 * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, sync_streams(c));
    return RTR_OK;
}

int rtr_timing_enable(rtr_ctx *c, int on) {
    if (!c) return RTR_ERR_INVALID;
    DevGuard g(c->device);
    (void)collect_timing(c);
    c->timing = on < 0 ? 0 : (on > 4 ? 1 : on);
    c->timing_tick = 0;
    return RTR_OK;
}

int rtr_timing_reset(rtr_ctx *c) {
    if (!c) return RTR_ERR_INVALID;
    DevGuard g(c->device);
    (void)collect_timing(c);
    for (int k = 0; k < RTR_K_COUNT; ++k) { c->total_ms[k] = 0; c->launches[k] = 0; }
    return RTR_OK;
}

int rtr_timing_get(rtr_ctx *c, int k, double *total_ms, uint64_t *launches) {
    if (!c) return RTR_ERR_INVALID;
    NEED(c, k >= 0 && k < RTR_K_COUNT, "bad kernel id");
    DevGuard g(c->device);
    if (int rc = collect_timing(c)) return rc;
    if (total_ms) *total_ms = c->total_ms[k];
    if (launches) *launches = c->launches[k];
    return RTR_OK;
}

}  // extern "C"
