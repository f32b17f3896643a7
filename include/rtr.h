/*
 * rtr.h -- C ABI of the MI355X point-cloud -> framebuffer projector (librtr_hip.so).
 *
 * Drop-in boundary for the hot path of the reference's `ProjectCloud` class
 * (reference: src/RTRenderer/include/project_cloud.h:11-19).  The reference has no
 * C ABI / FFI of its own -- its boundary is a C++ class dragging in OpenCV, glm
 * and libtorch -- so these entry points are what a binding of that class would
 * need: plain pointers and sizes, no torch / OpenCV / glm types.  The header-
 * compatible C++ facade (include/rtr_project_cloud.hpp) and the Python mirror are
 * thin wrappers over exactly these calls; INTEGRATION.md shows the reference-side
 * binding.
 *
 * Conventions
 *   - every call returns RTR_OK (0) or a negative rtr_status; rtr_last_error()
 *     gives the text.  The library never calls exit() (the reference does:
 *     project_cloud.cu:13-17).
 *   - one context = one GPU = one host thread at a time (project_cloud.cu is not
 *     re-entrant either).  Multi-GPU = one process (context) per GPU; the caller
 *     reduces the exposed device buffers between the phase calls (section 5).
 *   - all work is enqueued on the context's HIP stream; calls that fill host
 *     buffers synchronise that stream before returning, others do not.
 *   - the HIP extension is the only implementation: there is no CPU fallback.
 */
#ifndef RTR_H
#define RTR_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Bumped whenever a struct a caller allocates changes size or an entry point changes meaning (2: rtr_p2p_handles grew
 * from 6 to 9 handle blocks, RTR_ERR_INTERNAL, the asynchronous outputs).  A binding compares rtr_abi_version() -- what
 * the loaded library was built with -- against the RTR_ABI_VERSION it was compiled with and refuses to go on when they
 * differ (the Python mirror and rtr::ProjectCloud do). */
#define RTR_ABI_VERSION 2
#define RTR_EMPTY_DEPTH 0x7F7FFFFFu /* render.cu:166, project_cloud.cu:316: bits of FLT_MAX */

typedef struct rtr_ctx rtr_ctx;

typedef enum {
    RTR_OK = 0,
    RTR_ERR_INVALID = -1,   /* bad argument / call order                         */
    RTR_ERR_HIP = -2,       /* a HIP runtime call failed (text in last_error)    */
    RTR_ERR_NO_OUTPUT = -3, /* both host outputs NULL (project_cloud.cu:270-273) */
    RTR_ERR_UNSUPPORTED = -4,/* e.g. filter with W % 2^levels != 0 (quirk Q3)    */
    RTR_ERR_INTERNAL = -5   /* the tile store dropped entries: frames rendered since the last synchronising call
                               are incomplete (returned by rtr_synchronize and by every call that copies results
                               to the host; cannot happen unless the extent pool is mis-sized) */
} rtr_status;

/* Compile-time constants of the reference, made run-time parameters. */
typedef struct {
    float depth_window;       /* render.cu:106            default 0.02f  */
    float filter_strength;    /* project_cloud.cu:24      default 1.025f */
    float gradient_threshold; /* project_cloud.cu:25      default 0.03f  */
    int32_t levels;           /* project_cloud.cu:23      default 4      */
} rtr_params;

typedef enum { RTR_SCENE_UNIFORM_BOX = 0, RTR_SCENE_ROOM_SHELL = 1 } rtr_scene;

/* ---- 1. life cycle (ProjectCloud ctor/dtor, project_cloud.cu:189-266) ---------- */
int rtr_abi_version(void);
/* device: HIP ordinal of the GPU this context owns. */
int rtr_device_count(void); /* HIP devices visible to this process (0 when there is none or HIP fails) */
int rtr_create(rtr_ctx **out, int device);
int rtr_destroy(rtr_ctx *ctx);
const char *rtr_last_error(const rtr_ctx *ctx); /* ctx may be NULL: error of a failed rtr_create */
void rtr_default_params(rtr_params *p);
int rtr_set_params(rtr_ctx *ctx, const rtr_params *p);
int rtr_get_params(const rtr_ctx *ctx, rtr_params *p);
/* Tuning knobs that never change the frame.
 *  "mode": 1 (default) tile-binned -- the cloud is streamed ONCE, every in-frustum point is
 *          appended as one 8-byte entry to the stream of its screen tile and a per-tile LDS
 *          z-buffer does min + accumulate + resolve (no global atomics on the frame buffers);
 *          0 = the reference's structure: two full passes with atomicMin / atomicAdd
 *          (render.cu:53-130).
 *  "split_threshold" (default 32768; 0 = never), "split_slice" (default 16384): a screen tile
 *          holding more entries than the threshold is processed by several workgroups, each
 *          taking a slice of at least split_slice entries (a distant overview that packs the
 *          whole cloud into a few tiles would otherwise serialise on them).  Whole-frame calls launch the
 *          kernel that does this only while such tiles have been seen (the library learns it from the frames
 *          that have completed, without a sync; eight frames of grace after an upload, a new resolution or a
 *          change of these options): the first frame(s) of a view that suddenly packs the cloud into a few
 *          tiles are therefore rendered unsplit -- exact, slower -- until the report arrives.
 *  "auto_reorder": what follows every later rtr_upload_points / rtr_generate_synthetic.  2 (default): the
 *          cloud is Morton-sorted once (rtr_reorder_points) when its 256-point chunks are not spatially
 *          compact -- mean chunk-box diagonal above twice what an ideally ordered volume cloud of that
 *          size has, (256 / n)^(1/3) of the cloud's diagonal.  Scanner sweeps and Morton-like surfaces
 *          pass; a hash-ordered cloud and the reference loader's 0.25 m blocks (unordered inside) are
 *          sorted, which is worth 13 % ... 2.4 x per frame.  1: always, 0: never.  Best effort
 *          (skipped when the sort's scratch does not fit); clouds under 65536 points are left alone.
 *          Frames never depend on the point order; rtr_download_points returns the RESIDENT order, a
 *          permutation of the uploaded one when rtr_get_option("reordered") reads 1
 *          ("order_ratio_ppm": the measured chunk / cloud diagonal ratio in millionths).
 *  "cull": 1 = skip 256-point chunks whose bounding box is provably outside the frustum
 *          (exact: same frame; an algorithmic byte reduction, off by default and reported
 *          separately from the roofline figure; needs a spatially coherent point order).
 *  "lean": 1 (default) = a whole single-GPU frame (rtr_render and the calls built on it) whose point kernel needs no
 *          epilogue -- no tile above "split_threshold" seen lately, option "overlap" off, no peer-to-peer exchange open
 *          -- ends that kernel without last-workgroup detection and bookkeeping pass: the tile kernel's workgroups read
 *          and reset their stream counters themselves, one extra workgroup does the frame's bookkeeping off the
 *          critical path (5-6 us per frame).  0 = always the epilogue.  After a lean frame rtr_accumulate_pass
 *          re-projects the cloud (the bins were consumed).
 *  "lean_identity": 1 (default) = in a lean frame tile workgroup b takes tile b -- no look-up in the launch order
 *          (heaviest tiles first) -- whenever every workgroup of the tile launch is resident at once (1080p: 2041
 *          of 2048 slots), where the order buys nothing and costs a dependent memory round trip.  0 = always the order.
 *  "lean_early": lean frames: the tile workgroups request their first batch of entries BEFORE they know the
 *          streams' lengths (one round trip less; right when most tiles are full) -- 1 always, 0 never (after the
 *          lengths, from indices clamped to them: a near-empty tile then moves no stale memory), -1 (default) = when
 *          the previous frame had at least 1024 entries per tile on average.  Same frame either way.
 *  "lane_test": 1 (default) = the tile-binned point kernel first decodes and projects ONE point per lane (of
 *          the four consecutive points a lane holds) and bounds the other three by the chunk's "lane spread"
 *          (the largest coordinate difference inside a lane, measured once per upload): a 256-point chunk with
 *          no lane near the frustum is neither decoded nor projected in full.  Every point still goes through
 *          the exact arithmetic before it can reach a pixel; 0 = every point of every chunk, as before.
 *  "pack": the tile-binned point kernel reads the coordinates from a LOSSLESS packed form, built once after
 *          every upload / generation / sort (and at once for the resident cloud when the option is set): per
 *          256-point chunk and axis the fp32 bit patterns are a common prefix + the 0..25 (or 32) bits below it
 *          of every value, i.e. 5-9 B/pt instead of 12 for spatially ordered clouds (neighbours share sign,
 *          exponent and leading mantissa bits), in two bit streams per axis -- every lane's first value, and its
 *          other three -- so that the chunks the point kernel rejects on one point per lane are read through a
 *          quarter of their bytes; any bit pattern round-trips (NaNs, -0, mixed signs take 32 bits).  1 (default):
 *          used when it saves at least 1/8 of the stream; 0: never; 2: always, and the packed form is
 *          decoded and compared with the SoA arrays once (an error if a single point differs).  rtr_get_option:
 *          "packed" (1: in use), "packed_millibytes_per_point" (coordinate stream incl. headers, 12000 = raw).
 *  "keep_soa": 0 (default) = a packed cloud is resident in packed form ONLY: the fp32 SoA arrays (12 B per point) are
 *          freed once the cloud has been packed and decoded again -- bit for bit -- by the calls that read fp32
 *          coordinates (mode 0 and the phase calls with another matrix, rtr_reorder_points, rtr_download_points,
 *          option "pack" = 0); 1 = they stay resident beside the packed form.
 *  "pool_worst_case": 0 (default) = the pool of dynamic stream extents is sized by the frames the cloud has had (8 x the
 *          most in-frustum entries a completed frame reported, at least n / 2: 4 B per point instead of 16).  A frame
 *          whose entries jump past that overflows it; the next synchronising call (every call that copies results to the
 *          host, rtr_synchronize, rtr_download_buffer) then sizes the pool for the worst case and renders that frame
 *          again before it returns -- transparent, except for whoever consumes frames on the stream without ever
 *          synchronising, and for rtr_wait (asynchronous outputs), which reports RTR_ERR_INTERNAL once and asks for
 *          the frame again.  1 = sized for the worst case (2 n entries) from the start.
 *          rtr_get_option("resident_millibytes_per_point"): device memory held for the cloud and its frames, per point.
 *  "keep_accum": 1 = the whole-frame calls also write RTR_BUF_ACCUM (default 0; the phase
 *          calls always do).
 *  "overlap": 1 = rtr_render queues the point stream (T1) of a frame on a second, internal
 *          stream and alternates between two list / bin sets, so it may run beside the tail of
 *          the previous frame; results are still ordered on the context's stream.  Default 0:
 *          on MI355X the gain is 0-3 % ("Overlap").  "tail_cus" = t (0..31, set
 *          before "overlap") additionally gives the two streams disjoint CU masks, t CUs of every
 *          XCD for the tail.
 *  "point_grid": workgroups of the grid-stride point kernels (default 1024 = 4 per CU; at the default the
 *          tile-binned point kernel takes what is resident at once: 5 per CU when it reads packed coordinates
 *          and its registers admit it, else 4).
 *  "phases": the point kernel's workgroups are cut into this many groups that start at different
 *          places of the cloud (default 0 = automatic: 1, which keeps one dense streaming front, unless the
 *          previous frame had more than a quarter of the cloud inside the frustum, then 16 -- the claims of
 *          a distant overview spread over more stream counters); "fill_shift": the per-tile stream counters
 *          lie 4 << value bytes apart (0..6; packed counters share memory channels and queue up: a distant
 *          overview with every point in a dozen tiles takes 3.0 ms in the point kernel at 2, 1.5 ms at 4, for +1 %
 *          on an ordinary view).  Default -1 = automatic: 1 (8 bytes) on ordinary frames, 4 (64 bytes) while frames
 *          with tiles above "split_threshold" have been seen lately (and in the phase / sharded calls).
 *  "p2p_timeout_ms": how long a flag barrier of the peer-to-peer exchange (section 5b) waits for a rank
 *          that does not arrive before it flags the frame in rtr_p2p_status (default 2000).  rtr_get_option("p2p_open")
 *          reads 1 while the peers' buffers are mapped: a new cloud (rtr_upload_points, rtr_generate_synthetic with
 *          another point count) or resolution closes the exchange on this rank -- every rank must then export /
 *          open again, together.  The exchange and option "overlap" exclude each other.
 *  "debug_dyn_cap": test aid -- caps the pool of dynamic stream extents at this many entries (-1 = off), so that a
 *          heavy tile overflows it and the error path (RTR_ERR_INTERNAL) can be exercised.
 *  "xp": only in RTR_EXPERIMENT builds (make experiment): switches parts of the point kernel off for
 *          timing attribution -- frames are WRONG while it is non-zero; the shipped library rejects it.
 *  "probe_variant": measurement aid of tools/probe_variants.py (selects the rtr_stream_probe kernel). */
int rtr_set_option(rtr_ctx *ctx, const char *key, int value);
/* Reads an option back; also "p2p_open" (see "p2p_timeout_ms"), "reordered" (1: the resident cloud was sorted by the library),
 * "order_ratio_ppm" (mean chunk-box diagonal / cloud diagonal as uploaded, in millionths), "packed" and
 * "packed_millibytes_per_point" (see "pack"). */
int rtr_get_option(rtr_ctx *ctx, const char *key, int *value);
/* Run on a caller-owned hipStream_t (e.g. torch's current stream) instead of the context's
 * own non-blocking stream; NULL means HIP's default stream.  rtr_reset_stream returns to the
 * private stream. */
int rtr_set_stream(rtr_ctx *ctx, void *hip_stream);
int rtr_reset_stream(rtr_ctx *ctx);
int rtr_synchronize(rtr_ctx *ctx);

/* ---- 2. the resident cloud (project_cloud.cu:191-206) --------------------------- */
/* Copies n points from HOST memory and stores them as SoA x[] y[] z[] + packed
 * colour.  xyz_stride_bytes = 16 for the reference's float4 (x,y,z,1) layout
 * (Octreegrid.h:162-170), 12 for tight xyz.  rgb_stride_bytes = 4 for uchar4
 * (c0,c1,c2,255) (Octreegrid.h:172-180), 3 for tight triples; channel order is
 * preserved end to end (B,G,R in the reference).  Replaces any previous cloud. */
int rtr_upload_points(rtr_ctx *ctx, const float *xyz, size_t xyz_stride_bytes, const uint8_t *rgb,
                      size_t rgb_stride_bytes, size_t n);
/* Synthesises points [first, first+count) of a `total`-point scene directly in
 * HBM (counter-based generator; bit-identical to the oracle's). */
int rtr_generate_synthetic(rtr_ctx *ctx, int scene, uint64_t seed, uint64_t first, uint64_t count,
                           uint64_t total);
/* One-off Morton (Z-order) sort of the resident cloud: consecutive points become spatial
 * neighbours, like the reference loader's 0.25 m block order (cloudreader.cpp:8-82).  Never
 * changes a frame (min and integer sums commute); it changes rtr_download_points' order and
 * makes the per-tile appends long runs and option "cull" effective. */
int rtr_reorder_points(rtr_ctx *ctx);
int rtr_num_points(const rtr_ctx *ctx, uint64_t *n);
/* Copies the resident cloud back as float4 / uchar4 AoS (tests, debugging), in the RESIDENT order (see
 * option "auto_reorder"). */
int rtr_download_points(rtr_ctx *ctx, float *xyzw, uint8_t *rgba, uint64_t first, uint64_t count);

/* ---- 3. camera (project_cloud.cu:318, project_cloud.h:50-59) -------------------- */
/* P = K4 * E in fp32, row-major, exactly as the reference composes it with glm:
 * K row-major 3x3 intrinsics, E row-major 4x4 world->camera, both double. */
int rtr_compose_projection(const double K[9], const double E[16], float P[16]);
/* (Re)allocates the frame buffers; cheap no-op when unchanged (project_cloud.cu:275-298). */
int rtr_set_resolution(rtr_ctx *ctx, int width, int height);

/* ---- 4. whole-frame calls (the reference's public methods) ---------------------- */
/* computeRGBD (project_cloud.cu:268-312): clear, min-depth pass, accumulate pass,
 * resolve.  host_img: W*H*3 u8 or NULL; host_depth: W*H float or NULL (empty
 * pixel = FLT_MAX).  Both NULL -> RTR_ERR_NO_OUTPUT like the reference's -1;
 * use rtr_render() for device-resident output. */
int rtr_project(rtr_ctx *ctx, const float P[16], uint8_t *host_img, float *host_depth);
/* computeFilteredRGBD (project_cloud.cu:394-434): rtr_project + depth-heuristic
 * prefilter; masked pixels read depth -1 and colour 0; also fills the fp16
 * {1,5,H,W} device tensor consumed by the U-Net (project_cloud.cu:471). */
int rtr_project_filtered(rtr_ctx *ctx, const float P[16], uint8_t *host_img, float *host_depth);
/* Same frame sequences without any host copy or host sync (outputs stay in HBM). */
int rtr_render(rtr_ctx *ctx, const float P[16], int with_filter);

/* ---- 4b. the same frames with the host copies off the critical path ---------------
 * The reference's call shape runs the kernels and then the two device-to-host copies (W*H*4 + W*H*3 bytes:
 * 14.5 MB at 1080p, ~0.29 ms over PCIe) one after the other, every frame (project_cloud.cu:302-309,424-431).
 * rtr_project_async renders a frame like rtr_project / rtr_project_filtered, snapshots depth and image on the
 * device and queues their copies into the library's PINNED host buffers of `slot` on a second stream; it does
 * not wait.  With the slots used in rotation, frame k's copies run beside frame k + 1's kernels.
 * rtr_wait(slot) blocks until that slot's outputs are complete (slot = -1: all of them) and reports errors of
 * the frames since the last synchronising call; the buffers (rtr_host_output_buffers: W*H*3 u8, W*H float,
 * valid until the next rtr_set_resolution) may be read until the slot is used again.  A caller that needs the
 * frame in its own arrays copies from there (or keeps using the synchronous calls). */
#define RTR_ASYNC_SLOTS 2
int rtr_host_output_buffers(rtr_ctx *ctx, int slot, uint8_t **img, float **depth);
int rtr_project_async(rtr_ctx *ctx, const float P[16], int slot, int with_filter);
int rtr_wait(rtr_ctx *ctx, int slot);

/* ---- 5. phase calls (multi-GPU sharding: reduce between phases) ----------------- */
/*   rtr_clear -> rtr_min_depth_pass -> [all-reduce MIN of depth]
 *   -> rtr_accumulate_pass -> [all-reduce / reduce-scatter SUM of accum]
 *   -> rtr_resolve -> rtr_filter (optional)                                          */
int rtr_clear(rtr_ctx *ctx);                           /* render.cu:16-31 + project_cloud.cu:317 */
int rtr_min_depth_pass(rtr_ctx *ctx, const float P[16]);  /* render.cu:53-83   */
int rtr_accumulate_pass(rtr_ctx *ctx, const float P[16]); /* render.cu:85-130  */
int rtr_resolve(rtr_ctx *ctx);                         /* render.cu:132-163 */
/* Resolve only pixels [first_pixel, first_pixel + count) (first_pixel % 4 == 0) into
 * RTR_BUF_IMAGE, reading the accumulators either from RTR_BUF_ACCUM (acc_dev NULL) or from
 * a caller-owned device array of count*4 u32 -- the slice a reduce-scatter hands to a rank. */
int rtr_resolve_range(rtr_ctx *ctx, const void *acc_dev, uint64_t first_pixel, uint64_t count);
int rtr_filter(rtr_ctx *ctx);                          /* project_cloud.cu:331-392 */

/* ---- 5b. peer-to-peer exchange for the phase calls (one process per GPU on one node) ----
 * The hand-written counterpart of the two collectives above: every rank maps
 * the other ranks' frame buffers through hipIpc and pulls over xGMI.  Set up once per
 * resolution: each rank calls rtr_p2p_export, the handle blocks are exchanged by the host (any
 * transport: they are plain bytes), then every rank calls rtr_p2p_open with all of them.
 *   rtr_clear -> rtr_min_depth_pass -> rtr_p2p_min_depth     (RTR_BUF_DEPTH := MIN over ranks)
 *   -> rtr_accumulate_pass -> rtr_p2p_sum_resolve             (RTR_BUF_IMAGE := resolve(SUM of
 *   RTR_BUF_ACCUM over ranks); RTR_BUF_ACCUM itself stays local) -> rtr_filter (optional)
 * All ranks must issue the same sequence of rtr_p2p_* calls.  A rank that does not arrive
 * within the barrier timeout (option "p2p_timeout_ms", default 2 s) is flagged in rtr_p2p_status on the
 * ranks that waited for it; their frames since then are undefined.  The caller polls rtr_p2p_status
 * (a host word, no synchronisation), agrees with the other ranks and falls back to the collectives --
 * sharded.ShardedProjector does that every `check_every` frames.  rtr_set_resolution closes the mapping. */
#define RTR_P2P_MAX_RANKS 16
typedef struct rtr_p2p_handles {
    unsigned char depth[64], accum[64], image[64], reduced[64], flags[64], tiles[64];
    /* one hipIpcMemHandle_t each: the depth buffer and accumulators, exchange copies of the resolved
     * image and the reduced depth, the barrier flags, the per-tile occupancy bitmap */
    unsigned char store_meta[64], store_ext0[64], store_dyn[64];
    /* ... and this rank's tile store (stream lengths and extent directory, static extents, dynamic extent pool): the
     * owner-computes form reads the peers' entries out of it.  The store is sized by the resident cloud: export (and
     * open) again after rtr_upload_points / rtr_generate_synthetic. */
} rtr_p2p_handles;
int rtr_p2p_export(rtr_ctx *ctx, rtr_p2p_handles *mine);
int rtr_p2p_open(rtr_ctx *ctx, int rank, int world, const rtr_p2p_handles *all /* [world] */);
int rtr_p2p_close(rtr_ctx *ctx);
int rtr_p2p_min_depth(rtr_ctx *ctx);
int rtr_p2p_sum_resolve(rtr_ctx *ctx);
/* The whole sharded frame in one call: rtr_clear, rtr_min_depth_pass, rtr_p2p_min_depth,
 * rtr_accumulate_pass, rtr_p2p_sum_resolve and, if with_filter, rtr_filter -- trimmed in the tile-binned form
 * to eight launches (no clear, no slice reduction of the depth: one barrier, then the accumulate launch takes
 * the MIN over the occupying ranks' depth tiles itself; no gathers).  Afterwards RTR_BUF_DEPTH,
 * RTR_BUF_IMAGE (and the prefilter's outputs) hold the GLOBAL frame on every rank; RTR_BUF_ACCUM holds
 * this rank's partial sums, and in the tile-binned form only under the screen tiles that contain
 * points of this rank (the peers read nothing else: tiles without local points are not written). */
int rtr_p2p_render(rtr_ctx *ctx, const float P[16], int with_filter);
/* The owner-computes form of the sharded frame (tile-binned mode): every screen tile is produced by ONE of the ranks
 * that have points in it, which reads the other occupying ranks' entries out of their tile stores over xGMI and runs the
 * fused per-tile z-buffer once -- no MIN / SUM exchange, no second tile pass, two barriers and at most six launches per
 * frame.  Afterwards only rank `frame_owner` holds the GLOBAL frame (RTR_BUF_DEPTH / IMAGE and the prefilter's outputs:
 * it collects the other ranks' tiles); the buffers of the other ranks hold their own tiles only.  All ranks must pass
 * the same frame_owner; rotating it (frame k -> rank k mod world) spreads the collect + prefilter over the ranks, e.g.
 * one U-Net consumer per GPU.  Tiles are not split over workgroups in this form (split_threshold is ignored).
 * PRECONDITION on the ranks that are NOT the frame's owner: the call returns (its work queued) right after the second
 * barrier, while the owner is still reading this rank's RTR_BUF_DEPTH and image exchange copy over the mappings.  Only
 * the next rtr_p2p_render_owned's first barrier orders those reads against new writes: between two owned frames a
 * non-owner must not queue anything else that writes the frame buffers (rtr_render, rtr_clear, the phase calls,
 * rtr_p2p_render) without a barrier of the caller's own across the ranks (the Python ShardedProjector and bench.py
 * put a torch.distributed barrier / all-reduce there). */
int rtr_p2p_render_owned(rtr_ctx *ctx, const float P[16], int with_filter, int frame_owner);
int rtr_p2p_status(rtr_ctx *ctx, uint32_t *barrier_timeouts);

/* ---- 6. device-resident buffers (owned by the context, valid until the next
 *         rtr_set_resolution / rtr_destroy) ---------------------------------------- */
typedef enum {
    RTR_BUF_DEPTH = 0,  /* u32 [H*W]    float bits, RTR_EMPTY_DEPTH when empty (project_cloud.h:34) */
    RTR_BUF_ACCUM = 1,  /* u32 [H*W*4]  (sum c0, sum c1, sum c2, count)          (project_cloud.h:33) */
    RTR_BUF_IMAGE = 2,  /* u8  [H*W*3]  interleaved, input channel order         (project_cloud.h:26) */
    RTR_BUF_TENSOR = 3, /* f16 [5*H*W]  planar {1,5,H,W}                         (project_cloud.h:32) */
    RTR_BUF_MASK = 4,   /* u8  [H*W]    final keep-mask of the prefilter                              */
    RTR_BUF_MINMAX = 5  /* u32 [2]      depth min / max bits (project_cloud.h:30-31)                  */
} rtr_buffer;
int rtr_device_buffer(rtr_ctx *ctx, int which, void **dev_ptr, size_t *bytes);
/* Synchronous device->host copy of one buffer (bytes must equal its size). */
int rtr_download_buffer(rtr_ctx *ctx, int which, void *host, size_t bytes);

/* ---- 7. measurement -------------------------------------------------------------- */
typedef enum {
    RTR_K_CLEAR = 0, RTR_K_MIN_DEPTH = 1, RTR_K_ACCUMULATE = 2, RTR_K_RESOLVE = 3, RTR_K_FILTER = 4,
    RTR_K_PROBE = 5, RTR_K_TILE = 6, RTR_K_BIN = 7, RTR_K_COUNT = 8
} rtr_kernel_id;
/* mode 1: RTR_K_MIN_DEPTH = streaming projection + append to the tile streams (T1), RTR_K_TILE =
 * per-tile z-buffer (T4); RTR_K_BIN is unused since the tile sort was removed (kept for ABI
 * stability); RTR_K_CLEAR / ACCUMULATE / RESOLVE are then only used by the phase calls. */
/* Read-only probe: same loads and projection arithmetic as the point passes, no frame-
 * buffer traffic -- measures the streaming ceiling of the access pattern. */
int rtr_stream_probe(rtr_ctx *ctx, const float P[16]);
/* on = 1: every phase is bracketed by hipEvents on the context's stream; on = 2: only the
 * streaming point kernels (RTR_K_MIN_DEPTH, RTR_K_ACCUMULATE), i.e. two event records per
 * frame; on = 3: like 2 but only every 4th launch is bracketed (a bracket costs ~8 us of stream
 * time, 3 % of a frame), on = 4: every 2nd; 0: off.  The tile-binned form's RTR_K_MIN_DEPTH launch carries its two
 * events in the dispatch itself (start / stop stamps of that kernel: no extra packets, but a timed
 * dispatch still costs ~10 us of stream time).  rtr_timing_get synchronises and returns the
 * accumulated device time and the number of bracketed launches. */
int rtr_timing_enable(rtr_ctx *ctx, int on);
/* Statistics of the last binned frame (mode 1; synchronises the stream): out[0] work items of the
 * tile kernel, [1] of them slices of split tiles, [2] in-frustum entries, [3] entries of the
 * heaviest tile, [4] slice size used, [5] tile-store error bits of THAT frame's point kernel (0 = none; 1 = an extent
 * never appeared, 2 = the extent pool overflowed: entries were dropped; the tile kernels' own bits -- 4 = a contested
 * tile of an owner-computes sharded frame had more stream pieces than its table holds, 8 = the split tiles' second
 * phase gave up waiting for the first -- go straight to the word every synchronising call checks and surface there
 * as RTR_ERR_INTERNAL), [6] split tiles, [7] 256-point chunks with
 * at least one in-frustum point (each loads 1 KiB of colours). */
int rtr_frame_stats(rtr_ctx *ctx, uint32_t out[8]);
int rtr_timing_reset(rtr_ctx *ctx);
int rtr_timing_get(rtr_ctx *ctx, int kernel, double *total_ms, uint64_t *launches);

#ifdef __cplusplus
}
#endif
#endif /* RTR_H */
