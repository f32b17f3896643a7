/*
 * rtr_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * A plain-C restatement of the reference's point-cloud -> framebuffer projector
 * (EDM-Research/Real-time-Neural-Rendering-of-LiDAR-Point-Clouds, src/RTRenderer).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this library; the shipped HIP path never links, imports or calls it.
 *
 * PARITY UNPINNED: the reference ships no tests, golden vectors or fixtures for
 * this path and its CUDA sources cannot be built or
 * run in this image (no nvcc, glm, OpenCV, libtorch-CUDA).  The oracle is pinned
 * only by hand-derived known-answer tests (tests/test_oracle_kat.py) that follow
 * the cited reference lines.
 *
 * Arithmetic contract (binding for this oracle AND for the HIP kernels):
 *   every fp32 operation below is a single IEEE-754 round-to-nearest-even
 *   operation; products feeding a sum are contracted to fmaf ONLY where written.
 *   Build with -ffp-contract=off -fno-fast-math (orc_selftest() checks it).
 *
 * Reference lines followed (all under /root/reference/src/RTRenderer/):
 *   src/render.cu:16-31    fillBuffer            -> orc_clear
 *   src/render.cu:33-40    matmul                -> project_point (rows 0..2)
 *   src/render.cu:53-83    minDepthPass          -> orc_min_depth_pass
 *   src/render.cu:85-130   accumulatePass        -> orc_accumulate_pass
 *   src/render.cu:132-163  resolvePass           -> orc_resolve
 *   src/render.cu:166-240  find_*_minmax_kernel  -> depth_minmax
 *   src/project_cloud.cu:20-26    constants      -> orc_params defaults
 *   src/project_cloud.cu:28-53    reduce         -> f_reduce
 *   src/project_cloud.cu:55-79    laplacianKernel-> f_laplacian
 *   src/project_cloud.cu:81-126   compareImgs    -> f_compare
 *   src/project_cloud.cu:128-161  resizeKernel   -> f_resize
 *   src/project_cloud.cu:163-187  removeMask     -> f_remove_mask
 *   src/project_cloud.cu:314-329  computeRGBDInternal -> orc_project
 *   src/project_cloud.cu:331-392  applyDepthFilter    -> orc_filter
 *   include/project_cloud.h:50-59, src/CameraCalibration.cpp:17-27,
 *   src/project_cloud.cu:318      matrix composition  -> orc_compose_projection
 */
#include <math.h>
#include <pthread.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_EMPTY_DEPTH 0x7F7FFFFFu /* render.cu:166, project_cloud.cu:316 */

typedef struct {
    float depth_window;       /* render.cu:106           0.02f  */
    float filter_strength;    /* project_cloud.cu:24     1.025f */
    float gradient_threshold; /* project_cloud.cu:25     0.03f  */
    int levels;               /* project_cloud.cu:23     4      */
} orc_params;

void orc_default_params(orc_params *p) {
    p->depth_window = 0.02f;
    p->filter_strength = 1.025f;
    p->gradient_threshold = 0.03f;
    p->levels = 4;
}

/* ------------------------------------------------------------------------ */
/* single-rounding helpers (no contraction: -ffp-contract=off)               */
static inline float f_mul(float a, float b) { return a * b; }
static inline float f_add(float a, float b) { return a + b; }
static inline float f_sub(float a, float b) { return a - b; }
static inline float f_div(float a, float b) { return a / b; }
static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

/* returns 1 when the build honours the contract (no implicit fma, RNE mode) */
int orc_selftest(void) {
    volatile float a = 1.0f + 0x1p-12f, b = 1.0f + 0x1p-12f, c = -(1.0f + 0x1p-11f);
    float plain = f_add(f_mul(a, b), c); /* a*b = 1+2^-11+2^-24 rounds to 1+2^-11 -> 0 */
    float fused = fmaf(a, b, c);         /* exact 2^-24 */
    if (plain != 0.0f) return 0;
    if (fused != 0x1p-24f) return 0;
    volatile float h = 0.5f, t = 2.5f;
    if (rintf(h) != 0.0f || rintf(t) != 2.0f) return 0;
    return 1;
}

/* ------------------------------------------------------------------------ */
/* fp16 <-> fp32, round-to-nearest-even, NaN canonicalised to 0x7E00          */
static uint16_t f32_to_f16(float f) {
    uint32_t x = f2u(f);
    uint32_t sign = (x >> 16) & 0x8000u;
    uint32_t ax = x & 0x7FFFFFFFu;
    if (ax > 0x7F800000u) return 0x7E00u;                      /* NaN (canonical) */
    if (ax >= 0x47800000u) return (uint16_t)(sign | 0x7C00u);  /* >= 65536 (incl. inf) -> inf */
    if (ax >= 0x477FF000u) return (uint16_t)(sign | 0x7C00u);  /* rounds up to 65536 -> inf */
    if (ax < 0x33000001u) return (uint16_t)sign;               /* <= 2^-25 -> 0 (tie to even) */
    int e = (int)(ax >> 23) - 127;
    uint32_t m = (ax & 0x7FFFFFu) | 0x800000u; /* 24-bit significand */
    int shift;
    uint32_t base;
    if (e < -14) { /* subnormal half: value = m * 2^(e-23), unit 2^-24 */
        shift = -e - 1; /* 23 - (e + 24) */
        base = 0;
    } else {
        shift = 13;
        base = (uint32_t)(e + 15) << 10;
        m &= 0x7FFFFFu;
    }
    uint32_t q = m >> shift;
    uint32_t rem = m & ((1u << shift) - 1u);
    uint32_t half = 1u << (shift - 1);
    if (rem > half || (rem == half && (q & 1u))) q++;
    return (uint16_t)(sign | (base + q)); /* carry into exponent is correct by construction */
}

static float f16_to_f32(uint16_t h) {
    uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    uint32_t e = (h >> 10) & 0x1Fu, m = h & 0x3FFu;
    if (e == 0) {
        if (m == 0) return u2f(sign);
        float v = (float)m * 0x1p-24f;
        return sign ? -v : v;
    }
    if (e == 31) return u2f(sign | 0x7F800000u | (m << 13));
    return u2f(sign | ((e + 112u) << 23) | (m << 13));
}

uint16_t orc_f32_to_f16(float f) { return f32_to_f16(f); }
float orc_f16_to_f32(uint16_t h) { return f16_to_f32(h); }

/* ------------------------------------------------------------------------ */
/* A0: camera matrix.  project_cloud.cu:318 computes
 *   transpose( mat4(transpose(Kglm)) * E )  in fp32 with glm,
 * i.e. P = K4 * E, K4 = [[K,0],[0,1]], uploaded row-major.  Each entry is the
 * left-to-right fp32 sum of four separately rounded fp32 products (glm's
 * operator* without fma).  glm is an un-vendored, unpinned dependency of the
 * reference: parity unpinned at this boundary; for zero-skew K the result does
 * not depend on the summation order (at most two non-zero terms).             */
void orc_compose_projection(const double K[9], const double E[16], float P[16]) {
    float K4[16], Ef[16];
    memset(K4, 0, sizeof K4);
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) K4[4 * r + c] = (float)K[3 * r + c];
    K4[15] = 1.0f;
    for (int i = 0; i < 16; ++i) Ef[i] = (float)E[i];
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) {
            float s = f_mul(K4[4 * r + 0], Ef[0 + c]);
            s = f_add(s, f_mul(K4[4 * r + 1], Ef[4 + c]));
            s = f_add(s, f_mul(K4[4 * r + 2], Ef[8 + c]));
            s = f_add(s, f_mul(K4[4 * r + 3], Ef[12 + c]));
            P[4 * r + c] = s;
        }
}

/* ------------------------------------------------------------------------ */
/* A3/A4 shared projection (render.cu:33-40, :62-70)                          */
static inline int project_point(const float *P, float x, float y, float z, int W, int H,
                                uint32_t *pix, float *depth) {
    float rx = f_add(fmaf(P[2], z, fmaf(P[1], y, f_mul(P[0], x))), P[3]);
    float ry = f_add(fmaf(P[6], z, fmaf(P[5], y, f_mul(P[4], x))), P[7]);
    float rz = f_add(fmaf(P[10], z, fmaf(P[9], y, f_mul(P[8], x))), P[11]);
    if (!(rz > 0.0f)) return 0; /* render.cu:63 (also rejects NaN) */
    float inv = f_div(1.0f, rz); /* contract option B: one correctly rounded reciprocal */
    float fu = rintf(f_mul(rx, inv)); /* render.cu:65 */
    float fv = rintf(f_mul(ry, inv)); /* render.cu:66 */
    if (!(fu >= 0.0f && fu < (float)W && fv >= 0.0f && fv < (float)H)) return 0; /* :68 */
    *pix = (uint32_t)((int)fv * W + (int)fu); /* :70 */
    *depth = rz;
    return 1;
}

/* test hook: project one point, report pixel id / depth bits (-1 when culled) */
int64_t orc_project_point(const float P[16], float x, float y, float z, int W, int H,
                          uint32_t *depth_bits) {
    uint32_t pix; float d;
    if (!project_point(P, x, y, z, W, H, &pix, &d)) return -1;
    if (depth_bits) *depth_bits = f2u(d);
    return (int64_t)pix;
}

/* A1 + A2 (render.cu:16-31, project_cloud.cu:316-317) with FULL coverage
 * (reference quirk Q1: its grid misses the last rows when H % 16 != 0).        */
void orc_clear(uint32_t *depth, uint32_t *acc, size_t npix) {
    if (depth) for (size_t i = 0; i < npix; ++i) depth[i] = ORC_EMPTY_DEPTH;
    if (acc) memset(acc, 0, npix * 4 * sizeof(uint32_t));
}

static inline const float *xyz_at(const void *xyz, size_t stride, size_t i) {
    return (const float *)((const char *)xyz + i * stride);
}

/* A4 minDepthPass (render.cu:53-83): atomicMin of every surviving point.       */
void orc_min_depth_pass(const void *xyz, size_t xyz_stride, size_t n, const float P[16], int W,
                        int H, uint32_t *depth) {
    for (size_t i = 0; i < n; ++i) {
        const float *p = xyz_at(xyz, xyz_stride, i);
        uint32_t pix; float d;
        if (!project_point(P, p[0], p[1], p[2], W, H, &pix, &d)) continue;
        uint32_t b = f2u(d);
        if (b < depth[pix]) depth[pix] = b; /* render.cu:81 */
    }
}

/* A5 accumulatePass (render.cu:85-130)                                         */
void orc_accumulate_pass(const void *xyz, size_t xyz_stride, const uint8_t *rgb, size_t rgb_stride,
                         size_t n, const float P[16], int W, int H, const uint32_t *depth,
                         uint32_t *acc, float window) {
    for (size_t i = 0; i < n; ++i) {
        const float *p = xyz_at(xyz, xyz_stride, i);
        uint32_t pix; float d;
        if (!project_point(P, p[0], p[1], p[2], W, H, &pix, &d)) continue;
        float m = u2f(depth[pix]);
        if (d > f_add(m, window)) continue; /* render.cu:106 */
        const uint8_t *c = rgb + i * rgb_stride;
        acc[4 * (size_t)pix + 0] += c[0];
        acc[4 * (size_t)pix + 1] += c[1];
        acc[4 * (size_t)pix + 2] += c[2];
        acc[4 * (size_t)pix + 3] += 1u;
    }
}

/* A6 resolvePass (render.cu:132-163), full coverage (quirk Q2)                 */
void orc_resolve(const uint32_t *acc, size_t npix, uint8_t *img) {
    for (size_t i = 0; i < npix; ++i) {
        uint32_t c = acc[4 * i + 3];
        if (c == 0) { img[3 * i] = img[3 * i + 1] = img[3 * i + 2] = 0; continue; }
        img[3 * i + 0] = (uint8_t)(acc[4 * i + 0] / c);
        img[3 * i + 1] = (uint8_t)(acc[4 * i + 1] / c);
        img[3 * i + 2] = (uint8_t)(acc[4 * i + 2] / c);
    }
}

/* A7 computeRGBDInternal (project_cloud.cu:314-329), naive single-thread host loop */
void orc_project(const void *xyz, size_t xyz_stride, const uint8_t *rgb, size_t rgb_stride,
                 size_t n, const float P[16], int W, int H, const orc_params *prm,
                 uint32_t *depth, uint32_t *acc, uint8_t *img) {
    size_t npix = (size_t)W * H;
    orc_clear(depth, acc, npix);
    orc_min_depth_pass(xyz, xyz_stride, n, P, W, H, depth);
    orc_accumulate_pass(xyz, xyz_stride, rgb, rgb_stride, n, P, W, H, depth, acc, prm->depth_window);
    if (img) orc_resolve(acc, npix, img);
}

/* ------------------------------------------------------------------------ */
/* multi-thread CPU projector (the timed CPU baseline): contiguous point
 * slices, per-thread framebuffers merged by min / integer sum (no atomics).   */
typedef struct {
    const void *xyz; size_t xs; const uint8_t *rgb; size_t rs; size_t lo, hi;
    const float *P; int W, H; float window;
    uint32_t *depth; uint32_t *acc; const uint32_t *gdepth; int phase;
} mt_job;

static void *mt_worker(void *arg) {
    mt_job *j = (mt_job *)arg;
    size_t n = j->hi - j->lo;
    const void *xyz = (const char *)j->xyz + j->lo * j->xs;
    if (j->phase == 0) {
        orc_clear(j->depth, NULL, (size_t)j->W * j->H);
        orc_min_depth_pass(xyz, j->xs, n, j->P, j->W, j->H, j->depth);
    } else {
        memset(j->acc, 0, (size_t)j->W * j->H * 16);
        orc_accumulate_pass(xyz, j->xs, j->rgb + j->lo * j->rs, j->rs, n, j->P, j->W, j->H,
                            j->gdepth, j->acc, j->window);
    }
    return NULL;
}

typedef struct { uint32_t *dst; uint32_t **src; int nsrc; size_t lo, hi; int op; } merge_job;
static void *merge_worker(void *arg) {
    merge_job *m = (merge_job *)arg;
    for (size_t i = m->lo; i < m->hi; ++i) {
        uint32_t v = m->src[0][i];
        if (m->op == 0) { for (int t = 1; t < m->nsrc; ++t) if (m->src[t][i] < v) v = m->src[t][i]; }
        else { for (int t = 1; t < m->nsrc; ++t) v += m->src[t][i]; }
        m->dst[i] = v;
    }
    return NULL;
}

static void run_merge(uint32_t *dst, uint32_t **src, int nsrc, size_t count, int op, int nthreads) {
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * nthreads);
    merge_job *mj = (merge_job *)malloc(sizeof(merge_job) * nthreads);
    for (int t = 0; t < nthreads; ++t) {
        mj[t] = (merge_job){dst, src, nsrc, count * t / nthreads, count * (t + 1) / nthreads, op};
        pthread_create(&th[t], NULL, merge_worker, &mj[t]);
    }
    for (int t = 0; t < nthreads; ++t) pthread_join(th[t], NULL);
    free(th); free(mj);
}

/* scratch: caller provides nthreads * npix * 4 u32 words (reused for both phases) */
int orc_project_mt(const void *xyz, size_t xyz_stride, const uint8_t *rgb, size_t rgb_stride,
                   size_t n, const float P[16], int W, int H, const orc_params *prm, int nthreads,
                   uint32_t *scratch, uint32_t *depth, uint32_t *acc, uint8_t *img) {
    if (nthreads < 1) return -1;
    size_t npix = (size_t)W * H;
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * nthreads);
    mt_job *jobs = (mt_job *)malloc(sizeof(mt_job) * nthreads);
    uint32_t **bufs = (uint32_t **)malloc(sizeof(uint32_t *) * nthreads);
    for (int phase = 0; phase < 2; ++phase) {
        for (int t = 0; t < nthreads; ++t) {
            bufs[t] = scratch + (size_t)t * npix * 4;
            jobs[t] = (mt_job){xyz, xyz_stride, rgb, rgb_stride, n * t / nthreads,
                               n * (t + 1) / nthreads, P, W, H, prm->depth_window,
                               bufs[t], bufs[t], depth, phase};
            pthread_create(&th[t], NULL, mt_worker, &jobs[t]);
        }
        for (int t = 0; t < nthreads; ++t) pthread_join(th[t], NULL);
        if (phase == 0) run_merge(depth, bufs, nthreads, npix, 0, nthreads);
        else run_merge(acc, bufs, nthreads, npix * 4, 1, nthreads);
    }
    if (img) orc_resolve(acc, npix, img);
    free(th); free(jobs); free(bufs);
    return 0;
}

/* ------------------------------------------------------------------------ */
/* ENVELOPE of the unpinned arithmetic (tests/test_oracle_envelope.py, tools/oracle_envelope.py).
 * The reference's projection is not reproducible off an NVIDIA toolchain in two places:
 *   render.cu:33-40  matmul: `m[0]*v.x + m[1]*v.y + m[2]*v.z + m[3]` under nvcc's default -fmad=true -- which of
 *                    the products are contracted into the following add is the compiler's choice;
 *   render.cu:65-66  __fdividef(r.x, r.z): x times an APPROXIMATE reciprocal (<= 2 ulp by the CUDA documentation).
 * The variants below are the evaluations a CUDA build could plausibly produce; they exist only to MEASURE how far
 * the contract (mm = 0, dv = 0) can be from them -- nothing in the product or in the parity tests uses them.
 *   mm 0  contract: t = m0 x; t = fma(m1, y, t); t = fma(m2, z, t); r = t + m3        (left to right, right products fused)
 *   mm 1  the first add fused with the LEFT product: t = m1 y; t = fma(m0, x, t); t = fma(m2, z, t); r = t + m3
 *   mm 2  nothing contracted (-fmad=false): ((m0 x + m1 y) + m2 z) + m3
 *   mm 3  re-associated, all fused right to left: fma(m0, x, fma(m1, y, fma(m2, z, m3)))
 *   mm 4  re-associated, m3 folded into the first fma: fma(m2, z, fma(m1, y, fma(m0, x, m3)))
 *   dv 0  contract: x * RN(1 / z)      dv 1  IEEE x / z      dv 2..5  x * (RN(1 / z) -2, -1, +1, +2 ulp)            */
static inline float row_v(const float *m, float x, float y, float z, int mm) {
    switch (mm) {
    case 1: return f_add(fmaf(m[2], z, fmaf(m[0], x, f_mul(m[1], y))), m[3]);
    case 2: return f_add(f_add(f_add(f_mul(m[0], x), f_mul(m[1], y)), f_mul(m[2], z)), m[3]);
    case 3: return fmaf(m[0], x, fmaf(m[1], y, fmaf(m[2], z, m[3])));
    case 4: return fmaf(m[2], z, fmaf(m[1], y, fmaf(m[0], x, m[3])));
    default: return f_add(fmaf(m[2], z, fmaf(m[1], y, f_mul(m[0], x))), m[3]);
    }
}
static inline float quot_v(float a, float rz, int dv) {
    if (dv == 1) return f_div(a, rz);
    float inv = f_div(1.0f, rz);
    if (dv >= 2) {
        static const int k[4] = {-2, -1, 1, 2};
        uint32_t b = f2u(inv);
        if (b > 0x00800002u && b < 0x7F7FFFFDu) inv = u2f(b + (uint32_t)k[dv - 2]); /* positive normal: +- k ulp */
    }
    return f_mul(a, inv);
}
static inline int project_point_v(const float *P, float x, float y, float z, int W, int H, int mm, int dv,
                                  uint32_t *pix, float *depth) {
    float rx = row_v(P, x, y, z, mm), ry = row_v(P + 4, x, y, z, mm), rz = row_v(P + 8, x, y, z, mm);
    if (!(rz > 0.0f)) return 0;
    float fu = rintf(quot_v(rx, rz, dv)), fv = rintf(quot_v(ry, rz, dv));
    if (!(fu >= 0.0f && fu < (float)W && fv >= 0.0f && fv < (float)H)) return 0;
    *pix = (uint32_t)((int)fv * W + (int)fu);
    *depth = rz;
    return 1;
}
/* Per-point comparison of a variant with the contract over one slice of a cloud.
 * out[0] points the contract accepts, out[1] points either accepts, out[2] points whose acceptance or pixel differs,
 * out[3] largest |depth bits difference| (ulp) among the points both accept.                                         */
typedef struct { const void *xyz; size_t xs, lo, hi; const float *P; int W, H, mm, dv; uint64_t out[4]; } env_job;
static void *env_worker(void *arg) {
    env_job *j = (env_job *)arg;
    uint64_t acc0 = 0, either = 0, flips = 0, ulp = 0;
    for (size_t i = j->lo; i < j->hi; ++i) {
        const float *p = xyz_at(j->xyz, j->xs, i);
        uint32_t pa = 0, pb = 0; float da = 0.f, db = 0.f;
        const int a = project_point(j->P, p[0], p[1], p[2], j->W, j->H, &pa, &da);
        const int b = project_point_v(j->P, p[0], p[1], p[2], j->W, j->H, j->mm, j->dv, &pb, &db);
        acc0 += (uint64_t)a;
        either += (uint64_t)(a | b);
        if (a != b || (a && pa != pb)) flips += 1;
        if (a && b) {
            const uint32_t ua = f2u(da), ub = f2u(db);
            const uint64_t d = ua > ub ? ua - ub : ub - ua;
            if (d > ulp) ulp = d;
        }
    }
    j->out[0] = acc0; j->out[1] = either; j->out[2] = flips; j->out[3] = ulp;
    return NULL;
}
int orc_envelope_points(const void *xyz, size_t xyz_stride, size_t n, const float P[16], int W, int H, int mm, int dv,
                        int nthreads, uint64_t out[4]) {
    if (nthreads < 1 || mm < 0 || mm > 4 || dv < 0 || dv > 5) return -1;
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * nthreads);
    env_job *jobs = (env_job *)malloc(sizeof(env_job) * nthreads);
    for (int t = 0; t < nthreads; ++t) {
        jobs[t] = (env_job){xyz, xyz_stride, n * t / nthreads, n * (t + 1) / nthreads, P, W, H, mm, dv, {0, 0, 0, 0}};
        pthread_create(&th[t], NULL, env_worker, &jobs[t]);
    }
    out[0] = out[1] = out[2] = out[3] = 0;
    for (int t = 0; t < nthreads; ++t) {
        pthread_join(th[t], NULL);
        out[0] += jobs[t].out[0]; out[1] += jobs[t].out[1]; out[2] += jobs[t].out[2];
        if (jobs[t].out[3] > out[3]) out[3] = jobs[t].out[3];
    }
    free(th); free(jobs);
    return 0;
}
/* A whole frame (A1..A6, single thread) under a variant: what the frame buffers of such a build would hold.           */
int orc_project_variant(const void *xyz, size_t xyz_stride, const uint8_t *rgb, size_t rgb_stride, size_t n,
                        const float P[16], int W, int H, const orc_params *prm, int mm, int dv,
                        uint32_t *depth, uint32_t *acc, uint8_t *img) {
    if (mm < 0 || mm > 4 || dv < 0 || dv > 5) return -1;
    const size_t npix = (size_t)W * H;
    orc_clear(depth, acc, npix);
    for (size_t i = 0; i < n; ++i) {
        const float *p = xyz_at(xyz, xyz_stride, i);
        uint32_t pix; float d;
        if (!project_point_v(P, p[0], p[1], p[2], W, H, mm, dv, &pix, &d)) continue;
        const uint32_t b = f2u(d);
        if (b < depth[pix]) depth[pix] = b;
    }
    for (size_t i = 0; i < n; ++i) {
        const float *p = xyz_at(xyz, xyz_stride, i);
        uint32_t pix; float d;
        if (!project_point_v(P, p[0], p[1], p[2], W, H, mm, dv, &pix, &d)) continue;
        if (d > f_add(u2f(depth[pix]), prm->depth_window)) continue;
        const uint8_t *c = rgb + i * rgb_stride;
        acc[4 * (size_t)pix + 0] += c[0];
        acc[4 * (size_t)pix + 1] += c[1];
        acc[4 * (size_t)pix + 2] += c[2];
        acc[4 * (size_t)pix + 3] += 1u;
    }
    if (img) orc_resolve(acc, npix, img);
    return 0;
}

/* ------------------------------------------------------------------------ */
/* depth-heuristic prefilter                                                  */

/* A8 reduce (project_cloud.cu:28-53): 2x2 min-pool, source row stride 2*w */
static void f_reduce(const float *hi, float *lo, int w, int h) {
    int ws = w * 2;
    for (int idx = 0; idx < w * h; ++idx) {
        int x = idx % w, y = idx / w;
        float p0 = hi[(2 * y) * ws + 2 * x], p1 = hi[(2 * y) * ws + 2 * x + 1];
        float p2 = hi[(2 * y + 1) * ws + 2 * x], p3 = hi[(2 * y + 1) * ws + 2 * x + 1];
        float l0 = p0 < p1 ? p0 : p1;
        float l1 = p2 < p3 ? p2 : p3;
        lo[idx] = l0 < l1 ? l0 : l1;
    }
}

/* A9 laplacianKernel (project_cloud.cu:55-79) */
static void f_laplacian(const float *in, uint8_t *out, int w, int h, float thr) {
    static const int k[9] = {0, 1, 0, 1, -4, 1, 0, 1, 0}; /* project_cloud.cu:26 */
    for (int idx = 0; idx < w * h; ++idx) {
        int x = idx % w, y = idx / w;
        if (x == 0 || x == w - 1 || y == 0 || y == h - 1) { out[idx] = 0; continue; }
        float sum = 0.0f;
        int c = 0;
        for (int ky = -1; ky <= 1; ++ky)
            for (int kx = -1; kx <= 1; ++kx) {
                sum = fmaf(in[(y + ky) * w + (x + kx)], (float)k[c], sum); /* :73 */
                ++c;
            }
        out[idx] = (sum > thr) ? 255 : 0; /* :77 */
    }
}

/* project_cloud.cu:81-86 */
static inline float f_pixel(const float *lo, int x, int y, int w, int h) {
    if (x >= 0 && x < w && y >= 0 && y < h) return lo[y * w + x];
    return -1.0f;
}

/* A10 compareImgsKernel (project_cloud.cu:88-126) */
static void f_compare(const float *lo, const float *hi, const uint8_t *grad, uint8_t *mask,
                      int hw, int hh, float strength) {
    int lw = hw / 2, lh = hh / 2;
    for (int idx = 0; idx < hw * hh; ++idx) {
        int x = idx % hw, y = idx / hw;
        float cur = hi[idx];
        if ((double)cur >= 3.4028e38) { mask[idx] = 0; continue; } /* :97, MAX_FLOAT :21 */
        int lx = x / 2, ly = y / 2;
        int keep = 0;
        if (grad[ly * lw + lx] > 0) {
            for (int dx = -1; dx <= 1 && !keep; ++dx)
                for (int dy = -1; dy <= 1 && !keep; ++dy)
                    if (cur <= f_mul(f_pixel(lo, lx + dx, ly + dy, lw, lh), strength)) keep = 1;
        } else if (cur <= f_mul(f_pixel(lo, lx, ly, lw, lh), strength)) {
            keep = 1;
        }
        mask[idx] = keep ? 255 : 0;
    }
}

/* A11 resizeKernel (project_cloud.cu:128-161), in place into the finer level */
static void f_resize(const float *lo, float *hi, const uint8_t *mask, int ow, int oh) {
    int lw = ow / 2, lh = oh / 2;
    for (int idx = 0; idx < ow * oh; ++idx) {
        if (mask[idx] > 0) continue;
        int x = idx % ow, y = idx / ow;
        float inX = f_sub(f_div(f_add((float)x, 0.5f), 2.0f), 0.5f);
        float inY = f_sub(f_div(f_add((float)y, 0.5f), 2.0f), 0.5f);
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
}

/* A12 (render.cu:168-240): min/max of u32 bit patterns, skipping the sentinel */
static void depth_minmax(const uint32_t *d, size_t n, uint32_t *mn, uint32_t *mx) {
    uint32_t lo = 0xFFFFFFFFu, hi = 0u;
    for (size_t i = 0; i < n; ++i) {
        uint32_t v = d[i];
        if (v == ORC_EMPTY_DEPTH) continue;
        if (v < lo) lo = v;
        if (v > hi) hi = v;
    }
    *mn = lo; *mx = hi;
}

/* A13 removeMask for one pixel (project_cloud.cu:163-187).  `plane` is the
 * tensor plane stride: W*H here (the reference uses W*H_eff, quirk Q3).        */
static inline void remove_mask_px(float *depth, uint8_t *img, uint8_t m, uint16_t *tensor,
                                  size_t plane, size_t idx, float mn, float range) {
    if (m == 0) {
        depth[idx] = -1.0f;
        img[3 * idx] = img[3 * idx + 1] = img[3 * idx + 2] = 0;
        if (tensor) {
            tensor[0 * plane + idx] = 0; tensor[1 * plane + idx] = 0;
            tensor[2 * plane + idx] = 0; tensor[3 * plane + idx] = 0;
            tensor[4 * plane + idx] = 0xBC00u; /* half(-1) */
        }
        return;
    }
    if (!tensor) return;
    for (int k = 0; k < 3; ++k) {
        float hv = f16_to_f32(f32_to_f16((float)img[3 * idx + k])); /* (c10::Half)u8 */
        tensor[k * plane + idx] = f32_to_f16(f_div(hv, 255.0f));    /* :181-183 */
    }
    tensor[3 * plane + idx] = f32_to_f16(f_div(f16_to_f32(f32_to_f16((float)m)), 255.0f)); /* :184 */
    float hd = f16_to_f32(f32_to_f16(f_sub(depth[idx], mn)));       /* :185 cast binds first */
    tensor[4 * plane + idx] = f32_to_f16(f_div(hd, range));
}

/* A14 applyDepthFilter (project_cloud.cu:331-392).
 *  depth : W*H u32 float bits, in/out (masked -> -1.0f)
 *  img   : W*H*3 u8, in/out (masked -> 0)
 *  mask  : W*H u8 out (may be NULL);  tensor: 5*W*H fp16 bits out (may be NULL)
 *  minmax: 2 u32 out (may be NULL)
 * Rows >= H_eff = (H >> levels) << levels are outside the reference's filter
 * domain (quirk Q3): they skip the pyramid test, mask = non-empty ? 255 : 0,
 * and go through the same removeMask step.  Requires W % 2^levels == 0.
 * Returns 0, or -1 on unsupported dimensions.                                 */
int orc_filter(uint32_t *depth_bits, uint8_t *img, int W, int H, const orc_params *prm,
               uint8_t *mask_out, uint16_t *tensor, uint32_t *minmax_out) {
    int L = prm->levels;
    if (L < 1 || L > 8 || W <= 0 || H <= 0 || (W % (1 << L)) != 0 || (H >> L) < 1) return -1;
    float *lv[9]; int w[9], h[9];
    lv[0] = (float *)depth_bits; w[0] = W; h[0] = H;
    for (int i = 1; i <= L; ++i) {
        w[i] = w[i - 1] / 2; h[i] = h[i - 1] / 2;
        lv[i] = (float *)malloc(sizeof(float) * w[i] * h[i]);
        f_reduce(lv[i - 1], lv[i], w[i], h[i]); /* :348 */
    }
    int cw = w[L], ch = h[L];
    size_t npix = (size_t)W * H;
    uint8_t *mask = (uint8_t *)malloc(npix);
    for (int i = L; i >= 1; --i) {
        uint8_t *grad = (uint8_t *)malloc((size_t)cw * ch);
        f_laplacian(lv[i], grad, cw, ch, prm->gradient_threshold); /* :357 */
        cw *= 2; ch *= 2;                                          /* :360-361 */
        f_compare(lv[i], lv[i - 1], grad, mask, cw, ch, prm->filter_strength); /* :367 */
        free(grad);
        if (i == 1) {
            uint32_t mn, mx;
            depth_minmax(depth_bits, (size_t)cw * ch, &mn, &mx); /* :375-377 */
            if (minmax_out) { minmax_out[0] = mn; minmax_out[1] = mx; }
            float fmn = u2f(mn), range = f_sub(u2f(mx), u2f(mn));
            for (size_t idx = (size_t)cw * ch; idx < npix; ++idx)
                mask[idx] = ((double)lv[0][idx] >= 3.4028e38) ? 0 : 255; /* same empty test as :97 */
            for (size_t idx = 0; idx < npix; ++idx) /* :380 */
                remove_mask_px(lv[0], img, mask[idx], tensor, npix, idx, fmn, range);
        } else {
            f_resize(lv[i], lv[i - 1], mask, cw, ch); /* :385 */
        }
        free(lv[i]);
    }
    if (mask_out) memcpy(mask_out, mask, npix);
    free(mask);
    return 0;
}

/* ------------------------------------------------------------------------ */
/* deterministic synthetic clouds; counter-based so every shard
 * can be generated independently.  Output AoS: xyzw float4 (w = 1) + rgba u8.  */
static inline uint64_t mix64(uint64_t z) {
    z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull;
    z ^= z >> 27; z *= 0x94D049BB133111EBull;
    z ^= z >> 31; return z;
}
static inline uint64_t hsh(uint64_t seed, uint64_t i, uint64_t k) {
    return mix64(seed + (4ull * i + k) * 0x9E3779B97F4A7C15ull);
}
static inline float u01(uint64_t h) { return (float)(h >> 40) * 0x1p-24f; }

static inline uint32_t compact1by1(uint64_t v) { /* even bits of v */
    v &= 0x5555555555555555ull;
    v = (v | (v >> 1)) & 0x3333333333333333ull;
    v = (v | (v >> 2)) & 0x0F0F0F0F0F0F0F0Full;
    v = (v | (v >> 4)) & 0x00FF00FF00FF00FFull;
    v = (v | (v >> 8)) & 0x0000FFFF0000FFFFull;
    v = (v | (v >> 16)) & 0x00000000FFFFFFFFull;
    return (uint32_t)v;
}

/* room_shell surfaces: 6 box faces (x in [-4,4], y in [-1.5,1.5], z in [-4,4])
 * and 8 spheres r = 0.5 at (+-2, +-0.75, +-2); weights ~ area, sum 248.       */
static const uint32_t RS_W[14] = {24, 24, 64, 64, 24, 24, 3, 3, 3, 3, 3, 3, 3, 3};

static void room_shell_point(uint64_t seed, uint64_t i, uint64_t total, float *p) {
    uint64_t start = 0, cnt = 0; int s = 0;
    for (s = 0; s < 14; ++s) {
        cnt = (s == 13) ? (total - start) : (total * RS_W[s]) / 248ull;
        if (i < start + cnt || s == 13) break;
        start += cnt;
    }
    uint64_t j = i - start;
    int b = 0;
    while (b < 15 && (1ull << (2 * (b + 1))) <= cnt) ++b; /* 4^b <= cnt */
    uint64_t M = 1ull << (2 * b);
    uint64_t cell = (cnt > 0) ? (j * M) / cnt : 0;
    float cx = (float)compact1by1(cell), cy = (float)compact1by1(cell >> 1);
    float scale = u2f((uint32_t)(127 - b) << 23); /* 2^-b */
    float sp = f_mul(f_add(cx, u01(hsh(seed, i, 0))), scale);
    float tp = f_mul(f_add(cy, u01(hsh(seed, i, 1))), scale);
    if (s < 6) {
        float a8 = f_add(-4.0f, f_mul(sp, 8.0f)), b8 = f_add(-4.0f, f_mul(tp, 8.0f));
        float a3 = f_add(-1.5f, f_mul(sp, 3.0f)), b3 = f_add(-1.5f, f_mul(tp, 3.0f));
        switch (s) {
            case 0: p[0] = -4.0f; p[1] = a3; p[2] = b8; break;
            case 1: p[0] = 4.0f;  p[1] = a3; p[2] = b8; break;
            case 2: p[0] = a8; p[1] = -1.5f; p[2] = b8; break;
            case 3: p[0] = a8; p[1] = 1.5f;  p[2] = b8; break;
            case 4: p[0] = a8; p[1] = b3; p[2] = -4.0f; break;
            default: p[0] = a8; p[1] = b3; p[2] = 4.0f; break;
        }
    } else {
        int q = s - 6;
        float cxs = (q & 1) ? 2.0f : -2.0f, cys = (q & 2) ? 0.75f : -0.75f, czs = (q & 4) ? 2.0f : -2.0f;
        /* octahedral map of (sp,tp) to the unit sphere, no transcendentals */
        float a = f_sub(f_mul(2.0f, sp), 1.0f), bb = f_sub(f_mul(2.0f, tp), 1.0f);
        float aa = fabsf(a), ab = fabsf(bb);
        float vz = f_sub(f_sub(1.0f, aa), ab);
        float vx = a, vy = bb;
        if (vz < 0.0f) {
            vx = copysignf(f_sub(1.0f, ab), a);
            vy = copysignf(f_sub(1.0f, aa), bb);
        }
        float len = sqrtf(fmaf(vz, vz, fmaf(vy, vy, f_mul(vx, vx))));
        float k = f_div(0.5f, len);
        p[0] = fmaf(vx, k, cxs); p[1] = fmaf(vy, k, cys); p[2] = fmaf(vz, k, czs);
    }
}

/* scene 0 = uniform_box, 1 = room_shell.  Generates indices [first, first+count). */
int orc_generate(int scene, uint64_t seed, uint64_t first, uint64_t count, uint64_t total,
                 float *xyzw, uint8_t *rgba) {
    if (scene < 0 || scene > 1 || first + count > total) return -1;
    for (uint64_t t = 0; t < count; ++t) {
        uint64_t i = first + t;
        float *p = xyzw + 4 * t;
        if (scene == 0) {
            p[0] = f_add(-4.0f, f_mul(u01(hsh(seed, i, 0)), 8.0f));
            p[1] = f_add(-1.5f, f_mul(u01(hsh(seed, i, 1)), 3.0f));
            p[2] = f_add(-4.0f, f_mul(u01(hsh(seed, i, 2)), 8.0f));
        } else {
            room_shell_point(seed, i, total, p);
        }
        p[3] = 1.0f;
        uint64_t hc = hsh(seed, i, 3);
        rgba[4 * t + 0] = (uint8_t)(hc & 0xFF);
        rgba[4 * t + 1] = (uint8_t)((hc >> 8) & 0xFF);
        rgba[4 * t + 2] = (uint8_t)((hc >> 16) & 0xFF);
        rgba[4 * t + 3] = 255;
    }
    return 0;
}
