// Host side of libmrgingham_amd.so: the context (streams + scratch), the batch
// API and the reference's own C symbols as thin wrappers over it.
// See include/mrgingham_amd.h for the contract of every entry point.
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "../../include/mrgingham_amd.h"
#include "common.h"
#include "kernels.h"

namespace mrg {

constexpr int kMaxStreams = 8;

struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
};

}  // namespace mrg

struct mrgingham_amd_ctx {
    int device = 0;
    int nstreams = 4;
    hipStream_t streams[mrg::kMaxStreams] = {};
    std::string err;
    int cap_shift = 3;        // hot-pixel table capacity = pixels >> cap_shift per frame
    bool use_v0 = false;      // reference-shaped ChESS kernel instead of the tuned one
    bool async_error = false; // a queued batch could not be issued

    // scratch
    mrg::DevBuf level_img, resp, lidx, hot_cnt, status, hot_pix, parent, comp_cnt, roots, comp_box, arena, cand,
        sortkeys, leader, need, nseeds, seeds, cand_xy, cand_counts, io_frame, io_out;
    // what the scratch was sized for
    int active_nframes = 0;   // frames of the batches queued since the last sync
    int s_nframes = 0, s_w = 0, s_h = 0, s_pitch = 0, s_cap = 0, s_cand_cap = 0, s_sort_cap = 0, s_shift = -1;
    long long s_arena_cap = 0;

    // dominant-kernel timing
    bool timing = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> events;
    std::vector<hipEvent_t> event_pool;
    std::vector<int32_t> host_status;
};

namespace mrg {

static int fail(mrgingham_amd_ctx* ctx, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (ctx) ctx->err = buf;
    fprintf(stderr, "mrgingham_amd: %s\n", buf);
    return code;
}

int fail_hip(mrgingham_amd_ctx* ctx, hipError_t e, const char* what, const char* file, int line) {
    return fail(ctx, MRGINGHAM_AMD_ERR_DEVICE, "%s:%d: %s failed: %s", file, line, what, hipGetErrorString(e));
}

static int ensure(mrgingham_amd_ctx* ctx, DevBuf& b, size_t bytes) {
    if (bytes <= b.bytes) return 0;
    if (b.p) {
        MRG_HIP_CHECK(hipDeviceSynchronize());
        MRG_HIP_CHECK(hipFree(b.p));
        b.p = nullptr;
        b.bytes = 0;
    }
    const size_t want = bytes + bytes / 8 + 256;
    MRG_HIP_CHECK(hipMalloc(&b.p, want));
    b.bytes = want;
    return 0;
}

static int level_dims(int W, int H, int level, int* w, int* h) {
    if (level < 0 || level > 10) return -1;  // find_chessboard_corners.cc:433-441
    auto rnd = [level](int v) {              // cvRound(v / 2^level): ties to even
        const int s = 1 << level;
        int q = v >> level;
        const int rem = v & (s - 1), half = s >> 1;
        if (level > 0 && (rem > half || (rem == half && (q & 1)))) ++q;
        return q;
    };
    *w = rnd(W);
    *h = rnd(H);
    return 0;
}

static int validate_frames(mrgingham_amd_ctx* ctx, const mrgingham_amd_frames* f) {
    if (!ctx) return MRGINGHAM_AMD_ERR_ARG;
    if (!f || (!f->frames && f->nframes > 0) || f->nframes < 0 || f->width < 0 || f->height < 0 ||
        f->stride < f->width)
        return fail(ctx, MRGINGHAM_AMD_ERR_ARG, "bad frame batch descriptor");
    if (f->width > 32767 || f->height > 32767)  // int16 coordinates, find_chessboard_corners.cc:91
        return fail(ctx, MRGINGHAM_AMD_ERR_ARG, "frames larger than 32767 pixels per side are not supported");
    return 0;
}

// Scratch for a batch of nframes W x H frames with up to `pitch` points per frame.
static int ensure_scratch(mrgingham_amd_ctx* ctx, int nframes, int W, int H, int pitch) {
    if (nframes <= ctx->s_nframes && W == ctx->s_w && H == ctx->s_h && pitch <= ctx->s_pitch &&
        ctx->s_shift == ctx->cap_shift)
        return 0;
    nframes = nframes > ctx->s_nframes ? nframes : ctx->s_nframes;
    pitch = pitch > ctx->s_pitch ? pitch : ctx->s_pitch;
    const long long px = (long long)W * H;
    long long cap = px >> ctx->cap_shift;
    if (cap < 4096) cap = 4096;
    if (cap > px) cap = px > 0 ? px : 1;
    if (cap > 0x3fffffff) return fail(ctx, MRGINGHAM_AMD_ERR_ARG, "frame too large");
    long long cand_cap = cap / 2 + 1;
    if (cand_cap < pitch) cand_cap = pitch;
    long long sort_cap = 1;
    while (sort_cap < cand_cap) sort_cap <<= 1;
    const long long arena_cap = 5 * cap + 16LL * (pitch > 1024 ? pitch : 1024);
    int w1 = 0, h1 = 0;
    level_dims(W, H, 1, &w1, &h1);
    const size_t nf = (size_t)nframes;
    int rc = 0;
    if ((rc = ensure(ctx, ctx->level_img, nf * (size_t)w1 * (size_t)h1 + 16))) return rc;
    if ((rc = ensure(ctx, ctx->resp, nf * (size_t)px * 2 + 16))) return rc;
    if ((rc = ensure(ctx, ctx->lidx, nf * (size_t)px * 4 + 16))) return rc;
    if (nf * 4 > ctx->status.bytes) {
        if ((rc = ensure(ctx, ctx->hot_cnt, nf * 4))) return rc;
        if ((rc = ensure(ctx, ctx->status, nf * 4))) return rc;
        MRG_HIP_CHECK(hipMemset(ctx->hot_cnt.p, 0, ctx->hot_cnt.bytes));
        MRG_HIP_CHECK(hipMemset(ctx->status.p, 0, ctx->status.bytes));
    }
    if ((rc = ensure(ctx, ctx->hot_pix, nf * (size_t)cap * 4))) return rc;
    if ((rc = ensure(ctx, ctx->parent, nf * (size_t)cap * 4))) return rc;
    if ((rc = ensure(ctx, ctx->comp_cnt, nf * (size_t)cap * 4))) return rc;
    if ((rc = ensure(ctx, ctx->roots, nf * (size_t)cap * 4))) return rc;
    if ((rc = ensure(ctx, ctx->comp_box, nf * (size_t)cap * 16))) return rc;
    if ((rc = ensure(ctx, ctx->arena, nf * (size_t)arena_cap * 4))) return rc;
    if ((rc = ensure(ctx, ctx->cand, nf * (size_t)cand_cap * sizeof(Cand)))) return rc;
    if ((rc = ensure(ctx, ctx->sortkeys, nf * (size_t)sort_cap * 8))) return rc;
    const size_t np = nf * (size_t)(pitch > 0 ? pitch : 1);
    if ((rc = ensure(ctx, ctx->leader, np * 4))) return rc;
    if ((rc = ensure(ctx, ctx->need, np * 4))) return rc;
    if ((rc = ensure(ctx, ctx->nseeds, np * 4))) return rc;
    if ((rc = ensure(ctx, ctx->seeds, np * 9 * 4))) return rc;
    if ((rc = ensure(ctx, ctx->cand_xy, np * 8))) return rc;
    if ((rc = ensure(ctx, ctx->cand_counts, nf * 4))) return rc;
    ctx->s_nframes = nframes;
    ctx->s_w = W;
    ctx->s_h = H;
    ctx->s_pitch = pitch;
    ctx->s_cap = (int)cap;
    ctx->s_cand_cap = (int)cand_cap;
    ctx->s_sort_cap = (int)sort_cap;
    ctx->s_arena_cap = arena_cap;
    ctx->s_shift = ctx->cap_shift;
    return 0;
}

static CompTables tables_of(mrgingham_amd_ctx* ctx) {
    CompTables t;
    t.cap = ctx->s_cap;
    t.hot_cnt = (int32_t*)ctx->hot_cnt.p;
    t.hot_pix = (int32_t*)ctx->hot_pix.p;
    t.parent = (int32_t*)ctx->parent.p;
    t.comp_cnt = (int32_t*)ctx->comp_cnt.p;
    t.comp_box = (int4*)ctx->comp_box.p;
    t.roots = (int32_t*)ctx->roots.p;
    t.lidx = (int32_t*)ctx->lidx.p;
    t.lidx_pitch = (long long)ctx->s_w * ctx->s_h;
    t.arena = (uint32_t*)ctx->arena.p;
    t.arena_cap = ctx->s_arena_cap;
    t.cand_cap = ctx->s_cand_cap;
    t.cand = (Cand*)ctx->cand.p;
    t.sortkeys = (unsigned long long*)ctx->sortkeys.p;
    t.sort_cap = ctx->s_sort_cap;
    t.status = (int32_t*)ctx->status.p;
    return t;
}

// Level image + response of frames [f0, f0+n) at `level`, queued on `s`.
// Returns the LevelBatch the later kernels of the level use.
static LevelBatch queue_level_response(mrgingham_amd_ctx* ctx, const mrgingham_amd_frames* fr, int level, int f0,
                                       int n, bool clamp, bool hot, int16_t* resp_override, hipStream_t s,
                                       bool time_it) {
    LevelBatch lb;
    int w, h;
    level_dims(fr->width, fr->height, level, &w, &h);
    lb.nframes = fr->nframes;
    lb.w = w;
    lb.h = h;
    if (level == 0) {
        lb.img = fr->frames;
        lb.img_pitch = fr->frame_pitch;
        lb.img_stride = fr->stride;
    } else {
        FrameBatch fb{fr->frames, fr->frame_pitch, fr->width, fr->height, fr->stride};
        // scratch slices are per frame and level-independent: chunks on different
        // streams may be at different levels at the same time
        int w1, h1;
        level_dims(fr->width, fr->height, 1, &w1, &h1);
        const long long pitch = (long long)w1 * h1;
        launch_decimate(fb, level, (uint8_t*)ctx->level_img.p, pitch, w, h, f0, n, s);
        lb.img = (const uint8_t*)ctx->level_img.p;
        lb.img_pitch = pitch;
        lb.img_stride = w;
    }
    lb.resp = resp_override ? resp_override : (int16_t*)ctx->resp.p;
    lb.resp_pitch = resp_override ? (long long)w * h : (long long)fr->width * fr->height;
    CompTables t = tables_of(ctx);
    if (hot) {
        hipMemsetAsync((int32_t*)ctx->hot_cnt.p + f0, 0, sizeof(int32_t) * n, s);
        hipMemsetAsync((int32_t*)ctx->status.p + f0, 0, sizeof(int32_t) * n, s);
    }
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (time_it && ctx->timing) {
        auto get = [ctx]() {
            hipEvent_t e;
            if (!ctx->event_pool.empty()) { e = ctx->event_pool.back(); ctx->event_pool.pop_back(); }
            else hipEventCreate(&e);
            return e;
        };
        e0 = get();
        e1 = get();
        hipEventRecord(e0, s);
    }
    if (w > 0 && h > 0) {
        if (ctx->use_v0) launch_chess_v0(lb, t, f0, n, clamp, hot, s);
        else launch_chess(lb, t, f0, n, clamp, hot, s);
    }
    if (e0) {
        hipEventRecord(e1, s);
        ctx->events.emplace_back(e0, e1);
    }
    return lb;
}

static void chunk_of(const mrgingham_amd_ctx* ctx, int nframes, int c, int* f0, int* n) {
    const int per = (nframes + ctx->nstreams - 1) / ctx->nstreams;
    *f0 = c * per;
    int e = *f0 + per;
    if (e > nframes) e = nframes;
    *n = e > *f0 ? e - *f0 : 0;
}

}  // namespace mrg

using namespace mrg;

extern "C" {

int mrgingham_amd_abi_version(void) { return MRGINGHAM_AMD_ABI_VERSION; }

int mrgingham_amd_level_dims(int width, int height, int level, int* w, int* h) {
    if (!w || !h || width < 0 || height < 0) return MRGINGHAM_AMD_ERR_ARG;
    return level_dims(width, height, level, w, h) == 0 ? MRGINGHAM_AMD_OK : MRGINGHAM_AMD_ERR_ARG;
}

mrgingham_amd_ctx* mrgingham_amd_create(int device_ordinal) {
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) {
        fprintf(stderr, "mrgingham_amd: no usable HIP device (%s); this library has no CPU path\n",
                e != hipSuccess ? hipGetErrorString(e) : "device count is 0");
        return nullptr;
    }
    if (device_ordinal < 0 || device_ordinal >= ndev) {
        fprintf(stderr, "mrgingham_amd: device ordinal %d out of range (0..%d)\n", device_ordinal, ndev - 1);
        return nullptr;
    }
    if (hipSetDevice(device_ordinal) != hipSuccess) return nullptr;
    mrgingham_amd_ctx* ctx = new mrgingham_amd_ctx();
    ctx->device = device_ordinal;
    const char* ns = getenv("MRGINGHAM_AMD_STREAMS");
    if (ns) {
        const int v = atoi(ns);
        if (v >= 1 && v <= kMaxStreams) ctx->nstreams = v;
    }
    const char* v0 = getenv("MRGINGHAM_AMD_CHESS_V0");
    ctx->use_v0 = v0 && atoi(v0) != 0;
    for (int i = 0; i < ctx->nstreams; ++i)
        if (hipStreamCreateWithFlags(&ctx->streams[i], hipStreamNonBlocking) != hipSuccess) {
            fprintf(stderr, "mrgingham_amd: hipStreamCreate failed\n");
            delete ctx;
            return nullptr;
        }
    return ctx;
}

void mrgingham_amd_destroy(mrgingham_amd_ctx* ctx) {
    if (!ctx) return;
    hipSetDevice(ctx->device);
    hipDeviceSynchronize();
    DevBuf* bufs[] = {&ctx->level_img, &ctx->resp, &ctx->lidx, &ctx->hot_cnt, &ctx->status, &ctx->hot_pix,
                      &ctx->parent, &ctx->comp_cnt, &ctx->roots, &ctx->comp_box, &ctx->arena, &ctx->cand,
                      &ctx->sortkeys, &ctx->leader, &ctx->need, &ctx->nseeds, &ctx->seeds, &ctx->cand_xy,
                      &ctx->cand_counts, &ctx->io_frame, &ctx->io_out};
    for (DevBuf* b : bufs)
        if (b->p) hipFree(b->p);
    for (auto& pr : ctx->events) { hipEventDestroy(pr.first); hipEventDestroy(pr.second); }
    for (auto e : ctx->event_pool) hipEventDestroy(e);
    for (int i = 0; i < ctx->nstreams; ++i)
        if (ctx->streams[i]) hipStreamDestroy(ctx->streams[i]);
    delete ctx;
}

const char* mrgingham_amd_last_error(const mrgingham_amd_ctx* ctx) { return ctx ? ctx->err.c_str() : "no context"; }

void mrgingham_amd_set_kernel_timing(mrgingham_amd_ctx* ctx, int enable) {
    if (ctx) ctx->timing = enable != 0;
}

/* tunables (not part of the reference surface) */
int mrgingham_amd_set_option(mrgingham_amd_ctx* ctx, const char* name, int value) {
    if (!ctx || !name) return MRGINGHAM_AMD_ERR_ARG;
    if (!strcmp(name, "hot_capacity_shift")) {
        if (value < 0 || value > 8) return MRGINGHAM_AMD_ERR_ARG;
        ctx->cap_shift = value;
        return 0;
    }
    if (!strcmp(name, "chess_v0")) { ctx->use_v0 = value != 0; return 0; }
    if (!strcmp(name, "streams")) {
        if (value < 1 || value > kMaxStreams) return MRGINGHAM_AMD_ERR_ARG;
        hipSetDevice(ctx->device);
        for (int i = ctx->nstreams; i < value; ++i)
            if (!ctx->streams[i] && hipStreamCreateWithFlags(&ctx->streams[i], hipStreamNonBlocking) != hipSuccess)
                return MRGINGHAM_AMD_ERR_DEVICE;
        ctx->nstreams = value;
        return 0;
    }
    return MRGINGHAM_AMD_ERR_ARG;
}

double mrgingham_amd_chess_kernel_ms(mrgingham_amd_ctx* ctx, int* nlaunches) {
    if (nlaunches) *nlaunches = 0;
    if (!ctx) return 0.;
    hipSetDevice(ctx->device);
    hipDeviceSynchronize();
    double total = 0.;
    int n = 0;
    for (auto& pr : ctx->events) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, pr.first, pr.second) == hipSuccess) { total += ms; ++n; }
        ctx->event_pool.push_back(pr.first);
        ctx->event_pool.push_back(pr.second);
    }
    ctx->events.clear();
    if (nlaunches) *nlaunches = n;
    return n ? total / n : 0.;
}

int mrgingham_amd_sync(mrgingham_amd_ctx* ctx) {
    if (!ctx) return MRGINGHAM_AMD_ERR_ARG;
    MRG_HIP_CHECK(hipSetDevice(ctx->device));
    for (int i = 0; i < ctx->nstreams; ++i) MRG_HIP_CHECK(hipStreamSynchronize(ctx->streams[i]));
    MRG_HIP_CHECK(hipGetLastError());
    if (ctx->async_error) { ctx->async_error = false; return MRGINGHAM_AMD_ERR_DEVICE; }
    const int nact = ctx->active_nframes;
    ctx->active_nframes = 0;
    if (nact > 0 && ctx->status.p) {
        ctx->host_status.resize(nact);
        MRG_HIP_CHECK(hipMemcpy(ctx->host_status.data(), ctx->status.p, sizeof(int32_t) * nact,
                                hipMemcpyDeviceToHost));
        for (int f = 0; f < nact; ++f)
            if (ctx->host_status[f]) {
                MRG_HIP_CHECK(hipMemset(ctx->status.p, 0, sizeof(int32_t) * nact));
                return fail(ctx, MRGINGHAM_AMD_ERR_CAPACITY,
                            "frame %d: component tables overflowed (status %d); lower \"hot_capacity_shift\" "
                            "(now %d) with mrgingham_amd_set_option and re-run",
                            f, ctx->host_status[f], ctx->cap_shift);
            }
    }
    return MRGINGHAM_AMD_OK;
}

int mrgingham_amd_chess_response_batch(mrgingham_amd_ctx* ctx, const mrgingham_amd_frames* fr, int level,
                                       int clamp, int16_t* d_response, void* stream) {
    int rc = validate_frames(ctx, fr);
    if (rc) return rc;
    int w, h;
    if (level_dims(fr->width, fr->height, level, &w, &h) || !d_response)
        return fail(ctx, MRGINGHAM_AMD_ERR_ARG, "bad level %d or NULL response", level);
    if (fr->nframes == 0) return 0;
    MRG_HIP_CHECK(hipSetDevice(ctx->device));
    if (level > 0) {
        int w1, h1;
        level_dims(fr->width, fr->height, 1, &w1, &h1);
        if ((rc = ensure(ctx, ctx->level_img, (size_t)fr->nframes * w1 * h1 + 16))) return rc;
    }
    hipStream_t s = (hipStream_t)stream;  // used as given: NULL is HIP's default stream
    queue_level_response(ctx, fr, level, 0, fr->nframes, clamp != 0, false, d_response, s, level == 0);
    MRG_HIP_CHECK(hipGetLastError());
    return 0;
}

int mrgingham_amd_decimate_batch(mrgingham_amd_ctx* ctx, const mrgingham_amd_frames* fr, int level, uint8_t* d_out,
                                 void* stream) {
    int rc = validate_frames(ctx, fr);
    if (rc) return rc;
    int w, h;
    if (level_dims(fr->width, fr->height, level, &w, &h) || !d_out)
        return fail(ctx, MRGINGHAM_AMD_ERR_ARG, "bad level %d or NULL output", level);
    if (fr->nframes == 0) return 0;
    MRG_HIP_CHECK(hipSetDevice(ctx->device));
    hipStream_t s = (hipStream_t)stream;  // used as given: NULL is HIP's default stream
    if (level == 0) {
        for (int f = 0; f < fr->nframes; ++f)
            MRG_HIP_CHECK(hipMemcpy2DAsync(d_out + (size_t)f * w * h, w, fr->frames + (size_t)f * fr->frame_pitch,
                                           fr->stride, w, h, hipMemcpyDeviceToDevice, s));
        return 0;
    }
    FrameBatch fb{fr->frames, fr->frame_pitch, fr->width, fr->height, fr->stride};
    launch_decimate(fb, level, d_out, (long long)w * h, w, h, 0, fr->nframes, s);
    MRG_HIP_CHECK(hipGetLastError());
    return 0;
}

int mrgingham_amd_box_blur_batch(mrgingham_amd_ctx* ctx, const mrgingham_amd_frames* fr, int radius, uint8_t* d_out,
                                 void* stream) {
    int rc = validate_frames(ctx, fr);
    if (rc) return rc;
    if (radius < 0 || radius > 64 || !d_out) return fail(ctx, MRGINGHAM_AMD_ERR_ARG, "bad blur radius or output");
    if (fr->nframes == 0) return 0;
    MRG_HIP_CHECK(hipSetDevice(ctx->device));
    hipStream_t s = (hipStream_t)stream;  // used as given: NULL is HIP's default stream
    FrameBatch fb{fr->frames, fr->frame_pitch, fr->width, fr->height, fr->stride};
    launch_box_blur(fb, radius, d_out, 0, fr->nframes, s);
    MRG_HIP_CHECK(hipGetLastError());
    return 0;
}

int mrgingham_amd_detect_batch(mrgingham_amd_ctx* ctx, const mrgingham_amd_frames* fr, int level, int32_t* d_xy,
                               int capacity_per_frame, int32_t* d_counts) {
    int rc = validate_frames(ctx, fr);
    if (rc) return rc;
    int w, h;
    if (level_dims(fr->width, fr->height, level, &w, &h))
        return fail(ctx, MRGINGHAM_AMD_ERR_ARG, "Got an unreasonable image_pyramid_level = %d", level);
    if (!d_xy || !d_counts || capacity_per_frame < 0) return fail(ctx, MRGINGHAM_AMD_ERR_ARG, "NULL outputs");
    if (fr->nframes == 0) return 0;
    MRG_HIP_CHECK(hipSetDevice(ctx->device));
    if ((rc = ensure_scratch(ctx, fr->nframes, fr->width, fr->height, 0))) return rc;
    if (fr->nframes > ctx->active_nframes) ctx->active_nframes = fr->nframes;
    const CompTables t = tables_of(ctx);
    const DetectOut out{d_xy, capacity_per_frame, d_counts};
    for (int c = 0; c < ctx->nstreams; ++c) {
        int f0, n;
        chunk_of(ctx, fr->nframes, c, &f0, &n);
        if (n <= 0) continue;
        hipStream_t s = ctx->streams[c];
        const LevelBatch lb = queue_level_response(ctx, fr, level, f0, n, true, true, nullptr, s, level == 0);
        launch_cc_detect(lb, t, level, out, f0, n, s);
    }
    MRG_HIP_CHECK(hipGetLastError());
    return 0;
}

int mrgingham_amd_refine_batch(mrgingham_amd_ctx* ctx, const mrgingham_amd_frames* fr, int level, double* d_points,
                               signed char* d_levels, const int32_t* d_npoints, int points_pitch,
                               int32_t* d_nrefined) {
    int rc = validate_frames(ctx, fr);
    if (rc) return rc;
    int w, h;
    if (level_dims(fr->width, fr->height, level, &w, &h))
        return fail(ctx, MRGINGHAM_AMD_ERR_ARG, "Got an unreasonable image_pyramid_level = %d", level);
    if (!d_points || !d_levels || !d_npoints || points_pitch <= 0)
        return fail(ctx, MRGINGHAM_AMD_ERR_ARG, "NULL point buffers");
    if (fr->nframes == 0) return 0;
    MRG_HIP_CHECK(hipSetDevice(ctx->device));
    if ((rc = ensure_scratch(ctx, fr->nframes, fr->width, fr->height, points_pitch))) return rc;
    if (fr->nframes > ctx->active_nframes) ctx->active_nframes = fr->nframes;
    const CompTables t = tables_of(ctx);
    RefineIO io{d_points, d_levels, d_npoints, points_pitch, d_nrefined, (int32_t*)ctx->leader.p,
                (int32_t*)ctx->need.p, (int32_t*)ctx->nseeds.p, (uint32_t*)ctx->seeds.p};
    for (int c = 0; c < ctx->nstreams; ++c) {
        int f0, n;
        chunk_of(ctx, fr->nframes, c, &f0, &n);
        if (n <= 0) continue;
        hipStream_t s = ctx->streams[c];
        const LevelBatch lb = queue_level_response(ctx, fr, level, f0, n, true, true, nullptr, s, level == 0);
        launch_cc_refine(lb, t, level, io, f0, n, s);
    }
    MRG_HIP_CHECK(hipGetLastError());
    return 0;
}

int mrgingham_amd_chain_batch(mrgingham_amd_ctx* ctx, const mrgingham_amd_frames* fr, int start_level,
                              double* d_points, signed char* d_levels, int32_t* d_npoints, int points_pitch) {
    int rc = validate_frames(ctx, fr);
    if (rc) return rc;
    int w, h;
    if (level_dims(fr->width, fr->height, start_level, &w, &h))
        return fail(ctx, MRGINGHAM_AMD_ERR_ARG, "Got an unreasonable image_pyramid_level = %d", start_level);
    if (!d_points || !d_levels || !d_npoints || points_pitch <= 0)
        return fail(ctx, MRGINGHAM_AMD_ERR_ARG, "NULL point buffers");
    if (fr->nframes == 0) return 0;
    MRG_HIP_CHECK(hipSetDevice(ctx->device));
    if ((rc = ensure_scratch(ctx, fr->nframes, fr->width, fr->height, points_pitch))) return rc;
    if (fr->nframes > ctx->active_nframes) ctx->active_nframes = fr->nframes;
    const CompTables t = tables_of(ctx);
    const DetectOut out{(int32_t*)ctx->cand_xy.p, points_pitch, (int32_t*)ctx->cand_counts.p};
    RefineIO io{d_points, d_levels, d_npoints, points_pitch, nullptr, (int32_t*)ctx->leader.p,
                (int32_t*)ctx->need.p, (int32_t*)ctx->nseeds.p, (uint32_t*)ctx->seeds.p};
    for (int c = 0; c < ctx->nstreams; ++c) {
        int f0, n;
        chunk_of(ctx, fr->nframes, c, &f0, &n);
        if (n <= 0) continue;
        hipStream_t s = ctx->streams[c];
        LevelBatch lb = queue_level_response(ctx, fr, start_level, f0, n, true, true, nullptr, s, start_level == 0);
        launch_cc_detect(lb, t, start_level, out, f0, n, s);                       // mrgingham.cc:50
        launch_points_from_candidates(out.xy, out.capacity, out.counts, d_points,  // find_grid.cc:353-354
                                      d_levels, d_npoints, points_pitch, start_level, f0, n, s);
        for (int L = start_level - 1; L >= 0; --L) {                               // mrgingham.cc:87-99
            lb = queue_level_response(ctx, fr, L, f0, n, true, true, nullptr, s, L == 0);
            launch_cc_refine(lb, t, L, io, f0, n, s);
        }
    }
    MRG_HIP_CHECK(hipGetLastError());
    return 0;
}

/* ------------------------------------------------------------------------ */
/* Reference symbols: host buffers in, host results out                      */
/* ------------------------------------------------------------------------ */

static mrgingham_amd_ctx* thread_ctx() {
    // One context per calling thread: the reference is called from N worker
    // pthreads at once (mrgingham-from-image.cc:374-379).
    struct Holder {
        mrgingham_amd_ctx* ctx = nullptr;
        ~Holder() { /* leaked on purpose: HIP may already be torn down at thread exit */ }
    };
    static thread_local Holder h;
    if (!h.ctx) {
        const char* d = getenv("MRGINGHAM_AMD_DEVICE");
        h.ctx = mrgingham_amd_create(d ? atoi(d) : 0);
        if (h.ctx) mrgingham_amd_set_option(h.ctx, "streams", 1);
    }
    return h.ctx;
}

// Upload one host frame as a dense device image; fills `fr`.
static int upload_frame(mrgingham_amd_ctx* ctx, const void* host, int rows, int cols, int stride,
                        mrgingham_amd_frames* fr) {
    int rc;
    if ((rc = ensure(ctx, ctx->io_frame, (size_t)rows * cols + 64))) return rc;
    // stream-ordered on streams[0]: the kernels that read it are queued on the same stream
    if (rows > 0 && cols > 0)
        MRG_HIP_CHECK(hipMemcpy2DAsync(ctx->io_frame.p, cols, host, stride, cols, rows, hipMemcpyHostToDevice,
                                       ctx->streams[0]));
    fr->frames = (const uint8_t*)ctx->io_frame.p;
    fr->frame_pitch = (int64_t)rows * cols;
    fr->nframes = 1;
    fr->width = cols;
    fr->height = rows;
    fr->stride = cols;
    return 0;
}

void mrgingham_ChESS_response_5(int16_t* response, const uint8_t* image, int w, int h, int stride) {
    if (w < 15 || h < 15) return;  // no interior: the reference's loops do not execute (ChESS.c:62-63)
    mrgingham_amd_ctx* ctx = thread_ctx();
    if (!ctx || !response || !image) {
        fprintf(stderr, "mrgingham_amd: mrgingham_ChESS_response_5: no device context; response not written\n");
        return;
    }
    hipSetDevice(ctx->device);
    mrgingham_amd_frames fr;
    if (upload_frame(ctx, image, h, w, stride, &fr)) return;
    if (ensure(ctx, ctx->io_out, (size_t)w * h * 2 + 64)) return;
    if (mrgingham_amd_chess_response_batch(ctx, &fr, 0, 0, (int16_t*)ctx->io_out.p, ctx->streams[0])) return;
    // interior only, like the reference: the 7-pixel frame of `response` is not touched
    const size_t off = (size_t)kMargin * w + kMargin;
    hipError_t e = hipMemcpy2DAsync(response + off, (size_t)w * 2, (const int16_t*)ctx->io_out.p + off,
                                    (size_t)w * 2, (size_t)(w - 2 * kMargin) * 2, h - 2 * kMargin,
                                    hipMemcpyDeviceToHost, ctx->streams[0]);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->streams[0]);
    if (e != hipSuccess) fprintf(stderr, "mrgingham_amd: ChESS response failed: %s\n", hipGetErrorString(e));
}

// Common checks of apply_image_pyramid_scaling (find_chessboard_corners.cc:433-473).
static bool check_level_and_layout(const char* fn, int Nrows, int Ncols, int stride, int level) {
    if (level < 0 || level > 10) {
        fprintf(stderr, "mrgingham_amd: %s(): Got an unreasonable image_pyramid_level = %d. Sorry.\n", fn, level);
        return false;
    }
    if (level == 0 && stride != Ncols && Nrows != 1) {
        fprintf(stderr, "mrgingham_amd: %s(): I can only handle continuous arrays (stride == width) currently."
                        " Sorry.\n", fn);
        return false;
    }
    return true;
}

bool find_chessboard_corners_from_image_array_C(int Nrows, int Ncols, int stride, char* imagebuffer,
                                                int image_pyramid_level, bool doblobs, bool debug,
                                                bool (*add_points)(int* xy, int N, double scale, void* cookie),
                                                void* cookie) {
    (void)debug;  // the reference's /tmp debug dumps are not produced
    if (doblobs) {
        fprintf(stderr, "mrgingham_amd: the blob detector (find_blobs.cc) is not part of this library\n");
        return false;
    }
    if (Nrows < 0 || Ncols < 0 || stride < Ncols || !imagebuffer || !add_points) return false;
    if (!check_level_and_layout(__func__, Nrows, Ncols, stride, image_pyramid_level)) return false;
    mrgingham_amd_ctx* ctx = thread_ctx();
    if (!ctx) return false;
    hipSetDevice(ctx->device);
    const int saved_shift = ctx->cap_shift;
    std::vector<int32_t> xy;
    int32_t count = 0;
    bool ok = false;
    for (int attempt = 0; attempt < 2; ++attempt) {
        mrgingham_amd_frames fr;
        if (upload_frame(ctx, imagebuffer, Nrows, Ncols, stride, &fr)) break;
        if (ensure_scratch(ctx, 1, Ncols, Nrows, 0)) break;
        const int cap = ctx->s_cand_cap;
        if (ensure(ctx, ctx->io_out, (size_t)cap * 8 + 64)) break;
        if (mrgingham_amd_detect_batch(ctx, &fr, image_pyramid_level, (int32_t*)ctx->io_out.p, cap,
                                       (int32_t*)ctx->cand_counts.p))
            break;
        const int rc = mrgingham_amd_sync(ctx);
        if (rc == MRGINGHAM_AMD_ERR_CAPACITY && attempt == 0) {
            ctx->cap_shift = 0;  // adversarial texture: retry with a table entry for every pixel
            continue;
        }
        if (rc) break;
        if (hipMemcpy(&count, ctx->cand_counts.p, sizeof(count), hipMemcpyDeviceToHost) != hipSuccess) break;
        if (count > 0) {
            xy.resize((size_t)count * 2);
            if (hipMemcpy(xy.data(), ctx->io_out.p, (size_t)count * 8, hipMemcpyDeviceToHost) != hipSuccess) break;
        }
        ok = true;
        break;
    }
    ctx->cap_shift = saved_shift;
    if (!ok || count <= 0) return false;  // bridge.cc:61: nothing found -> false, add_points not called
    return (*add_points)(xy.data(), (int)count, 1. / kGridScale, cookie);  // bridge.cc:66-69
}

int refine_chessboard_corners_from_image_array_C(int Nrows, int Ncols, int stride, char* imagebuffer,
                                                 double* points_xy, signed char* level, int Npoints,
                                                 int image_pyramid_level, bool debug) {
    (void)debug;
    if (Nrows < 0 || Ncols < 0 || stride < Ncols || !imagebuffer || Npoints < 0) return 0;
    if (Npoints > 0 && (!points_xy || !level)) return 0;
    if (!check_level_and_layout(__func__, Nrows, Ncols, stride, image_pyramid_level)) return 0;
    if (Npoints == 0) return 0;
    mrgingham_amd_ctx* ctx = thread_ctx();
    if (!ctx) return 0;
    hipSetDevice(ctx->device);
    const int saved_shift = ctx->cap_shift;
    int32_t nrefined = 0;
    bool ok = false;
    for (int attempt = 0; attempt < 2; ++attempt) {
        mrgingham_amd_frames fr;
        if (upload_frame(ctx, imagebuffer, Nrows, Ncols, stride, &fr)) break;
        if (ensure_scratch(ctx, 1, Ncols, Nrows, Npoints)) break;
        // layout of io_out: points | levels | npoints | nrefined
        const size_t o_lv = (size_t)Npoints * 16, o_np = o_lv + (((size_t)Npoints + 7) & ~(size_t)7), o_nr = o_np + 8;
        if (ensure(ctx, ctx->io_out, o_nr + 8)) break;
        char* base = (char*)ctx->io_out.p;
        const int32_t np = Npoints;
        hipStream_t s0 = ctx->streams[0];
        if (hipMemcpyAsync(base, points_xy, (size_t)Npoints * 16, hipMemcpyHostToDevice, s0) != hipSuccess) break;
        if (hipMemcpyAsync(base + o_lv, level, (size_t)Npoints, hipMemcpyHostToDevice, s0) != hipSuccess) break;
        if (hipMemcpyAsync(base + o_np, &np, 4, hipMemcpyHostToDevice, s0) != hipSuccess) break;
        if (hipStreamSynchronize(s0) != hipSuccess) break;  // `np` lives on this stack frame
        if (mrgingham_amd_refine_batch(ctx, &fr, image_pyramid_level, (double*)base, (signed char*)(base + o_lv),
                                       (const int32_t*)(base + o_np), Npoints, (int32_t*)(base + o_nr)))
            break;
        const int rc = mrgingham_amd_sync(ctx);
        if (rc == MRGINGHAM_AMD_ERR_CAPACITY && attempt == 0) { ctx->cap_shift = 0; continue; }
        if (rc) break;
        if (hipMemcpy(&nrefined, base + o_nr, 4, hipMemcpyDeviceToHost) != hipSuccess) break;
        if (hipMemcpy(points_xy, base, (size_t)Npoints * 16, hipMemcpyDeviceToHost) != hipSuccess) break;
        if (hipMemcpy(level, base + o_lv, (size_t)Npoints, hipMemcpyDeviceToHost) != hipSuccess) break;
        ok = true;
        break;
    }
    ctx->cap_shift = saved_shift;
    return ok && nrefined > 0 ? nrefined : 0;
}

}  // extern "C"
