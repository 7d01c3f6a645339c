// Host side of libmrgingham_amd.so: the context (streams + scratch), the batch
// API and the reference's own C symbols as thin wrappers over it.
// See include/mrgingham_amd.h for the contract of every entry point.
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include <dlfcn.h>

#include "../../include/mrgingham_amd.h"
#include "common.h"
#include "grid.h"
#include "image_io.h"
#include "kernels.h"

namespace mrg {

constexpr int kMaxLevel = 10;  // find_chessboard_corners.cc:433-436

struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
};

// Scratch of one pyramid level: level images, the dense response and the
// component tables.  Levels have their own scratch because the pixel kernels of
// level L-1 run while the component search of level L is still working.
struct LevelScratch {
    int w = 0, h = 0, nframes = 0, cap = 0, cand_cap = 0, sort_cap = 0, pitch = 0, shift = -1;
    long long arena_cap = 0;
    DevBuf img, resp, gidx, hot_xy, parent, comp_cnt, roots, comp_first, comp_box, arena, cand, sortkeys;
};

}  // namespace mrg

// Host worker threads of a context (the grid finder of mrgingham_amd_find_boards_batch): started
// once and parked on a condition variable, because spawning threads per call cost more than the
// grid finder itself.
struct HostPool {
    std::vector<std::thread> threads;
    std::mutex m;
    std::condition_variable cv_start, cv_done;
    std::function<void()> job;
    long generation = 0;
    int wanted = 0, running = 0;
    bool stop = false;

    void loop(int id) {
        long seen = 0;
        for (;;) {
            std::unique_lock<std::mutex> lk(m);
            cv_start.wait(lk, [&] { return stop || (generation != seen && id < wanted); });
            if (stop) return;
            seen = generation;
            lk.unlock();
            job();
            lk.lock();
            if (--running == 0) cv_done.notify_all();
        }
    }
    // starts f on n pool threads (the caller is not one of them) and returns; wait() returns when they are done
    void start(int n, const std::function<void()>& f) {
        if (n <= 0) return;
        {
            std::unique_lock<std::mutex> lk(m);
            while ((int)threads.size() < n) {
                const int id = (int)threads.size();
                threads.emplace_back([this, id] { loop(id); });
            }
            job = f;
            wanted = n;
            running = n;
            ++generation;
        }
        cv_start.notify_all();
    }
    void wait() {
        std::unique_lock<std::mutex> lk(m);
        cv_done.wait(lk, [&] { return running == 0; });
        wanted = 0;
    }
    // runs f on n threads (the caller is one of them) and returns when all of them are done
    void run(int n, const std::function<void()>& f) {
        if (n <= 1) { f(); return; }
        start(n - 1, f);
        f();
        wait();
    }
    ~HostPool() {
        {
            std::unique_lock<std::mutex> lk(m);
            stop = true;
        }
        cv_start.notify_all();
        for (auto& t : threads) t.join();
    }
};

constexpr int kMaxSets = 3;  // scratch sets a context can rotate through (option "scratch_sets": 2 or 3)

struct mrgingham_amd_ctx {
    int device = 0;
    int nsets = 2;
    bool nsets_fixed = false;  // option "scratch_sets" given: no automatic choice
    double max_set_bytes = 0;
    // HIP streams of a context: `pix` runs the pixel kernels (pyramid, ChESS) back to back, each
    // over the whole batch; `ccs[set]` run the latency-bound component kernels (a serial chain
    // detect -> refine -> refine ... per call) underneath them, one stream per scratch set so that
    // the chains of consecutive calls overlap each other as well.  Events order cc(L) after pix(L).
    hipStream_t pix = nullptr;
    hipStream_t ccs[kMaxSets] = {};
    // Device buffers the last call of each set wrote / read (caller-owned outputs and inputs):
    // consecutive calls run on different component streams, so a call that touches a buffer the
    // previous call wrote (or writes one it read) must wait for it explicitly.
    struct Span { const char* p; size_t n; };
    std::vector<Span> last_w[kMaxSets], last_r[kMaxSets];
    hipEvent_t ev_pix[mrg::kMaxLevel + 1] = {};
    // Level scratch exists `nsets` times (2, or 3 with option "scratch_sets"): call N+1 fills set (N+1) % nsets on
    // the pixel stream while the component streams still work through the calls before it in the other sets.
    // Two sets keep two component chains in flight, which hides them as long as a chain is shorter than two
    // steps of the pixel kernels; small frames (64 x 640x480: chain 430 us, pixel kernels 62 us) and dense boards
    // want three.
    hipEvent_t ev_cc_done[kMaxSets] = {};
    hipEvent_t ev_ext = nullptr;  // mrgingham_amd_after_stream
    bool cc_pending[kMaxSets] = {};
    int cur = 0;  // scratch set of the call being queued
    std::string err;
    // hot-pixel / component table capacity = level pixels >> shift entries per frame, shift = min(cap_shift,
    // grown_shift[level]).  The default (1/128: 98 304 hot pixels for a 4096x3072 frame, whose bench frames have
    // ~1.3e3 and whose textured ones ~7e4) keeps the tables at 0.35 B per pixel; a frame that needs more is
    // reported (MRGINGHAM_AMD_ERR_CAPACITY at the sync) and the tables of its level GROW to what it asked for, so
    // the same call succeeds when it is made again.
    int cap_shift = 7;
    int grown_shift[mrg::kMaxLevel + 1];
    bool use_v0 = false;  // reference-shaped ChESS kernel instead of the tuned one
    int sparse_subsets = 2;  // option "sparse_subsets": workgroups per frame of the sparse refinement (1 .. 4; 4 measures like 2)
    int chess_variant_hot = 0;  // the levels of a chain (clamp + hot list): 16 = chess_v16_hot_kernel / chess_v16_multi_kernel, 0 = chess_v1
    int pre_fused = 1;  // option "preprocess_fused": CLAHE blend + 3x3 blur in one kernel where the geometry allows (0: always two kernels, the A/B and test hook)
    int chess_seg = 0, chess16_seg = 0;  // options "chess_seg" / "chess16_seg": rows per workgroup of the response kernels, 0 = automatic
    int chess_variant = 0;  // the response without a hot list: 0 = chess_v16_kernel (chess16.hip) where it pays, 1 = chess_v1 always, 16 = chess_v16 wherever it can run
    // levels 3..1 of a chain in one launch (set_option "multi_level_launch"): +1.5 % chain rate, but the
    // component chains then start later and overlap the level-0 launch more (+5 % on that launch): off
    // chain_batch: 0 = one ChESS launch per level; 1 = levels 3..1 in one launch (default: two kernel
    // boundaries fewer per step, 1.129 -> 1.113 ms per 64 frames of 4096x3072); 2 = levels 0..3 in one
    // launch (measured slower: 1.171 ms)
    int multi_level = 1;
    int last_fused = 0, last_merged = 0;  // mrgingham_amd_chain_info
    // option "sparse_refine": chain_batch computes the response of the levels BELOW the start level only in the cells
    // around the points it refines there (chain_batch_sparse)
    int sparse_refine = 1;     // (default: where it pays)
    bool sparse_seen = false;  // a chain has taken the sparse schedule (choose_sets)
    mrg::DevBuf sparse_stat;   // [0]: frames the sparse schedule reported and the library repeated densely (mrgingham_amd_sparse_fallbacks)
    int fuse_pyramid = 1;   // option "fuse_pyramid": chain calls take the level images 1..3 out of the level-0 response kernel
    // component-chain schedule of chain_batch: 0 = every level's component kernels start as soon as
    // that level's response is done; 1 (default) = levels 1 and 0 wait for the level-0 response (they then
    // run underneath the NEXT call's pyramid and small levels instead of underneath this call's level 0:
    // same step time, level-0 launch 668 -> 657 us); 2 = every level waits for the level-0 response
    int cc_schedule = 1;
    int cc_lds = 1;  // component search out of LDS for frames with few hot pixels (option "cc_lds"; bits 1-3: timing ablations)

    mrg::LevelScratch lvs[kMaxSets][mrg::kMaxLevel + 1];
    mrg::DevBuf counters2[kMaxSets];  // per scratch set: hot_cnt words [level][counters_nf], then status words, then path words
    int counters_nf = 0;
    struct PointScratch { mrg::DevBuf leader, need, nseeds, seeds, sroot, cand_xy, cand_counts, cell_list, cell_cnt, flag_list; } pts[kMaxSets];  // per scratch set
    mrg::DevBuf aux_img, io_frame, io_out, io_counts;
    void* io_res_pin = nullptr;  // page-locked: count + first candidates of the single-frame detector
    // page-locked copies of the sets' status words ([level][counters_nf], what mrgingham_amd_sync inspects): they follow every
    // op on its component stream (end_op), so that the sync behind it reads host memory instead of making a blocking copy
    int32_t* status_pin[kMaxSets] = {};
    size_t status_pin_words[kMaxSets] = {};
    bool status_copied[kMaxSets] = {};  // the LAST op on the set left its words in status_pin
    void* io_pin = nullptr;  // page-locked staging of mrgingham_ChESS_response_5's way back
    size_t io_pin_bytes = 0;
    hipEvent_t io_ev[4] = {};
    mrg::DevBuf clk;  // two u64: shader cycles and constant-rate ticks of the probed workgroups (mrgingham_amd_sclk_mhz)
    mrg::DevBuf pre_scratch, pre_tmp, pre_out, pre16_scratch, io_frame16, dbg_img, dbg_resp, blob_scratch, blob_nodes, blob_out;
    mrg::DevBuf fb_xy, fb_cnt, fb_pts, fb_lv, fb_np, fb_frames, fb_frames2;  // find_boards_batch: candidates, counts, boards, levels, point counts
    // find_boards_batch's frame-by-frame retries (full-capacity detect, 1-by-1 refine) run on a single-frame
    // context of THIS context's device, created on first use -- not on the calling thread's default context, which
    // lives on MRGINGHAM_AMD_DEVICE / device 0 and cannot touch another GPU's frames
    mrgingham_amd_ctx* one = nullptr;
    HostPool submit_pool;  // mrgingham_amd_chain_multi: the thread that queues this context's shard
    HostPool pool;  // preprocessing: extrema + tile histograms + LUTs, CLAHE output before the blur
    // mrgingham_amd_find_boards_submit / _collect: one job per scratch set (its level images stay in the set's scratch
    // between the first pass and the refinement)
    struct BoardsJob {
        int state = 0;  // 0 free, 1 first pass queued, 2 host part done (refinement queued, or nothing to refine)
        int ticket = -1, set = 0;
        mrgingham_amd_frames fr{};
        int gridn = 0, level_arg = 0, nthreads = 0, nlev = 0, levs[3] = {0, 0, 0}, cap = 0;
        double* h_boards = nullptr;
        signed char* h_found = nullptr;
        signed char* h_levels = nullptr;  // optional: the refinement level of every corner, [frame][gridn^2]
        bool do_refine = true;
        hipEvent_t ev_a = nullptr, ev_b = nullptr;
        hipEvent_t ev_a0 = nullptr, ev_b0 = nullptr;  // where the two device parts begin (mrgingham_amd_find_boards_stats)
        bool refine_queued = false;
        int top = 0;  // the highest level a board of the job was found at (levels below it are refined)
        // device images of the pinned staging, laid out like it (fb_layout) so that each direction is ONE copy:
        // d_cnt = counts | candidates (first pass), d_pts = boards | levels | point counts (refinement), d_pts0 = boards |
        // levels as they went in (what the dense repeat of a sparse refinement starts from)
        mrg::DevBuf d_cnt, d_pts, d_pts0;
        void* pin = nullptr;  // pinned host staging: counts, candidates | boards, levels, point counts
        size_t pin_bytes = 0;
        // the host part in progress (fb_host_begin .. fb_host_end): candidate lists of frames re-run at full capacity,
        // the frame counter of the grid-finder threads
        std::vector<std::vector<int32_t>> big;
        std::atomic<int> next{0};
        bool grid_running = false;
        int nworkers = 0;
        mrgingham_amd_ctx* owner = nullptr;
    } jobs[kMaxSets];
    // mrgingham_amd_chain_multi: this context's shard of the outputs before it travels to the first context's device
    mrg::DevBuf mg_pts, mg_lv, mg_np;
    hipStream_t mg_stream = nullptr;
    hipEvent_t mg_done = nullptr;
    bool mg_pending = false;
    int next_ticket = 0;
    std::vector<std::pair<int, int>> done_tickets;  // (ticket, status) of jobs completed before they were collected
    int fb_pipeline = 1;  // option "find_boards_pipeline"
    // mrgingham_amd_find_boards_stats: host milliseconds by phase of submit / collect, batches, and what the grid-finder
    // threads did (their thread-local clocks, grid.h, added up under the mutex when a worker leaves)
    double fb_prof[10] = {};
    long fb_prof_n = 0;
    int fb_threads_used = 0;
    double fb_dev_ms[2] = {0, 0};  // device milliseconds: first passes (submit .. candidates on the host), refinements
    std::mutex fb_stat_mu;
    mrg::GridPhaseClock fb_grid{0, 0, 0, 0, 0, 0};
    int pts_nframes = 0, pts_pitch = 0;
    // levels (and frame counts) whose status words must be checked at the next sync
    int pending_frames[kMaxSets][mrg::kMaxLevel + 1] = {};

    // dominant-kernel timing
    bool timing = false;
    bool clk_on = false;  // the engine-clock probe of the level-0 response launches (mrgingham_amd_sclk_mhz)
    std::vector<std::pair<hipEvent_t, hipEvent_t>> events;
    std::vector<hipEvent_t> event_pool;
    std::vector<int32_t> host_status;
    std::vector<char> io_host_block;  // refine_on_device: the points block as it travels
};

namespace mrg {

static inline LevelScratch* cur_levels(mrgingham_amd_ctx* ctx) { return ctx->lvs[ctx->cur]; }

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
    // growth headroom for the small buffers only: the per-level tables of a large batch are gigabytes, and 1/8 on top
    // of them was 1 GiB of the 64-frame bench's scratch
    const size_t want = bytes + (bytes < (64u << 20) ? bytes / 8 : 0) + 256;
    MRG_HIP_CHECK(hipMalloc(&b.p, want));
    b.bytes = want;
    return 0;
}

// Rows between host and device (or device and device): ONE plain copy whenever both sides are dense (every caller's usual case).  The 2-D copy
// of the runtime is a slow path into pageable memory and serialises the threads of a process.
static hipError_t copy_rows_async(void* dst, size_t dpitch, const void* src, size_t spitch, size_t width_bytes, size_t rows,
                                  hipMemcpyKind kind, hipStream_t s) {
    if (dpitch == width_bytes && spitch == width_bytes) return hipMemcpyAsync(dst, src, width_bytes * rows, kind, s);
    return hipMemcpy2DAsync(dst, dpitch, src, spitch, width_bytes, rows, kind, s);
}

static int level_dims(int W, int H, int level, int* w, int* h) {
    if (level < 0 || level > kMaxLevel) return -1;  // find_chessboard_corners.cc:433-441
    auto rnd = [level](int v) {                     // cvRound(v / 2^level): ties to even
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

// Scratch of level `level` for a batch of nframes W x H frames and up to `pitch` points per frame.
static int ensure_level_set(mrgingham_amd_ctx* ctx, int set, int level, int nframes, int W, int H, int pitch) {
    LevelScratch& L = ctx->lvs[set][level];
    int w, h;
    level_dims(W, H, level, &w, &h);
    const int shift = ctx->cap_shift < ctx->grown_shift[level] ? ctx->cap_shift : ctx->grown_shift[level];
    if (nframes <= L.nframes && w == L.w && h == L.h && pitch <= L.pitch && L.shift == shift) return 0;
    if (w == L.w && h == L.h) {
        nframes = nframes > L.nframes ? nframes : L.nframes;
        pitch = pitch > L.pitch ? pitch : L.pitch;
    }
    const long long px = (long long)w * h;
    long long cap = px >> shift;
    if (cap < 4096) cap = 4096;
    if (cap > px) cap = px > 0 ? px : 1;
    if (cap > 0x3fffffff) return fail(ctx, MRGINGHAM_AMD_ERR_ARG, "frame too large");
    // Components that pass the size / peak / margin tests: at most cap / 2 (two pixels each).  Tables of
    // that size are only allocated at shift 0 (the "one entry per pixel" retry of the reference-symbol
    // wrappers); otherwise a fraction, with overflow reported like a hot-list overflow.
    long long cand_cap = shift == 0 ? cap / 2 + 1 : cap / 16 + 1024;
    if (cand_cap < pitch) cand_cap = pitch;
    long long sort_cap = 1;
    while (sort_cap < cand_cap) sort_cap <<= 1;
    // LIFO arena: a super-component of n hot pixels gets 4n + 1 words (every pixel is pushed at most
    // once per neighbour), so 5 * cap bounds a frame.  Same policy as the candidate table.
    const long long arena_cap = (shift == 0 ? 5 * cap : cap + cap / 4) + 16LL * (pitch > 1024 ? pitch : 1024);
    const size_t nf = (size_t)nframes;
    int rc = 0;
    if (level > 0 && (rc = ensure(ctx, L.img, nf * (size_t)px + 16))) return rc;
    if ((rc = ensure(ctx, L.resp, nf * (size_t)px * 2 + 16))) return rc;
    if ((rc = ensure(ctx, L.gidx, nf * (size_t)((w + 7) / 8) * h * 8 + 16))) return rc;
    if (nframes > ctx->counters_nf) {
        MRG_HIP_CHECK(hipDeviceSynchronize());
        const int cnf = nframes + nframes / 8 + 8;
        for (int k = 0; k < kMaxSets; ++k) {
            if ((rc = ensure(ctx, ctx->counters2[k], (size_t)(kMaxLevel + 1) * 3 * cnf * 4))) return rc;
            MRG_HIP_CHECK(hipMemset(ctx->counters2[k].p, 0, ctx->counters2[k].bytes));
            ctx->status_copied[k] = false;  // (the layout of the words changes with counters_nf)
        }
        ctx->counters_nf = cnf;
    }
    if ((rc = ensure(ctx, L.hot_xy, nf * (size_t)cap * 4))) return rc;
    if ((rc = ensure(ctx, L.parent, nf * (size_t)cap * 4))) return rc;
    if ((rc = ensure(ctx, L.comp_cnt, nf * (size_t)cap * 4))) return rc;
    if ((rc = ensure(ctx, L.roots, nf * (size_t)cap * 4))) return rc;
    if ((rc = ensure(ctx, L.comp_first, nf * (size_t)cap * 4))) return rc;
    if ((rc = ensure(ctx, L.comp_box, nf * (size_t)cap * 16))) return rc;
    if ((rc = ensure(ctx, L.arena, nf * (size_t)arena_cap * 4))) return rc;
    if ((rc = ensure(ctx, L.cand, nf * (size_t)cand_cap * sizeof(Cand)))) return rc;
    if ((rc = ensure(ctx, L.sortkeys, nf * (size_t)sort_cap * 8))) return rc;
    L.w = w; L.h = h; L.nframes = nframes; L.pitch = pitch;
    L.cap = (int)cap; L.cand_cap = (int)cand_cap; L.sort_cap = (int)sort_cap; L.arena_cap = arena_cap;
    L.shift = shift;
    return 0;
}

// Every device buffer a context owns (one list for destroy and for mrgingham_amd_scratch_bytes).
static std::vector<DevBuf*> level_set_buffers(mrgingham_amd_ctx* ctx, int set, bool with_points) {
    std::vector<DevBuf*> v;
    for (LevelScratch& L : ctx->lvs[set])
        for (DevBuf* b : {&L.img, &L.resp, &L.gidx, &L.hot_xy, &L.parent, &L.comp_cnt, &L.roots, &L.comp_first, &L.comp_box,
                          &L.arena, &L.cand, &L.sortkeys})
            v.push_back(b);
    if (!with_points) return v;
    auto& ps = ctx->pts[set];
    for (DevBuf* b : {&ps.leader, &ps.need, &ps.nseeds, &ps.seeds, &ps.sroot, &ps.cand_xy, &ps.cand_counts, &ps.cell_list, &ps.cell_cnt, &ps.flag_list})
        v.push_back(b);
    return v;
}
static std::vector<DevBuf*> all_buffers(mrgingham_amd_ctx* ctx) {
    std::vector<DevBuf*> v;
    for (int set = 0; set < kMaxSets; ++set) {
        for (DevBuf* b : level_set_buffers(ctx, set, true)) v.push_back(b);
        v.push_back(&ctx->counters2[set]);
    }
    v.push_back(&ctx->sparse_stat);
    for (DevBuf* b : {&ctx->mg_pts, &ctx->mg_lv, &ctx->mg_np}) v.push_back(b);
    for (auto& j : ctx->jobs)
        for (DevBuf* b : {&j.d_cnt, &j.d_pts, &j.d_pts0}) v.push_back(b);
    for (DevBuf* b : {&ctx->io_counts, &ctx->aux_img, &ctx->io_frame, &ctx->io_out, &ctx->pre_scratch, &ctx->pre_tmp,
                      &ctx->pre_out, &ctx->pre16_scratch, &ctx->io_frame16, &ctx->dbg_img, &ctx->dbg_resp, &ctx->blob_scratch, &ctx->blob_nodes, &ctx->blob_out,
                      &ctx->fb_xy, &ctx->fb_cnt, &ctx->fb_pts, &ctx->fb_lv, &ctx->fb_np, &ctx->fb_frames, &ctx->fb_frames2})
        v.push_back(b);
    return v;
}

// How many scratch sets (= calls whose component searches may be in flight) this batch shape gets, unless the
// option "scratch_sets" fixed it: three while three sets of the whole chain's scratch stay below 8 GB (small
// frames, whose search chain is much longer than their pixel kernels: 64 x 640x480 chains 280 k -> 399 k
// frames/s), two otherwise (64 x 4096x3072: no gain from a third, 9.9 GiB each).  A change of the rotation waits
// for everything in flight first.
static int choose_sets(mrgingham_amd_ctx* ctx, const mrgingham_amd_frames* fr) {
    if (ctx->nsets_fixed) return 0;
    const double per_set = 5.0 * (double)fr->nframes * fr->width * fr->height;  // bytes; measured 4.9 per frame pixel at the default table size
    if (per_set > ctx->max_set_bytes) ctx->max_set_bytes = per_set;  // the largest batch so far decides (scratch only grows)
    // (a context that has run sparse chains keeps three sets up to 16 GB: their component chain is what limits a step,
    // (pixel kernels + chain) / sets -- 64 x 4096x3072: 0.47 -> 0.39 ms per step for 11 instead of 7.3 GiB.  Sticky, so
    // that a dense repeat of one call does not free and reallocate a set.)
    const int want = 3.0 * ctx->max_set_bytes <= (ctx->sparse_seen ? 16e9 : 8e9) ? 3 : 2;
    if (want == ctx->nsets) return 0;
    const int rc = mrgingham_amd_sync(ctx);
    if (want < ctx->nsets)  // the sets that leave the rotation give their scratch back (a larger batch has arrived)
        for (int set = want; set < ctx->nsets; ++set) {
            for (DevBuf* b : level_set_buffers(ctx, set, false))  // (the small per-point scratch stays: it is sized for all sets)
                if (b->p) { hipFree(b->p); *b = DevBuf(); }
            for (LevelScratch& L : ctx->lvs[set]) { L.nframes = 0; L.w = L.h = 0; L.pitch = 0; }
        }
    ctx->nsets = want;
    ctx->cur = 0;
    return rc;
}

static int ensure_level(mrgingham_amd_ctx* ctx, int level, int nframes, int W, int H, int pitch) {
    int rc = 0;
    for (int set = 0; !rc && set < ctx->nsets; ++set) rc = ensure_level_set(ctx, set, level, nframes, W, H, pitch);
    return rc;
}

constexpr long long kSparsePaysPixels = 96ll << 20;  // option "sparse_refine" 1: calls with at least this many frame pixels
constexpr int kCellsPerPoint = 9;  // sparse refinement: distinct cells the 3 x 3 seeds of one point can mark (2 x 2 each, one pixel apart)
// Per-call point scratch shared by the levels (the component kernels of the levels of one call
// run one after the other on that call's component stream); one copy per scratch set.
static int ensure_points(mrgingham_amd_ctx* ctx, int nframes, int pitch) {
    if (nframes <= ctx->pts_nframes && pitch <= ctx->pts_pitch) return 0;
    nframes = nframes > ctx->pts_nframes ? nframes : ctx->pts_nframes;
    pitch = pitch > ctx->pts_pitch ? pitch : ctx->pts_pitch;
    const size_t np = (size_t)nframes * (size_t)(pitch > 0 ? pitch : 1);
    int rc = 0;
    for (auto& ps : ctx->pts) {
        if ((rc = ensure(ctx, ps.leader, np * 4))) return rc;
        if ((rc = ensure(ctx, ps.need, np * 4))) return rc;
        if ((rc = ensure(ctx, ps.nseeds, np * 4))) return rc;
        if ((rc = ensure(ctx, ps.seeds, np * 9 * 4))) return rc;
        if ((rc = ensure(ctx, ps.sroot, np * 9 * 4))) return rc;
        if ((rc = ensure(ctx, ps.cand_xy, np * 8))) return rc;
        if ((rc = ensure(ctx, ps.cand_counts, (size_t)nframes * 4))) return rc;
        // sparse refinement: at most 4 cells per seed position of a point and 9 of those, of which at most 9 distinct
        if ((rc = ensure(ctx, ps.cell_list, 2 * np * kCellsPerPoint * 4))) return rc;  // two buffers: consecutive levels alternate
        if ((rc = ensure(ctx, ps.cell_cnt, (size_t)nframes * 4 * kCellHdr * (kMaxLevel + 1)))) return rc;  // per level and frame: the list's header
        if ((rc = ensure(ctx, ps.flag_list, ((size_t)nframes + 1) * 4))) return rc;  // the frames a sparse chain reported
    }
    ctx->pts_nframes = nframes;
    ctx->pts_pitch = pitch;
    return 0;
}

static int32_t* hot_cnt_of(mrgingham_amd_ctx* ctx, int level) {
    return (int32_t*)ctx->counters2[ctx->cur].p + (size_t)level * ctx->counters_nf;
}
static int32_t* status_of(mrgingham_amd_ctx* ctx, int level) {
    return (int32_t*)ctx->counters2[ctx->cur].p + (size_t)(kMaxLevel + 1 + level) * ctx->counters_nf;
}

static int32_t* path_of(mrgingham_amd_ctx* ctx, int level) {
    return (int32_t*)ctx->counters2[ctx->cur].p + (size_t)(2 * (kMaxLevel + 1) + level) * ctx->counters_nf;
}

static CompTables tables_of(mrgingham_amd_ctx* ctx, int level) {
    const LevelScratch& L = cur_levels(ctx)[level];
    CompTables t;
    t.cap = L.cap;
    t.hot_cnt = hot_cnt_of(ctx, level);
    t.hot_xy = (uint32_t*)L.hot_xy.p;
    t.parent = (int32_t*)L.parent.p;
    t.comp_cnt = (int32_t*)L.comp_cnt.p;
    t.comp_box = (int4*)L.comp_box.p;
    t.roots = (int32_t*)L.roots.p;
    t.comp_first = (int32_t*)L.comp_first.p;
    t.gidx = (uint2*)L.gidx.p;
    t.gw = (L.w + 7) / 8;
    t.gidx_pitch = (long long)t.gw * L.h;
    t.arena = (uint32_t*)L.arena.p;
    t.arena_cap = L.arena_cap;
    t.cand_cap = L.cand_cap;
    t.cand = (Cand*)L.cand.p;
    t.sortkeys = (unsigned long long*)L.sortkeys.p;
    t.sort_cap = L.sort_cap;
    t.status = status_of(ctx, level);
    t.path = path_of(ctx, level);
    t.lds_path = ctx->cc_lds;
    t.only = nullptr;
    return t;
}

// The reference-symbol wrappers (and the calls that span several devices: chain_multi, sync_multi, stream_wait_multi,
// gather_rccl) work on the calling thread's context, which may live on another device than the one the
// CALLER has current (the k-th thread's context is on device k % devices): they put the caller's device back when they
// return -- a worker thread of a multi-GPU host (PyTorch, ...) keeps the current device it had.
struct CallerDevice {
    int prev = -1;
    CallerDevice() {
        if (hipGetDevice(&prev) != hipSuccess) {
            prev = -1;
            (void)hipGetLastError();
        }
    }
    ~CallerDevice() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

static hipEvent_t timing_event(mrgingham_amd_ctx* ctx) {
    hipEvent_t e;
    if (!ctx->event_pool.empty()) { e = ctx->event_pool.back(); ctx->event_pool.pop_back(); }
    else hipEventCreate(&e);
    return e;
}

static void launch_chess_any(mrgingham_amd_ctx* ctx, const LevelBatch& lb, const CompTables& t, int n, bool clamp,
                             bool hot, hipStream_t s, bool time_it) {
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (time_it && ctx->timing) {
        e0 = timing_event(ctx);
        e1 = timing_event(ctx);
        hipEventRecord(e0, s);
    }
    if (lb.w > 0 && lb.h > 0 && n > 0) {
#ifdef MRG_EXPERIMENT
        if (ctx->use_v0) launch_chess_v0(lb, t, 0, n, clamp, hot, s);
        else
#endif
        if (!hot && ((ctx->chess_variant == 0 && chess16_pays(lb, n)) || (ctx->chess_variant == 16 && chess16_ok(lb)))) launch_chess16(lb, 0, n, clamp, s, ctx->chess16_seg);
#ifdef MRG_EXPERIMENT
        else if (hot && (ctx->chess_variant_hot & 16) && !t.only && chess16_ok(lb)) launch_chess16_hot(lb, t, 0, n, s);
#endif
        else launch_chess(lb, t, 0, n, clamp, hot, s, ctx->chess_seg);
    }
    if (e0) {
        hipEventRecord(e1, s);
        ctx->events.emplace_back(e0, e1);
    }
}

// Every detect / refine / chain call starts here: the pixel stream must not
// overwrite level scratch the component stream of the previous call still reads.
static void begin_op(mrgingham_amd_ctx* ctx, int max_level) {
    (void)max_level;
    ctx->cur = (ctx->cur + 1) % ctx->nsets;  // this set was last used nsets calls ago
    ctx->status_copied[ctx->cur] = false;    // (until this op's end_op has queued its copy)
    if (ctx->cc_pending[ctx->cur]) hipStreamWaitEvent(ctx->pix, ctx->ev_cc_done[ctx->cur], 0);
    // The hot-pixel counters of this set are zero here: they are zeroed at allocation and again by
    // end_op behind the component kernels that consumed them -- on the component stream, off the
    // pixel stream's critical path.  (Status words only ever accumulate; mrgingham_amd_sync reads
    // and clears them.)
}
static hipStream_t cur_cc(mrgingham_amd_ctx* ctx) { return ctx->ccs[ctx->cur]; }
// Registers the caller-owned device buffers this call writes (w) and reads (r) and makes its
// component stream wait for the previous call (which runs on the OTHER component stream) when
// they overlap anything that call wrote or read-then-we-write.  Call after begin_op.
static void order_after_previous(mrgingham_amd_ctx* ctx, std::initializer_list<mrgingham_amd_ctx::Span> w,
                                 std::initializer_list<mrgingham_amd_ctx::Span> r) {
    const int cur = ctx->cur;
    auto overlaps = [](const mrgingham_amd_ctx::Span& a, const mrgingham_amd_ctx::Span& b) {
        return a.p && b.p && a.n && b.n && a.p < b.p + b.n && b.p < a.p + a.n;
    };
    for (int prev = 0; prev < kMaxSets; ++prev) {  // every call that may still be running on another component stream
        if (prev == cur || !ctx->cc_pending[prev]) continue;
        bool dep = false;
        for (const auto& pw : ctx->last_w[prev]) {
            for (const auto& x : w) dep |= overlaps(x, pw);
            for (const auto& x : r) dep |= overlaps(x, pw);
        }
        for (const auto& pr : ctx->last_r[prev])
            for (const auto& x : w) dep |= overlaps(x, pr);
        if (dep) hipStreamWaitEvent(ctx->ccs[cur], ctx->ev_cc_done[prev], 0);
    }
    ctx->last_w[cur].assign(w.begin(), w.end());
    ctx->last_r[cur].assign(r.begin(), r.end());
}
static void end_op(mrgingham_amd_ctx* ctx) {
    hipMemsetAsync(ctx->counters2[ctx->cur].p, 0, (size_t)(kMaxLevel + 1) * ctx->counters_nf * sizeof(int32_t),
                   cur_cc(ctx));
    hipEventRecord(ctx->ev_cc_done[ctx->cur], cur_cc(ctx));
    ctx->cc_pending[ctx->cur] = true;
    // the set's status words, behind everything of this op that can set one (behind the event too: whoever waits for
    // the op does not wait for the copy; mrgingham_amd_sync waits for the stream)
    const int set = ctx->cur;
    const size_t words = (size_t)(kMaxLevel + 1) * (size_t)ctx->counters_nf;
    if (ctx->status_pin_words[set] < words) {
        if (ctx->status_pin[set]) hipHostFree(ctx->status_pin[set]);
        ctx->status_pin[set] = nullptr;
        ctx->status_pin_words[set] = 0;
        void* p = nullptr;
        if (hipHostMalloc(&p, words * sizeof(int32_t), hipHostMallocDefault) == hipSuccess) {
            ctx->status_pin[set] = (int32_t*)p;
            ctx->status_pin_words[set] = words;
        } else {
            (void)hipGetLastError();
        }
    }
    ctx->status_copied[set] = ctx->status_pin[set] != nullptr && words > 0 &&
                              hipMemcpyAsync(ctx->status_pin[set], status_of(ctx, 0), words * sizeof(int32_t), hipMemcpyDeviceToHost,
                                             cur_cc(ctx)) == hipSuccess;
}

// Level images of levels [1, max_level] of the batch into the level scratch, on the pixel stream.
// One level image (level >= 1) of the batch into `out` (dense, frames back to back): levels 1..3 through the
// one-pass pyramid kernel restricted to that level (16 x 8 source blocks per thread, 16-byte loads; the
// per-pixel kernel took 425 us for level 1 of 64 frames of 4096x3072, this one reads the frames at HBM speed),
// which falls back to the per-pixel kernels itself for ragged shapes.
static void launch_one_level_image(const mrgingham_amd_frames* fr, int level, uint8_t* out, int w, int h, hipStream_t s) {
    const FrameBatch fb{fr->frames, fr->frame_pitch, fr->width, fr->height, fr->stride};
    if (level <= 3) {
        PyramidOut po{};
        po.out[level - 1] = out;
        po.w[level - 1] = w;
        po.h[level - 1] = h;
        launch_pyramid(fb, po, level, fr->nframes, s);
    } else {
        launch_decimate(fb, level, out, (long long)w * h, w, h, 0, fr->nframes, s);
    }
}

static PyramidOut pyramid_out_of(mrgingham_amd_ctx* ctx, int max_level) {
    PyramidOut po{};
    const int top = max_level < 3 ? max_level : 3;
    for (int L = 1; L <= top; ++L) {
        po.out[L - 1] = (uint8_t*)cur_levels(ctx)[L].img.p;
        po.w[L - 1] = cur_levels(ctx)[L].w;
        po.h[L - 1] = cur_levels(ctx)[L].h;
    }
    return po;
}
// `levels_1_to_3` false: those come out of the level-0 response kernel (launch_chess_pyramid)
static void queue_level_images(mrgingham_amd_ctx* ctx, const mrgingham_amd_frames* fr, int max_level,
                               bool levels_1_to_3 = true, bool gentle = false) {
    const FrameBatch fb{fr->frames, fr->frame_pitch, fr->width, fr->height, fr->stride};
    const int top = max_level < 3 ? max_level : 3;
    if (top >= 1 && levels_1_to_3) launch_pyramid(fb, pyramid_out_of(ctx, max_level), top, fr->nframes, ctx->pix, gentle);
    for (int L = 4; L <= max_level; ++L)
        launch_decimate(fb, L, (uint8_t*)cur_levels(ctx)[L].img.p, (long long)cur_levels(ctx)[L].w * cur_levels(ctx)[L].h,
                        cur_levels(ctx)[L].w, cur_levels(ctx)[L].h, 0, fr->nframes, ctx->pix);
}

// ChESS response (+ hot list) of one level for the whole batch on the pixel
// stream; records ev_pix[level].  Level images of levels > 0 must already be queued.
static LevelBatch level_batch_of(mrgingham_amd_ctx* ctx, const mrgingham_amd_frames* fr, int level) {
    LevelScratch& L = cur_levels(ctx)[level];
    LevelBatch lb;
    lb.nframes = fr->nframes;
    lb.w = L.w;
    lb.h = L.h;
    if (level == 0) {
        lb.img = fr->frames;
        lb.img_pitch = fr->frame_pitch;
        lb.img_stride = fr->stride;
    } else {
        lb.img = (const uint8_t*)L.img.p;
        lb.img_pitch = (long long)L.w * L.h;
        lb.img_stride = L.w;
    }
    lb.resp = (int16_t*)L.resp.p;
    lb.resp_pitch = (long long)L.w * L.h;
    if (level == 0 && ctx->clk_on) lb.clk = (unsigned long long*)ctx->clk.p;  // (mrgingham_amd_sclk_mhz)
    return lb;
}
static LevelBatch queue_level_chess(mrgingham_amd_ctx* ctx, const mrgingham_amd_frames* fr, int level) {
    const LevelBatch lb = level_batch_of(ctx, fr, level);
    launch_chess_any(ctx, lb, tables_of(ctx, level), fr->nframes, true, true, ctx->pix, level == 0);
    hipEventRecord(ctx->ev_pix[level], ctx->pix);
    if (fr->nframes > ctx->pending_frames[ctx->cur][level]) ctx->pending_frames[ctx->cur][level] = fr->nframes;
    return lb;
}


// SPARSE REFINEMENT of the points in `io` (at pyramid level `top`, level images of all levels in the current set's
// scratch) through levels top-1 .. 0, on the current set's component stream, level by level: list the cells around the
// points (sparse_cells_kernel for the first level, the refinement kernel of the level above for the others) ->
// response + hot masks in those cells (chess_cells_kernel) -> refinement out of LDS on exactly those hot pixels
// (window mode, `marked<BOXED = true>`).  A frame the LDS kernel cannot take (a blob that reaches the edge of its
// cells, > 512 points, > 2048 hot pixels in the cells that no band cut separates) sets kStatusSparse in its status
// words -- at that level and, because nobody lists its cells any more, at every level below -- and is REPEATED DENSELY
// behind the last sparse level, on the device, before the call completes: its points go back to where they started
// (`restore`), the ordinary response kernel computes its levels and the global-memory refinement replays them, on
// the flagged frames alone (see the end of this function), and the flags are cleared.  So the outputs are the dense
// schedule's on every frame, with no host round trip and nothing for the caller to repeat.
// `dense_only`: no sparse pass at all -- the ordinary kernels on every frame, level by level, on the component stream
// (the refinement of find_boards_submit when the sparse schedule is switched off or does not pay).
static int queue_sparse_levels(mrgingham_amd_ctx* ctx, const mrgingham_amd_frames* fr, int top, RefineIO io,
                               const SparseRestore& restore, bool dense_only = false) {
    auto& ps = ctx->pts[ctx->cur];
    const int nf = fr->nframes;
    hipStream_t cc = cur_cc(ctx);
    LevelBatch lbs[kMaxLevel + 1];
    if (dense_only) {
        for (int L = top - 1; L >= 0; --L) {
            lbs[L] = level_batch_of(ctx, fr, L);
            const CompTables t = tables_of(ctx, L);
            launch_chess_any(ctx, lbs[L], t, nf, true, true, cc, false);
            launch_cc_refine(lbs[L], t, L, io, 0, nf, cc);
            if (nf > ctx->pending_frames[ctx->cur][L]) ctx->pending_frames[ctx->cur][L] = nf;
        }
        return 0;
    }
    const int list_pitch = kCellsPerPoint * io.pitch;
    io.subsets = ctx->sparse_subsets;
    uint32_t* const lists[2] = {(uint32_t*)ps.cell_list.p, (uint32_t*)ps.cell_list.p + (size_t)nf * list_pitch};
    io.list_pitch = list_pitch;
    int32_t* cnt = (int32_t*)ps.cell_cnt.p;  // [level][frame][kCellHdr]
    for (int L = top - 1; L >= 0; --L) {
        lbs[L] = level_batch_of(ctx, fr, L);
        CompTables t = tables_of(ctx, L);
        t.lds_path |= kLdsPathSparse;
        io.cell_list = lists[L & 1];
        io.next_list = lists[(L & 1) ^ 1];
        io.cell_cnt = cnt + (size_t)L * nf * kCellHdr;
        // the cells of this level: listed by the refinement kernel of the level above, by a kernel of its own
        // for the first one (its points come out of the detection / from the caller)
        if (L == top - 1)
            launch_sparse_cells(lbs[L], t, L, io, io.cell_list, cnt + (size_t)L * nf * kCellHdr, list_pitch, 0, nf, cc, cnt, nf);
        launch_chess_cells(lbs[L], t, io.cell_list, io.cell_cnt, list_pitch, 0, nf, cc);
        io.next_cnt = nullptr;
        if (L > 0) {
            const LevelScratch& nx = cur_levels(ctx)[L - 1];
            io.next_cnt = cnt + (size_t)(L - 1) * nf * kCellHdr;
            io.next_w = nx.w;
            io.next_h = nx.h;
            io.next_max_items = tables_of(ctx, L - 1).gidx_pitch / 4;
        }
        launch_cc_refine(lbs[L], t, L, io, 0, nf, cc);
        if (nf > ctx->pending_frames[ctx->cur][L]) ctx->pending_frames[ctx->cur][L] = nf;
    }
    // the dense repeat of what was reported (flag = the level-0 status word: a frame given up at any level is given up
    // at every level below it): three small launches -- the flagged frames as a list; their dense responses at every
    // level (one grid, laid out for kOnlySlots frames whatever the batch); per listed frame restore + the refinement of
    // every level + clear.  Every kernel boundary of this chain costs ~10 us whether or not a frame is flagged, and a
    // full-size grid of the response kernel that finds nothing to do still waits for LDS and registers on a chip the
    // pixel stream keeps full: eleven full-size launches were 8 % of a sparse step.
    int32_t* flags = status_of(ctx, 0);
    int32_t* list = (int32_t*)ps.flag_list.p;
    launch_sparse_flag_list(flags, nf, list, cc);
    RefineIO dio = io;
    dio.cell_list = nullptr;
    dio.cell_cnt = nullptr;
    dio.list_pitch = 0;
    dio.next_cnt = nullptr;
    LevelBatch mlb[kRefineLevelsMax];   // largest level first (launch_chess_multi)
    CompTables mt[kRefineLevelsMax], lt[kRefineLevelsMax];
    for (int L = 0; L < top; ++L) {
        lt[L] = tables_of(ctx, L);
        mlb[L] = lbs[L];
        mt[L] = lt[L];
        mt[L].only = list;
    }
    const bool merged = top >= 2 && chess_multi_ok(mlb, top, nf) && launch_chess_multi(mlb, mt, top, nf, cc, ctx->chess_seg);
    if (!merged)
        for (int L = top - 1; L >= 0; --L) launch_chess(lbs[L], mt[L], 0, nf, true, true, cc, ctx->chess_seg);
    launch_cc_refine_flagged_levels(lbs, lt, top, dio, restore, list, flags, ctx->counters_nf, (int32_t*)ctx->sparse_stat.p, cc);
    return 0;
}

}  // namespace mrg

using namespace mrg;

// find_boards_submit / _collect jobs in flight hold scratch sets between their device passes: every other call that
// rotates through the sets or resizes them completes those jobs first (their results stay collectable)
static void fb_drain(mrgingham_amd_ctx* ctx);
static int fb_submit(mrgingham_amd_ctx* ctx, const mrgingham_amd_frames* fr, int gridn, int image_pyramid_level,
                     double* h_boards, signed char* h_found_level, int nthreads, bool do_refine, signed char* h_levels);

extern "C" {

int mrgingham_amd_abi_version(void) { return MRGINGHAM_AMD_ABI_VERSION; }

#ifndef MRG_KERNEL_ID
#define MRG_KERNEL_ID "unknown"
#endif
const char* mrgingham_amd_kernel_id(void) { return MRG_KERNEL_ID; }

int mrgingham_amd_device_count(void) {
    int n = 0;
    return hipGetDeviceCount(&n) == hipSuccess ? n : 0;
}

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
    for (int& g : ctx->grown_shift) g = 31;
#ifdef MRG_EXPERIMENT
    const char* v0 = getenv("MRGINGHAM_AMD_CHESS_V0");
    ctx->use_v0 = v0 && atoi(v0) != 0;
#endif
    int prio_lo = 0, prio_hi = 0;  // the component stream gets the highest dispatch priority
    hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
    // Experiment hooks (tools/interference_ab.py): MRGINGHAM_AMD_CC_CUS = k confines the component
    // streams to k CUs per XCD (CU-mask bit i is XCD i % 8, CU i / 8: probed with
    // tools/ubench/cu_mask.hip); MRGINGHAM_AMD_PIX_COMPLEMENT = 1 keeps the pixel stream off them.
    // (only in -DMRG_EXPERIMENT builds: the shipped library reads no environment variable but MRGINGHAM_AMD_DEVICE)
#ifdef MRG_EXPERIMENT
    const char* ecc = getenv("MRGINGHAM_AMD_CC_CUS");
    const int cc_cus = ecc ? atoi(ecc) : 0;
    const char* epx = getenv("MRGINGHAM_AMD_PIX_COMPLEMENT");
    const bool pix_compl = epx && atoi(epx) != 0;
#else
    const int cc_cus = 0;
    const bool pix_compl = false;
#endif
    bool ok = true;
    if (cc_cus > 0 && cc_cus < 32) {
        uint32_t mask[8] = {}, inv[8];
        for (int b = 0; b < 8 * cc_cus; ++b) mask[b >> 5] |= 1u << (b & 31);
        for (int i = 0; i < 8; ++i) inv[i] = ~mask[i];
        for (int k = 0; ok && k < kMaxSets; ++k) ok = hipExtStreamCreateWithCUMask(&ctx->ccs[k], 8, mask) == hipSuccess;
        ok = ok &&
             (pix_compl ? hipExtStreamCreateWithCUMask(&ctx->pix, 8, inv)
                        : hipStreamCreateWithPriority(&ctx->pix, hipStreamNonBlocking, prio_lo)) == hipSuccess;
    } else {
        ok = hipStreamCreateWithPriority(&ctx->pix, hipStreamNonBlocking, prio_lo) == hipSuccess;
        for (int k = 0; ok && k < kMaxSets; ++k)
            ok = hipStreamCreateWithPriority(&ctx->ccs[k], hipStreamNonBlocking, prio_hi) == hipSuccess;
    }
    for (int k = 0; ok && k < kMaxSets; ++k)
        ok = hipEventCreateWithFlags(&ctx->ev_cc_done[k], hipEventDisableTiming) == hipSuccess;
    for (int i = 0; ok && i <= kMaxLevel; ++i)
        ok = hipEventCreateWithFlags(&ctx->ev_pix[i], hipEventDisableTiming) == hipSuccess;
    if (!ok) {
        fprintf(stderr, "mrgingham_amd: could not create HIP streams/events\n");
        mrgingham_amd_destroy(ctx);
        return nullptr;
    }
    return ctx;
}

int mrgingham_amd_find_boards_stats(mrgingham_amd_ctx* ctx, double* out, int n, int reset) {
    if (!ctx || !out || n < 0) return MRGINGHAM_AMD_ERR_ARG;
    fb_drain(ctx);
    double v[MRGINGHAM_AMD_FB_STATS] = {};
    const double tick = grid_clock_tick_us();  // (may wait 0.2 ms when the process has only just started: not under the lock)
    {
        std::lock_guard<std::mutex> lk(ctx->fb_stat_mu);
        v[0] = (double)ctx->fb_prof_n;
        v[1] = (double)ctx->fb_threads_used;
        for (int i = 0; i < 7; ++i) v[2 + i] = ctx->fb_prof[i];
        v[9] = (double)ctx->fb_grid.calls; v[10] = (double)ctx->fb_grid.found;
        v[11] = ctx->fb_grid.graph_t * tick; v[12] = ctx->fb_grid.adjacency_t * tick; v[13] = ctx->fb_grid.sequences_t * tick;
        v[14] = ctx->fb_grid.cycles_t * tick;
        v[15] = ctx->fb_dev_ms[0]; v[16] = ctx->fb_dev_ms[1];
        if (reset) {
            for (double& x : ctx->fb_prof) x = 0;
            ctx->fb_prof_n = 0;
            ctx->fb_grid = GridPhaseClock{0, 0, 0, 0, 0, 0};
            ctx->fb_dev_ms[0] = ctx->fb_dev_ms[1] = 0;
        }
    }
    for (int i = 0; i < n && i < MRGINGHAM_AMD_FB_STATS; ++i) out[i] = v[i];
    return MRGINGHAM_AMD_FB_STATS;
}

int mrgingham_amd_grid_clock(double* out6, int reset) {
    if (!out6) return MRGINGHAM_AMD_ERR_ARG;
    GridPhaseClock& c = g_grid_clock;
    const double tick = grid_clock_tick_us();
    out6[0] = (double)c.calls; out6[1] = (double)c.found; out6[2] = c.graph_t * tick; out6[3] = c.adjacency_t * tick;
    out6[4] = c.sequences_t * tick; out6[5] = c.cycles_t * tick;
    if (reset) c = GridPhaseClock{0, 0, 0, 0, 0, 0};
    return 0;
}

void mrgingham_amd_destroy(mrgingham_amd_ctx* ctx) {
    if (!ctx) return;
#ifdef MRG_EXPERIMENT
    if (ctx->fb_prof_n > 0 && getenv("MRG_DBG_FB")) {
        static const char* names[10] = {"submit: checks + scratch", "submit: host part begins (big frames, threads start)", "submit: device passes queued",
                                        "host part: grid finder joined", "host part: refinement queued", "collect: wait for the refinement",
                                        "collect: boards copied", "", "", ""};
        for (int i = 0; i < 7; ++i) fprintf(stderr, "  [fb] %-56s %8.3f ms per batch\n", names[i], ctx->fb_prof[i] / ctx->fb_prof_n);
    }
#endif
    for (auto& j : ctx->jobs)  // batches still in flight are abandoned: only their host threads have to be out
        if (j.grid_running) {
            ctx->pool.wait();
            j.grid_running = false;
        }
    if (ctx->one) mrgingham_amd_destroy(ctx->one);
    hipSetDevice(ctx->device);
    hipDeviceSynchronize();
    for (DevBuf* b : all_buffers(ctx))
        if (b->p) hipFree(b->p);
    for (auto& pr : ctx->events) { hipEventDestroy(pr.first); hipEventDestroy(pr.second); }
    for (auto e : ctx->event_pool) hipEventDestroy(e);
    for (hipEvent_t e : ctx->ev_pix)
        if (e) hipEventDestroy(e);
    for (hipEvent_t e : ctx->ev_cc_done)
        if (e) hipEventDestroy(e);
    if (ctx->ev_ext) hipEventDestroy(ctx->ev_ext);
    if (ctx->io_pin) hipHostFree(ctx->io_pin);
    if (ctx->io_res_pin) hipHostFree(ctx->io_res_pin);
    for (int k = 0; k < kMaxSets; ++k)
        if (ctx->status_pin[k]) hipHostFree(ctx->status_pin[k]);
    for (hipEvent_t e : ctx->io_ev)
        if (e) hipEventDestroy(e);
    if (ctx->mg_done) hipEventDestroy(ctx->mg_done);
    if (ctx->mg_stream) hipStreamDestroy(ctx->mg_stream);
    for (auto& j : ctx->jobs) {
        if (j.ev_a) hipEventDestroy(j.ev_a);
        if (j.ev_b) hipEventDestroy(j.ev_b);
        if (j.ev_a0) hipEventDestroy(j.ev_a0);
        if (j.ev_b0) hipEventDestroy(j.ev_b0);
        if (j.pin) hipHostFree(j.pin);
    }
    if (ctx->pix) hipStreamDestroy(ctx->pix);
    for (hipStream_t c : ctx->ccs)
        if (c) hipStreamDestroy(c);
    delete ctx;
}

int mrgingham_amd_debug_paths(mrgingham_amd_ctx* ctx, int level, int nframes, int32_t* h_paths) {
    if (!ctx || !h_paths || level < 0 || level > kMaxLevel || nframes < 0 || nframes > ctx->counters_nf)
        return MRGINGHAM_AMD_ERR_ARG;
    MRG_HIP_CHECK(hipSetDevice(ctx->device));
    MRG_HIP_CHECK(hipDeviceSynchronize());
    MRG_HIP_CHECK(hipMemcpy(h_paths, path_of(ctx, level), sizeof(int32_t) * (size_t)nframes, hipMemcpyDeviceToHost));
    return MRGINGHAM_AMD_OK;
}

int mrgingham_amd_debug_refine_clock(mrgingham_amd_ctx* ctx, long long* h_ticks12) {
    if (!ctx || !h_ticks12) return MRGINGHAM_AMD_ERR_ARG;
    MRG_HIP_CHECK(hipSetDevice(ctx->device));
    MRG_HIP_CHECK(hipDeviceSynchronize());
    const DevBuf& b = ctx->pts[ctx->cur].sroot;
    if (!b.p || b.bytes < 12 * sizeof(long long)) return fail(ctx, MRGINGHAM_AMD_ERR_ARG, "no refinement has run yet");
    MRG_HIP_CHECK(hipMemcpy(h_ticks12, b.p, 12 * sizeof(long long), hipMemcpyDeviceToHost));
    return MRGINGHAM_AMD_OK;
}

int mrgingham_amd_chain_info(const mrgingham_amd_ctx* ctx, int* fused_pyramid, int* merged_levels) {
    if (!ctx) return MRGINGHAM_AMD_ERR_ARG;
    if (fused_pyramid) *fused_pyramid = ctx->last_fused;
    if (merged_levels) *merged_levels = ctx->last_merged;
    return 0;
}

long long mrgingham_amd_scratch_bytes(const mrgingham_amd_ctx* ctx) {
    if (!ctx) return 0;
    long long total = 0;
    for (const DevBuf* b : all_buffers(const_cast<mrgingham_amd_ctx*>(ctx))) total += (long long)b->bytes;
    if (ctx->one) total += mrgingham_amd_scratch_bytes(ctx->one);
    return total;
}

const char* mrgingham_amd_last_error(const mrgingham_amd_ctx* ctx) { return ctx ? ctx->err.c_str() : "no context"; }

int mrgingham_amd_sparse_fallbacks(mrgingham_amd_ctx* ctx) {
    if (!ctx) return MRGINGHAM_AMD_ERR_ARG;
    if (!ctx->sparse_stat.p) return 0;
    MRG_HIP_CHECK(hipSetDevice(ctx->device));
    MRG_HIP_CHECK(hipDeviceSynchronize());
    int32_t n = 0;
    MRG_HIP_CHECK(hipMemcpy(&n, ctx->sparse_stat.p, sizeof(n), hipMemcpyDeviceToHost));
    MRG_HIP_CHECK(hipMemset(ctx->sparse_stat.p, 0, sizeof(n)));
    return n;
}

void mrgingham_amd_set_kernel_timing(mrgingham_amd_ctx* ctx, int enable) {
    if (!ctx) return;
    ctx->timing = (enable & 1) != 0;   // bit 0: hipEvents around the level-0 response launches
    ctx->clk_on = enable != 0;         // any non-zero value: the engine-clock probe (2 = the probe alone, no events)
    if (ctx->clk_on && !ctx->clk.p) {  // the engine-clock probe's two counters (without them the probe stays off)
        const CallerDevice keep;
        hipSetDevice(ctx->device);
        if (ensure(ctx, ctx->clk, 2 * sizeof(unsigned long long)) == 0) hipMemset(ctx->clk.p, 0, 2 * sizeof(unsigned long long));
    }
}

double mrgingham_amd_sclk_mhz(mrgingham_amd_ctx* ctx) {
    if (!ctx || !ctx->clk.p) return 0.;
    const CallerDevice keep;
    hipSetDevice(ctx->device);
    if (hipDeviceSynchronize() != hipSuccess) return 0.;
    unsigned long long c[2] = {0, 0};
    if (hipMemcpy(c, ctx->clk.p, sizeof(c), hipMemcpyDeviceToHost) != hipSuccess) return 0.;
    hipMemset(ctx->clk.p, 0, sizeof(c));
    int khz = 0;  // rate of s_memrealtime
    if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, ctx->device) != hipSuccess || khz <= 0) khz = 100000;
    return c[1] ? (double)c[0] / (double)c[1] * khz * 1e-3 : 0.;
}

/* tunables (not part of the reference surface) */
int mrgingham_amd_set_option(mrgingham_amd_ctx* ctx, const char* name, int value) {
    if (!ctx || !name) return MRGINGHAM_AMD_ERR_ARG;
    fb_drain(ctx);
    if (!strcmp(name, "hot_capacity_shift")) {
        if (value < 0 || value > 10) return MRGINGHAM_AMD_ERR_ARG;
        ctx->cap_shift = value;
        for (int& g : ctx->grown_shift) g = 31;  // an explicit choice starts over
        return 0;
    }
    if (!strcmp(name, "hot_capacity_shift_temporary")) {
        // the retry of a caller (one table entry per pixel for one call, then back): the capacity changes, what the tables
        // have GROWN to is kept -- a context that has met an adversarial frame does not forget it (the C wrappers do the
        // same around their last attempt)
        if (value < 0 || value > 10) return MRGINGHAM_AMD_ERR_ARG;
        ctx->cap_shift = value;
        return 0;
    }
#ifdef MRG_EXPERIMENT
    if (!strcmp(name, "chess_v0")) { ctx->use_v0 = value != 0; return 0; }
#endif
    if (!strcmp(name, "chess_variant")) {
        if (value != 0 && value != 1 && value != 16) return MRGINGHAM_AMD_ERR_ARG;
        ctx->chess_variant = value;
        return 0;
    }
#ifdef MRG_EXPERIMENT
    if (!strcmp(name, "chess16_pair")) { mrg::chess16_pair = value != 0; return 0; }
    if (!strcmp(name, "chess_variant_hot")) {
        if (value & ~48) return MRGINGHAM_AMD_ERR_ARG;  // 16: levels below 0 on chess_v16, 32: level 0 with the level images on chess_v16
        ctx->chess_variant_hot = value;
        return 0;
    }
#endif
    if (!strcmp(name, "sparse_subsets")) {
        if (value < 1 || value > 4) return MRGINGHAM_AMD_ERR_ARG;
        ctx->sparse_subsets = value;
        return 0;
    }
#ifdef MRG_EXPERIMENT
    if (!strcmp(name, "clahe_hist_copies")) { mrg::clahe_hist_copies = value; return 0; }
#endif
    if (!strcmp(name, "preprocess_fused")) { ctx->pre_fused = value != 0; return 0; }
    if (!strcmp(name, "chess16_seg") || !strcmp(name, "chess_seg")) {
        // rows per workgroup of chess_v16_kernel / the chess_v1 kernels of THIS context (0 = automatic): the frame is cut into
        // ceil(height / value) balanced segments (common.h, segment_rows)
        if (value < 0 || value > 65536) return fail(ctx, MRGINGHAM_AMD_ERR_ARG, "%s: 0 (automatic) or a row count", name);
        (name[5] == '1' ? ctx->chess16_seg : ctx->chess_seg) = value;
        return 0;
    }
    if (!strcmp(name, "multi_level_launch")) { ctx->multi_level = value < 0 ? 0 : value > 2 ? 2 : value; return 0; }
#ifdef MRG_EXPERIMENT
    if (!strcmp(name, "cc_schedule")) { ctx->cc_schedule = value; return 0; }
    if (!strcmp(name, "chess_multi_min_blocks")) { mrg::chess_multi_min_blocks = value; return 0; }
    if (!strcmp(name, "chess_stage")) { mrg::chess_stage_override = value; return 0; }
    if (!strcmp(name, "pyramid_lds_pad")) { mrg::pyramid_lds_pad = value; return 0; }
#endif
    if (!strcmp(name, "scratch_sets")) {
        if (value != 0 && (value < 2 || value > kMaxSets))
            return fail(ctx, MRGINGHAM_AMD_ERR_ARG, "scratch_sets must be 0 (automatic), 2 or %d", kMaxSets);
        const int rc = mrgingham_amd_sync(ctx);  // nothing may be in flight when the rotation changes
        ctx->nsets_fixed = value != 0;
        if (value) ctx->nsets = value;
        ctx->cur = 0;
        return rc;
    }
    if (!strcmp(name, "fuse_pyramid")) { ctx->fuse_pyramid = value != 0; return 0; }
    if (!strcmp(name, "find_boards_pipeline")) { ctx->fb_pipeline = value != 0; return 0; }
    if (!strcmp(name, "sparse_refine")) {
        if (value < 0 || value > 2) return MRGINGHAM_AMD_ERR_ARG;
        ctx->sparse_refine = value;
        return 0;
    }
    if (!strcmp(name, "cc_lds")) {
        // 0 / 1 and the test hook 256 (no banding, no windows: every result is still exact); the timing ablations
        // (bits 2, 4, 8, 16, 128) and the phase clock (512) exist in -DMRG_EXPERIMENT builds only
#ifndef MRG_EXPERIMENT
        if (value & ~(1 | 256)) return fail(ctx, MRGINGHAM_AMD_ERR_ARG, "cc_lds: only 0, 1 and 1 | 256 in this build");
#endif
        ctx->cc_lds = value;
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

// The status words of one (scratch set, level) as the host has them (`host`: nact words), inspected; `words` = where they
// live on the device: cleared when any is set.  A table overflow grows the tables of the level to what the fullest frame
// asked for.  *rc keeps the first error.  Nothing of that set may be running at that level.
static int inspect_status(mrgingham_amd_ctx* ctx, int set, int level, const int32_t* host, int nact, int32_t* words, int* rc,
                          bool quiet) {
    // every pending status block is inspected and cleared; only the first error is reported
    bool dirty = false;
    int flags = 0, first = -1;
    long long need = 0;  // hot pixels the fullest frame asked for (status words carry it in units of 64)
    for (int f = 0; f < nact; ++f) {
        const int st = host[f];
        if (!st) continue;
        dirty = true;
        if (first < 0) first = f;
        flags |= st & 0xff;
        const long long n = (long long)((uint32_t)st >> 8) * 64;
        if (n > need) need = n;
    }
    if (!dirty) return 0;
    if ((flags & kStatusSparse) && !(flags & (kStatusHotOverflow | kStatusCandOverflow))) {
        // cannot happen: the dense repeat behind every sparse refinement clears the flag (queue_sparse_levels)
        if (*rc == MRGINGHAM_AMD_OK)
            *rc = fail(ctx, MRGINGHAM_AMD_ERR_DEVICE, "internal: frame %d, level %d left a sparse-refinement flag behind", first, level);
    } else {
        // grow the tables of this level to what was asked for (+25 %); candidate / LIFO overflow: four times
        const LevelScratch& LS = ctx->lvs[set][level];
        const long long px = (long long)LS.w * LS.h;
        int sh = LS.shift;
        if (flags & kStatusHotOverflow)
            while (sh > 0 && (px >> sh) < need + need / 4) --sh;
        if (flags & kStatusCandOverflow) sh = sh >= 2 ? (sh - 2 < LS.shift - 2 ? sh - 2 : LS.shift - 2) : 0;
        if (sh < 0) sh = 0;
        if (sh < ctx->grown_shift[level]) ctx->grown_shift[level] = sh;
        if (quiet) *rc = MRGINGHAM_AMD_ERR_CAPACITY;  // (the caller has dealt with the frames; the tables grow for the next batch)
        else if (*rc == MRGINGHAM_AMD_OK)
            *rc = fail(ctx, MRGINGHAM_AMD_ERR_CAPACITY,
                       "frame %d, level %d: component tables overflowed (status %d, %lld hot pixels asked for); "
                       "the tables of this level grow from 1/%d to 1/%d of its pixels: make the call again",
                       first, level, flags, need, 1 << LS.shift, 1 << sh);
    }
    MRG_HIP_CHECK(hipMemset(words, 0, sizeof(int32_t) * nact));
    return 0;
}

static int32_t* status_words(mrgingham_amd_ctx* ctx, int set, int level) {
    const int saved = ctx->cur;
    ctx->cur = set;
    int32_t* const words = status_of(ctx, level);
    ctx->cur = saved;
    return words;
}

// one (scratch set, level): its words copied, inspected, cleared
static int harvest_status(mrgingham_amd_ctx* ctx, int set, int level, int* rc, bool quiet = false) {
    const int nact = ctx->pending_frames[set][level];
    ctx->pending_frames[set][level] = 0;
    if (nact <= 0 || !ctx->counters2[set].p) return 0;
    int32_t* const words = status_words(ctx, set, level);
    ctx->host_status.resize(nact);
    MRG_HIP_CHECK(hipMemcpy(ctx->host_status.data(), words, sizeof(int32_t) * nact, hipMemcpyDeviceToHost));
    return inspect_status(ctx, set, level, ctx->host_status.data(), nact, words, rc, quiet);
}

// every level of a scratch set: the levels' status words are one block of the set's counter buffer, so ONE copy brings all
// of them (a blocking 256-byte copy costs 15-30 us: a chain call's sync made four of them, a sync behind pipelined chain
// calls up to eight)
static int harvest_set(mrgingham_amd_ctx* ctx, int set, int* rc) {
    int lo = -1, hi = -1;
    for (int level = 0; level <= kMaxLevel; ++level)
        if (ctx->pending_frames[set][level] > 0) {
            if (lo < 0) lo = level;
            hi = level;
        }
    if (lo < 0 || !ctx->counters2[set].p) {
        for (int level = 0; level <= kMaxLevel; ++level) ctx->pending_frames[set][level] = 0;
        return 0;
    }
    const size_t cnf = (size_t)ctx->counters_nf, nwords = (size_t)(hi - lo + 1) * cnf;
    const int32_t* host;
    const bool pinned = ctx->status_copied[set] && ctx->status_pin[set] && ctx->status_pin_words[set] >= (size_t)(kMaxLevel + 1) * cnf;
    if (pinned) {  // (the set's stream has been waited for: the copy end_op queued has landed)
        host = ctx->status_pin[set] + (size_t)lo * cnf;
    } else {
        ctx->host_status.resize(nwords);
        MRG_HIP_CHECK(hipMemcpy(ctx->host_status.data(), status_words(ctx, set, lo), sizeof(int32_t) * nwords, hipMemcpyDeviceToHost));
        host = ctx->host_status.data();
    }
    for (int level = lo; level <= hi; ++level) {
        const int nact = ctx->pending_frames[set][level];
        ctx->pending_frames[set][level] = 0;
        if (nact <= 0) continue;
        const int32_t* hw = host + (size_t)(level - lo) * cnf;
        bool any = false;
        for (int f = 0; f < nact && !any; ++f) any = hw[f] != 0;
        const int r = inspect_status(ctx, set, level, hw, nact, status_words(ctx, set, level), rc, false);
        if (r) return r;
        if (any && pinned) memset(ctx->status_pin[set] + (size_t)level * cnf, 0, (size_t)nact * sizeof(int32_t));  // (cleared on the device: here too)
    }
    return 0;
}

int mrgingham_amd_sync(mrgingham_amd_ctx* ctx) {
    if (!ctx) return MRGINGHAM_AMD_ERR_ARG;
    MRG_HIP_CHECK(hipSetDevice(ctx->device));
    MRG_HIP_CHECK(hipStreamSynchronize(ctx->pix));
    for (int set = 0; set < kMaxSets; ++set) {
        MRG_HIP_CHECK(hipStreamSynchronize(ctx->ccs[set]));
        ctx->cc_pending[set] = false;
    }
    MRG_HIP_CHECK(hipGetLastError());
    int rc = MRGINGHAM_AMD_OK;
    for (int set = 0; set < kMaxSets; ++set) {
        const int r = harvest_set(ctx, set, &rc);
        if (r) return r;
    }
    return rc;
}

int mrgingham_amd_stream_wait(mrgingham_amd_ctx* ctx, void* stream) {
    if (!ctx) return MRGINGHAM_AMD_ERR_ARG;
    MRG_HIP_CHECK(hipSetDevice(ctx->device));
    for (int set = 0; set < kMaxSets; ++set)  // consecutive calls finish on different component streams
        if (ctx->cc_pending[set]) MRG_HIP_CHECK(hipStreamWaitEvent((hipStream_t)stream, ctx->ev_cc_done[set], 0));
    return MRGINGHAM_AMD_OK;
}

int mrgingham_amd_after_stream(mrgingham_amd_ctx* ctx, void* stream) {
    if (!ctx) return MRGINGHAM_AMD_ERR_ARG;
    MRG_HIP_CHECK(hipSetDevice(ctx->device));
    if (!ctx->ev_ext) MRG_HIP_CHECK(hipEventCreateWithFlags(&ctx->ev_ext, hipEventDisableTiming));
    MRG_HIP_CHECK(hipEventRecord(ctx->ev_ext, (hipStream_t)stream));
    // every kernel of a call is ordered behind the pixel stream's work of that call
    MRG_HIP_CHECK(hipStreamWaitEvent(ctx->pix, ctx->ev_ext, 0));
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
    hipStream_t s = (hipStream_t)stream;  // used as given: NULL is HIP's default stream
    LevelBatch lb;
    lb.nframes = fr->nframes;
    lb.w = w;
    lb.h = h;
    if (level == 0) {
        lb.img = fr->frames;
        lb.img_pitch = fr->frame_pitch;
        lb.img_stride = fr->stride;
    } else {
        if ((rc = ensure(ctx, ctx->aux_img, (size_t)fr->nframes * w * h + 16))) return rc;
        launch_one_level_image(fr, level, (uint8_t*)ctx->aux_img.p, w, h, s);
        lb.img = (const uint8_t*)ctx->aux_img.p;
        lb.img_pitch = (long long)w * h;
        lb.img_stride = w;
    }
    lb.resp = d_response;
    lb.resp_pitch = (long long)w * h;
    if (level == 0 && ctx->clk_on) lb.clk = (unsigned long long*)ctx->clk.p;
    launch_chess_any(ctx, lb, CompTables{}, fr->nframes, clamp != 0, false, s, level == 0);
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
            MRG_HIP_CHECK(copy_rows_async(d_out + (size_t)f * w * h, w, fr->frames + (size_t)f * fr->frame_pitch,
                                          fr->stride, w, h, hipMemcpyDeviceToDevice, s));
        return 0;
    }
    launch_one_level_image(fr, level, d_out, w, h, s);
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

int mrgingham_amd_preprocess_batch(mrgingham_amd_ctx* ctx, const mrgingham_amd_frames* fr, int do_clahe,
                                   int blur_radius, uint8_t* d_out, void* stream) {
    int rc = validate_frames(ctx, fr);
    if (rc) return rc;
    if (blur_radius < 0 || blur_radius > 64 || !d_out)
        return fail(ctx, MRGINGHAM_AMD_ERR_ARG, "bad blur radius or output");
    if (fr->nframes == 0) return 0;
    MRG_HIP_CHECK(hipSetDevice(ctx->device));
    hipStream_t s = (hipStream_t)stream;  // used as given: NULL is HIP's default stream
    FrameBatch fb{fr->frames, fr->frame_pitch, fr->width, fr->height, fr->stride};
    if (do_clahe) {
        if (fr->width < 8 || fr->height < 8)
            return fail(ctx, MRGINGHAM_AMD_ERR_ARG, "CLAHE needs a frame of at least 8x8 pixels");
        const size_t frame_bytes = (size_t)fr->width * fr->height;
        if ((rc = ensure(ctx, ctx->pre_scratch, clahe_scratch_bytes(fr->nframes)))) return rc;
        // clip limit 8, default 8x8 tiles: mrgingham-from-image.cc:41-45
        if (blur_radius == 1 && ctx->pre_fused) {
            // the tool's default chain: blend + 3x3 blur in one pass over the frame where the geometry allows it
            uint8_t* tmp = nullptr;
            if (!clahe_blur3_fused(fb, d_out)) {
                if ((rc = ensure(ctx, ctx->pre_tmp, frame_bytes * fr->nframes))) return rc;
                tmp = (uint8_t*)ctx->pre_tmp.p;
            }
            launch_clahe(fb, fr->nframes, 8.0, true, d_out, ctx->pre_scratch.p, s, true, tmp, ctx->clk_on ? (unsigned long long*)ctx->clk.p : nullptr);
        } else {
            uint8_t* clahe_out = d_out;
            if (blur_radius > 0) {
                if ((rc = ensure(ctx, ctx->pre_tmp, frame_bytes * fr->nframes))) return rc;
                clahe_out = (uint8_t*)ctx->pre_tmp.p;
            }
            launch_clahe(fb, fr->nframes, 8.0, true, clahe_out, ctx->pre_scratch.p, s);
            if (blur_radius > 0) {
                const FrameBatch tb{clahe_out, (long long)frame_bytes, fr->width, fr->height, fr->width};
                launch_box_blur(tb, blur_radius, d_out, 0, fr->nframes, s);
            }
        }
    } else {
        launch_box_blur(fb, blur_radius, d_out, 0, fr->nframes, s);  // radius 0 = dense copy
    }
    MRG_HIP_CHECK(hipGetLastError());
    return 0;
}

int mrgingham_amd_detect_batch(mrgingham_amd_ctx* ctx, const mrgingham_amd_frames* fr, int level, int32_t* d_xy,
                               int capacity_per_frame, int32_t* d_counts) {
    int rc = validate_frames(ctx, fr);
    if (rc) return rc;
    fb_drain(ctx);
    int w, h;
    if (level_dims(fr->width, fr->height, level, &w, &h))
        return fail(ctx, MRGINGHAM_AMD_ERR_ARG, "Got an unreasonable image_pyramid_level = %d", level);
    if (fr->nframes == 0) return 0;
    if (!d_xy || !d_counts || capacity_per_frame < 0) return fail(ctx, MRGINGHAM_AMD_ERR_ARG, "NULL outputs");
    MRG_HIP_CHECK(hipSetDevice(ctx->device));
    if ((rc = choose_sets(ctx, fr))) return rc;
    if ((rc = ensure_level(ctx, level, fr->nframes, fr->width, fr->height, 0))) return rc;
    begin_op(ctx, level);
    if (level > 0) {
        launch_one_level_image(fr, level, (uint8_t*)cur_levels(ctx)[level].img.p, w, h, ctx->pix);
    }
    const LevelBatch lb = queue_level_chess(ctx, fr, level);
    order_after_previous(ctx, {{(const char*)d_xy, (size_t)fr->nframes * capacity_per_frame * 8},
                               {(const char*)d_counts, (size_t)fr->nframes * 4}}, {});
    MRG_HIP_CHECK(hipStreamWaitEvent(cur_cc(ctx), ctx->ev_pix[level], 0));
    launch_cc_detect(lb, tables_of(ctx, level), level, DetectOut{d_xy, capacity_per_frame, d_counts}, 0,
                     fr->nframes, cur_cc(ctx));
    end_op(ctx);
    MRG_HIP_CHECK(hipGetLastError());
    return 0;
}

int mrgingham_amd_refine_batch(mrgingham_amd_ctx* ctx, const mrgingham_amd_frames* fr, int level, double* d_points,
                               signed char* d_levels, const int32_t* d_npoints, int points_pitch,
                               int32_t* d_nrefined) {
    int rc = validate_frames(ctx, fr);
    if (rc) return rc;
    fb_drain(ctx);
    int w, h;
    if (level_dims(fr->width, fr->height, level, &w, &h))
        return fail(ctx, MRGINGHAM_AMD_ERR_ARG, "Got an unreasonable image_pyramid_level = %d", level);
    if (fr->nframes == 0) return 0;
    if (!d_points || !d_levels || !d_npoints || points_pitch <= 0)
        return fail(ctx, MRGINGHAM_AMD_ERR_ARG, "NULL point buffers");
    MRG_HIP_CHECK(hipSetDevice(ctx->device));
    if ((rc = choose_sets(ctx, fr))) return rc;
    if ((rc = ensure_level(ctx, level, fr->nframes, fr->width, fr->height, points_pitch))) return rc;
    if ((rc = ensure_points(ctx, fr->nframes, points_pitch))) return rc;
    begin_op(ctx, level);
    if (level > 0) {
        launch_one_level_image(fr, level, (uint8_t*)cur_levels(ctx)[level].img.p, w, h, ctx->pix);
    }
    const LevelBatch lb = queue_level_chess(ctx, fr, level);
    auto& ps = ctx->pts[ctx->cur];
    RefineIO io{d_points, d_levels, d_npoints, points_pitch, d_nrefined, (int32_t*)ps.leader.p,
                (int32_t*)ps.need.p, (int32_t*)ps.nseeds.p, (uint32_t*)ps.seeds.p, (int32_t*)ps.sroot.p};
    const size_t np = (size_t)fr->nframes * points_pitch;
    order_after_previous(ctx, {{(const char*)d_points, np * 16}, {(const char*)d_levels, np},
                               {(const char*)d_nrefined, d_nrefined ? (size_t)fr->nframes * 4 : 0}},
                         {{(const char*)d_npoints, (size_t)fr->nframes * 4}});
    MRG_HIP_CHECK(hipStreamWaitEvent(cur_cc(ctx), ctx->ev_pix[level], 0));
    launch_cc_refine(lb, tables_of(ctx, level), level, io, 0, fr->nframes, cur_cc(ctx));
    end_op(ctx);
    MRG_HIP_CHECK(hipGetLastError());
    return 0;
}

int mrgingham_amd_chain_batch(mrgingham_amd_ctx* ctx, const mrgingham_amd_frames* fr, int start_level,
                              double* d_points, signed char* d_levels, int32_t* d_npoints, int points_pitch) {
    int rc = validate_frames(ctx, fr);
    if (rc) return rc;
    fb_drain(ctx);
    int w, h;
    if (level_dims(fr->width, fr->height, start_level, &w, &h))
        return fail(ctx, MRGINGHAM_AMD_ERR_ARG, "Got an unreasonable image_pyramid_level = %d", start_level);
    if (fr->nframes == 0) return 0;
    if (!d_points || !d_levels || !d_npoints || points_pitch <= 0)
        return fail(ctx, MRGINGHAM_AMD_ERR_ARG, "NULL point buffers");
    MRG_HIP_CHECK(hipSetDevice(ctx->device));
    // (option "sparse_refine" 1 = where it pays: the dense response of a small call is cheaper than the longer chain --
    // measured crossover at 80-100 Mpx per call, e.g. 64 x 1280x960 or 8 x 4096x3072; 2 = always)
    const bool sparse_pays = ctx->sparse_refine == 2 || (long long)fr->width * fr->height * fr->nframes >= kSparsePaysPixels;
    const bool sparse_now = ctx->sparse_refine && sparse_pays && start_level >= 1 && start_level <= kRefineLevelsMax &&
                            !ctx->use_v0 && ctx->cc_lds;
    if (sparse_now) ctx->sparse_seen = true;
    if ((rc = choose_sets(ctx, fr))) return rc;
    for (int L = 0; L <= start_level; ++L)
        if ((rc = ensure_level(ctx, L, fr->nframes, fr->width, fr->height, points_pitch))) return rc;
    if ((rc = ensure_points(ctx, fr->nframes, points_pitch))) return rc;
    if (sparse_now && !ctx->sparse_stat.p) {
        if ((rc = ensure(ctx, ctx->sparse_stat, 256))) return rc;
        MRG_HIP_CHECK(hipMemset(ctx->sparse_stat.p, 0, 256));
    }
    begin_op(ctx, start_level);
    auto& ps = ctx->pts[ctx->cur];
    DetectOut out{(int32_t*)ps.cand_xy.p, points_pitch, (int32_t*)ps.cand_counts.p};
    out.points = d_points;
    out.levels = d_levels;
    out.npoints = d_npoints;
    out.points_pitch = points_pitch;
    RefineIO io{d_points, d_levels, d_npoints, points_pitch, nullptr, (int32_t*)ps.leader.p,
                (int32_t*)ps.need.p, (int32_t*)ps.nseeds.p, (uint32_t*)ps.seeds.p, (int32_t*)ps.sroot.p};
    {
        const size_t np = (size_t)fr->nframes * points_pitch;
        order_after_previous(ctx, {{(const char*)d_points, np * 16}, {(const char*)d_levels, np},
                                   {(const char*)d_npoints, (size_t)fr->nframes * 4}}, {});
    }
    LevelBatch lbs[kMaxLevel + 1];
    hipEvent_t lev_ev[kMaxLevel + 1] = {};
    auto note_pending = [&](int L) {
        if (fr->nframes > ctx->pending_frames[ctx->cur][L]) ctx->pending_frames[ctx->cur][L] = fr->nframes;
    };
    if (sparse_now) {
        // SPARSE REFINEMENT.  The dense schedule computes the response of levels start-1 .. 0 for whole frames and then
        // looks at it around ~100 points.  Here: every level image in one pass over the frames (pyramid kernel; the
        // variance windows need them around any peak), the dense response only at the START level (its detection needs
        // every component), and below it, level by level on the component stream: queue_sparse_levels.
        // what is timed in this mode (mrgingham_amd_chess_kernel_ms): the kernel that reads the frames, i.e. the launch
        // that writes the level images (the dominant kernel of a sparse step; 1 B/px read + 0.328 B/px written)
        hipEvent_t e0 = nullptr;
        if (ctx->timing) {
            e0 = timing_event(ctx);
            hipEventRecord(e0, ctx->pix);
        }
        queue_level_images(ctx, fr, start_level, true, true);
        if (e0) {
            hipEvent_t em = timing_event(ctx);
            hipEventRecord(em, ctx->pix);
            ctx->events.emplace_back(e0, em);
        }
        lbs[start_level] = level_batch_of(ctx, fr, start_level);
        launch_chess_any(ctx, lbs[start_level], tables_of(ctx, start_level), fr->nframes, true, true, ctx->pix, false);
        hipEvent_t e1 = ctx->ev_pix[start_level];
        hipEventRecord(e1, ctx->pix);
        note_pending(start_level);
        ctx->last_fused = 0;
        ctx->last_merged = -1;  // (mrgingham_amd_chain_info: a sparse step)
        MRG_HIP_CHECK(hipStreamWaitEvent(cur_cc(ctx), e1, 0));
        launch_cc_detect(lbs[start_level], tables_of(ctx, start_level), start_level, out, 0, fr->nframes, cur_cc(ctx));
        SparseRestore src{out.xy, out.capacity, start_level, nullptr, nullptr};
        if ((rc = queue_sparse_levels(ctx, fr, start_level, io, src))) return rc;
        end_op(ctx);
        MRG_HIP_CHECK(hipGetLastError());
        return 0;
    }
    // pixel stream.  Frames of whole 16 x 8 blocks (every BASELINE size): level 0 first, its kernel also
    // writes the level images 1..3 out of the rows it holds in LDS anyway, so the batch is read from HBM
    // once instead of twice; the small levels follow.  Other shapes: every level image in one pass over
    // the frames (pyramid kernel), then the responses top-down.
    lbs[0] = level_batch_of(ctx, fr, 0);
    const bool fused = ctx->fuse_pyramid && !ctx->use_v0 && start_level >= 1 && ctx->multi_level != 2 &&
                       chess_pyramid_ok(lbs[0], fr->nframes);
    queue_level_images(ctx, fr, start_level, !fused);
    if (fused) {
        hipEvent_t e0 = nullptr;
        if (ctx->timing) {
            e0 = timing_event(ctx);
            hipEventRecord(e0, ctx->pix);
        }
#ifdef MRG_EXPERIMENT
        if (!((ctx->chess_variant_hot & 32) && launch_chess16_pyramid(lbs[0], tables_of(ctx, 0), pyramid_out_of(ctx, start_level), fr->nframes, ctx->pix)))
#endif
            launch_chess_pyramid(lbs[0], tables_of(ctx, 0), pyramid_out_of(ctx, start_level), fr->nframes, ctx->pix, ctx->chess_seg);
        if (e0) {
            hipEvent_t e1 = timing_event(ctx);
            hipEventRecord(e1, ctx->pix);
            ctx->events.emplace_back(e0, e1);
            lev_ev[0] = e1;
        }  // else: the event behind level 1 stands in (the component chain reaches level 0 last anyway)
        note_pending(0);
    }
    // levels 3 (or the top), 2, 1 -- or all of them, level 0 included -- in one launch when the shapes
    // allow it.  Every hipEventRecord on the pixel stream is a packet of its own between two kernels
    // (~4 us each in the kernel trace), so a boundary gets ONE: the levels of a merged launch share an
    // event, and with kernel timing on the timing marks double as the hand-over events.
    bool merged = false;
    hipEvent_t before_l0 = nullptr;  // timing mode: an event recorded right before the level-0 launch, if there is one
    const int top = start_level < 3 ? start_level : 3;
    const int lowest = ctx->multi_level == 2 ? 0 : 1;  // lowest level inside the merged launch
    if (top - lowest >= 1 && !ctx->use_v0 && ctx->multi_level) {
        LevelBatch mlb[4];
        CompTables mt[4];
        int n = 0;
        for (int L = lowest; L <= top; ++L, ++n) {  // largest level first
            mlb[n] = level_batch_of(ctx, fr, L);
            mt[n] = tables_of(ctx, L);
        }
        // decided BEFORE anything is queued: a level must not be appended to its hot list twice
        if (chess_multi_ok(mlb, n, fr->nframes)) {
            for (int L = start_level; L > top; --L) {
                lbs[L] = queue_level_chess(ctx, fr, L);
                lev_ev[L] = ctx->ev_pix[L];
            }
            hipEvent_t e0 = nullptr;
            if (lowest == 0 && ctx->timing) {
                e0 = timing_event(ctx);
                hipEventRecord(e0, ctx->pix);
            }
            merged =
#ifdef MRG_EXPERIMENT
                ((ctx->chess_variant_hot & 16) && launch_chess16_multi(mlb, mt, n, fr->nframes, ctx->pix)) ||
#endif
                launch_chess_multi(mlb, mt, n, fr->nframes, ctx->pix, ctx->chess_seg);
            if (merged) {
                hipEvent_t em = (ctx->timing && !fused) ? timing_event(ctx) : ctx->ev_pix[top];
                hipEventRecord(em, ctx->pix);
                if (e0) ctx->events.emplace_back(e0, em);  // all levels in one launch: that launch is what is timed
                else if (ctx->timing) before_l0 = em;
                for (int L = top; L >= lowest; --L) {
                    lbs[L] = mlb[L - lowest];
                    lev_ev[L] = em;
                    note_pending(L);
                }
            } else if (e0) {
                ctx->event_pool.push_back(e0);
            }
        }
    }
    for (int L = merged ? lowest - 1 : start_level; L >= 1; --L) {
        lbs[L] = queue_level_chess(ctx, fr, L);
        lev_ev[L] = ctx->ev_pix[L];
    }
    if (fused) {
        if (!lev_ev[0]) lev_ev[0] = lev_ev[1];
    } else if (!(merged && lowest == 0)) {  // level 0 on its own
        const CompTables t0 = tables_of(ctx, 0);
        if (ctx->timing) {
            hipEvent_t e0 = before_l0;
            if (!e0) {
                e0 = timing_event(ctx);
                hipEventRecord(e0, ctx->pix);
            }
            launch_chess_any(ctx, lbs[0], t0, fr->nframes, true, true, ctx->pix, false);
            hipEvent_t e1 = timing_event(ctx);
            hipEventRecord(e1, ctx->pix);
            ctx->events.emplace_back(e0, e1);
            lev_ev[0] = e1;
        } else {
            launch_chess_any(ctx, lbs[0], t0, fr->nframes, true, true, ctx->pix, false);
            hipEventRecord(ctx->ev_pix[0], ctx->pix);
            lev_ev[0] = ctx->ev_pix[0];
        }
        note_pending(0);
    }
    ctx->last_fused = fused;
    ctx->last_merged = merged ? top - lowest + 1 : 0;
    // component stream: detect at the top (mrgingham.cc:50), candidates -> corners
    // (find_grid.cc:353-354), then refine level by level (mrgingham.cc:87-99)
    // which pixel-stream event a level's search waits for: level 0 runs LAST on the pixel stream in the
    // classic order (cc_schedule 1 holds levels 1 and 0 back until then, 2 holds everything back) and
    // FIRST in the fused order (then level 1 is the last)
    auto gate_of = [&](int L) {
        if (fused) return lev_ev[ctx->cc_schedule == 2 ? 1 : (L > 1 ? L : 1)];
        return lev_ev[(ctx->cc_schedule == 1 && L <= 1) || ctx->cc_schedule == 2 ? 0 : L];
    };
    const bool no_cc = (ctx->cc_lds & 128) != 0;  // timing experiment only (tools/interference_ab.py): pixel kernels alone
    MRG_HIP_CHECK(hipStreamWaitEvent(cur_cc(ctx), gate_of(start_level), 0));
    if (!no_cc)
        launch_cc_detect(lbs[start_level], tables_of(ctx, start_level), start_level, out, 0, fr->nframes, cur_cc(ctx));
    for (int L = start_level - 1; L >= 0; --L) {
        MRG_HIP_CHECK(hipStreamWaitEvent(cur_cc(ctx), gate_of(L), 0));
        if (!no_cc) launch_cc_refine(lbs[L], tables_of(ctx, L), L, io, 0, fr->nframes, cur_cc(ctx));
    }
    end_op(ctx);
    MRG_HIP_CHECK(hipGetLastError());
    return 0;
}

int mrgingham_amd_cc_on_response_batch(mrgingham_amd_ctx* ctx, const int16_t* d_response,
                                       const uint8_t* d_level_image, int nframes, int w, int h, int level,
                                       int32_t* d_xy, int capacity_per_frame, int32_t* d_counts,
                                       double* d_points, signed char* d_levels, const int32_t* d_npoints,
                                       int points_pitch, int32_t* d_nrefined) {
    if (!ctx) return MRGINGHAM_AMD_ERR_ARG;
    fb_drain(ctx);
    const bool detect = d_xy != nullptr, refine = d_points != nullptr;
    if (detect == refine) return fail(ctx, MRGINGHAM_AMD_ERR_ARG, "exactly one of d_xy (detect) and d_points (refine)");
    if (nframes < 0 || w < 0 || h < 0 || w > 32767 || h > 32767 || level < 0 || level > kMaxLevel ||
        (nframes > 0 && (!d_response || !d_level_image)))
        return fail(ctx, MRGINGHAM_AMD_ERR_ARG, "bad response batch descriptor");
    if (detect && (!d_counts || capacity_per_frame < 0)) return fail(ctx, MRGINGHAM_AMD_ERR_ARG, "NULL outputs");
    if (refine && (!d_levels || !d_npoints || points_pitch <= 0))
        return fail(ctx, MRGINGHAM_AMD_ERR_ARG, "NULL point buffers");
    if (nframes == 0) return 0;
    MRG_HIP_CHECK(hipSetDevice(ctx->device));
    // the level-0 scratch of a w x h "frame": the response is the level image's as far as the
    // component search is concerned; `level` only enters through the coordinate scale
    int rc;
    const int pitch = refine ? points_pitch : 0;
    if ((rc = ensure_level(ctx, 0, nframes, w, h, pitch))) return rc;
    if (refine && (rc = ensure_points(ctx, nframes, points_pitch))) return rc;
    begin_op(ctx, 0);
    const LevelScratch& L = cur_levels(ctx)[0];
    LevelBatch lb;
    lb.nframes = nframes;
    lb.w = w;
    lb.h = h;
    lb.img = d_level_image;
    lb.img_pitch = (long long)w * h;
    lb.img_stride = w;
    lb.resp = (int16_t*)L.resp.p;
    lb.resp_pitch = (long long)w * h;
    const CompTables t = tables_of(ctx, 0);
    launch_hot_from_response(d_response, lb, t, 0, nframes, ctx->pix);
    hipEventRecord(ctx->ev_pix[0], ctx->pix);
    if (nframes > ctx->pending_frames[ctx->cur][0]) ctx->pending_frames[ctx->cur][0] = nframes;
    if (detect) {
        order_after_previous(ctx, {{(const char*)d_xy, (size_t)nframes * capacity_per_frame * 8},
                                   {(const char*)d_counts, (size_t)nframes * 4}}, {});
        MRG_HIP_CHECK(hipStreamWaitEvent(cur_cc(ctx), ctx->ev_pix[0], 0));
        launch_cc_detect(lb, t, level, DetectOut{d_xy, capacity_per_frame, d_counts}, 0, nframes, cur_cc(ctx));
    } else {
        auto& ps = ctx->pts[ctx->cur];
        RefineIO io{d_points, d_levels, d_npoints, points_pitch, d_nrefined, (int32_t*)ps.leader.p,
                    (int32_t*)ps.need.p, (int32_t*)ps.nseeds.p, (uint32_t*)ps.seeds.p, (int32_t*)ps.sroot.p};
        const size_t np = (size_t)nframes * points_pitch;
        order_after_previous(ctx, {{(const char*)d_points, np * 16}, {(const char*)d_levels, np},
                                   {(const char*)d_nrefined, d_nrefined ? (size_t)nframes * 4 : 0}},
                             {{(const char*)d_npoints, (size_t)nframes * 4}});
        MRG_HIP_CHECK(hipStreamWaitEvent(cur_cc(ctx), ctx->ev_pix[0], 0));
        launch_cc_refine(lb, t, level, io, 0, nframes, cur_cc(ctx));
    }
    end_op(ctx);
    MRG_HIP_CHECK(hipGetLastError());
    return 0;
}

/* ------------------------------------------------------------------------ */
/* Reference symbols: host buffers in, host results out                      */
/* ------------------------------------------------------------------------ */

// Single-frame context on the same device as `ctx` (see mrgingham_amd_ctx::one).
static mrgingham_amd_ctx* same_device_ctx(mrgingham_amd_ctx* ctx) {
    if (!ctx->one) {
        ctx->one = mrgingham_amd_create(ctx->device);
        if (ctx->one) ctx->one->cap_shift = ctx->cap_shift;
        hipSetDevice(ctx->device);
    }
    return ctx->one;
}

// Which device the k-th thread that calls a reference symbol gets when nobody said otherwise: MRGINGHAM_AMD_DEVICE
// (every thread on that device) or, with the variable unset, k modulo the number of devices -- the reference's own
// parallelism is N worker threads with image i on worker i % N (mrgingham-from-image.cc:50, :374-379), and mapped this
// way its workers spread over the GPUs of a node by themselves.
static std::atomic<int> g_thread_counter{0};
struct ThreadCtxHolder {
    mrgingham_amd_ctx* ctx = nullptr;
    int requested = -1;  // mrgingham_amd_set_thread_device
    ~ThreadCtxHolder() { /* leaked on purpose: HIP may already be torn down at thread exit */ }
};
static thread_local ThreadCtxHolder t_holder;

static mrgingham_amd_ctx* thread_ctx() {
    // One context per calling thread: the reference is called from N worker
    // pthreads at once (mrgingham-from-image.cc:374-379).
    ThreadCtxHolder& h = t_holder;
    if (!h.ctx) {
        int dev = h.requested;
        const bool counted = dev < 0;
        if (counted) dev = mrgingham_amd_device_for_thread(g_thread_counter.fetch_add(1), mrgingham_amd_device_count(),
                                                           getenv("MRGINGHAM_AMD_DEVICE"));
        h.ctx = mrgingham_amd_create(dev);
        if (!h.ctx && counted) g_thread_counter.fetch_sub(1);  // a slot of the round-robin is used by a context, not by an attempt
    }
    return h.ctx;
}

int mrgingham_amd_device_for_thread(int thread_index, int ndevices, const char* env_value) {
    if (env_value && *env_value) return atoi(env_value);
    if (ndevices <= 0) return 0;
    return (int)((unsigned)(thread_index < 0 ? 0 : thread_index) % (unsigned)ndevices);
}

int mrgingham_amd_set_thread_device(int device_ordinal) {
    const int ndev = mrgingham_amd_device_count();
    if (device_ordinal < 0 || device_ordinal >= ndev) {
        fprintf(stderr, "mrgingham_amd: device ordinal %d out of range (%d device(s))\n", device_ordinal, ndev);
        return MRGINGHAM_AMD_ERR_ARG;
    }
    ThreadCtxHolder& h = t_holder;
    h.requested = device_ordinal;
    if (h.ctx && h.ctx->device != device_ordinal) {
        mrgingham_amd_destroy(h.ctx);
        h.ctx = nullptr;
    }
    return MRGINGHAM_AMD_OK;
}

int mrgingham_amd_thread_device(void) {
    CallerDevice caller_device_;
    mrgingham_amd_ctx* ctx = thread_ctx();
    return ctx ? ctx->device : -1;
}

void* mrgingham_amd_host_alloc(size_t bytes) {
    void* p = nullptr;
    if (bytes == 0 || hipHostMalloc(&p, bytes, hipHostMallocPortable) != hipSuccess) return nullptr;
    return p;
}
void mrgingham_amd_host_free(void* p) {
    if (p) hipHostFree(p);
}
int mrgingham_amd_host_register(void* p, size_t bytes) {
    if (!p || bytes == 0) return MRGINGHAM_AMD_ERR_ARG;
    return hipHostRegister(p, bytes, hipHostRegisterPortable) == hipSuccess ? MRGINGHAM_AMD_OK : MRGINGHAM_AMD_ERR_DEVICE;
}
int mrgingham_amd_host_unregister(void* p) {
    if (!p) return MRGINGHAM_AMD_ERR_ARG;
    return hipHostUnregister(p) == hipSuccess ? MRGINGHAM_AMD_OK : MRGINGHAM_AMD_ERR_DEVICE;
}

int mrgingham_amd_set_wait_policy(int policy) {
    unsigned flag;
    switch (policy) {
        case 0: flag = hipDeviceScheduleAuto; break;
        case 1: flag = hipDeviceScheduleSpin; break;
        case 2: flag = hipDeviceScheduleYield; break;
        case 3: flag = hipDeviceScheduleBlockingSync; break;
        default: return MRGINGHAM_AMD_ERR_ARG;
    }
    int ndev = 0, prev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return MRGINGHAM_AMD_ERR_DEVICE;
    hipGetDevice(&prev);
    int rc = MRGINGHAM_AMD_OK;
    for (int d = 0; d < ndev; ++d)
        if (hipSetDevice(d) != hipSuccess || hipSetDeviceFlags(flag) != hipSuccess) rc = MRGINGHAM_AMD_ERR_DEVICE;
    hipSetDevice(prev);
    (void)hipGetLastError();
    return rc;
}

int mrgingham_amd_shard_range(int total, int k, int n, int* first, int* count) {
    if (total < 0 || n <= 0 || k < 0 || k >= n || !first || !count) return MRGINGHAM_AMD_ERR_ARG;
    const int q = total / n, r = total % n;  // the first r shards take one frame more
    *first = k * q + (k < r ? k : r);
    *count = q + (k < r ? 1 : 0);
    return MRGINGHAM_AMD_OK;
}

// One shard of mrgingham_amd_chain_multi: the chain on its context and, for a shard that is not on the root device, the
// copy of its block to the root behind it.  Runs on the context's submit thread (or on the caller for a single shard).
static int chain_multi_shard(mrgingham_amd_ctx* ctx, int root_device, const mrgingham_amd_frames* shard, int start_level,
                             double* dst_p, signed char* dst_l, int32_t* dst_n, int points_pitch) {
    const int B = shard->nframes;
    const size_t np = (size_t)B * points_pitch;
    int rc;
    if (ctx->device == root_device) {
        if ((rc = mrgingham_amd_chain_batch(ctx, shard, start_level, dst_p, dst_l, dst_n, points_pitch))) return rc;
        ctx->mg_pending = false;
        return MRGINGHAM_AMD_OK;
    }
    MRG_HIP_CHECK(hipSetDevice(ctx->device));
    if (!ctx->mg_stream) {
        MRG_HIP_CHECK(hipStreamCreateWithFlags(&ctx->mg_stream, hipStreamNonBlocking));
        MRG_HIP_CHECK(hipEventCreateWithFlags(&ctx->mg_done, hipEventDisableTiming));
        int can = 0;  // direct peer copies where the link allows them (otherwise HIP stages through the host)
        if (hipDeviceCanAccessPeer(&can, ctx->device, root_device) == hipSuccess && can) {
            hipError_t e = hipDeviceEnablePeerAccess(root_device, 0);
            if (e != hipSuccess) (void)hipGetLastError();  // (already enabled, or refused: the copy still works)
        }
    }
    if ((rc = ensure(ctx, ctx->mg_pts, np * 16)) || (rc = ensure(ctx, ctx->mg_lv, np)) || (rc = ensure(ctx, ctx->mg_np, (size_t)B * 4)))
        return rc;
    if (ctx->mg_pending) MRG_HIP_CHECK(hipStreamWaitEvent(ctx->pix, ctx->mg_done, 0));  // the gather before this one has read the buffers
    if ((rc = mrgingham_amd_chain_batch(ctx, shard, start_level, (double*)ctx->mg_pts.p, (signed char*)ctx->mg_lv.p,
                                        (int32_t*)ctx->mg_np.p, points_pitch)))
        return rc;
    if ((rc = mrgingham_amd_stream_wait(ctx, ctx->mg_stream))) return rc;
    MRG_HIP_CHECK(hipMemcpyPeerAsync(dst_p, root_device, ctx->mg_pts.p, ctx->device, np * 16, ctx->mg_stream));
    MRG_HIP_CHECK(hipMemcpyPeerAsync(dst_l, root_device, ctx->mg_lv.p, ctx->device, np, ctx->mg_stream));
    MRG_HIP_CHECK(hipMemcpyPeerAsync(dst_n, root_device, ctx->mg_np.p, ctx->device, (size_t)B * 4, ctx->mg_stream));
    MRG_HIP_CHECK(hipEventRecord(ctx->mg_done, ctx->mg_stream));
    ctx->mg_pending = true;
    return MRGINGHAM_AMD_OK;
}

/* chain_batch over several contexts -- one per device of a node, or several on one -- in ONE call: context k takes
 * shards[k] (frames in the memory of ITS device), and the corner lists of every shard arrive in d_points / d_levels /
 * d_npoints, buffers on the device of ctxs[0] laid out for the sum of the shards' frames in shard order (frame-major):
 * a shard on that device writes its block in place, a shard elsewhere writes into its own context's buffers and the
 * block travels device to device behind its chain (hipMemcpyPeerAsync: xGMI between the GPUs of a node) -- the ONE
 * exchange of the path.  Asynchronous; mrgingham_amd_sync_multi waits for everything.
 * Every shard is queued by a submit thread of its own context, all at once (queueing one chain costs the host ~70 us:
 * eight of them from one thread would be 0.56 ms per call, more than a sparse step takes on the device); the call returns
 * when all of them are queued. */
int mrgingham_amd_chain_multi(mrgingham_amd_ctx* const* ctxs, int nctx, const mrgingham_amd_frames* shards, int start_level,
                              double* d_points, signed char* d_levels, int32_t* d_npoints, int points_pitch) {
    if (!ctxs || nctx <= 0 || !shards || !ctxs[0]) return MRGINGHAM_AMD_ERR_ARG;
    const CallerDevice keep;  // (the shards' contexts live on several devices: the caller's current one is put back)
    mrgingham_amd_ctx* root = ctxs[0];
    if (!d_points || !d_levels || !d_npoints || points_pitch <= 0)
        return fail(root, MRGINGHAM_AMD_ERR_ARG, "NULL point buffers");
    for (int k = 0; k < nctx; ++k) {
        if (!ctxs[k]) return fail(root, MRGINGHAM_AMD_ERR_ARG, "NULL context %d", k);
        for (int j = 0; j < k; ++j)
            if (ctxs[j] == ctxs[k]) return fail(root, MRGINGHAM_AMD_ERR_ARG, "context %d is context %d again: one context per shard", k, j);
    }
    int nwork = 0;
    for (int k = 0; k < nctx; ++k) {
        const int rc = validate_frames(ctxs[k], &shards[k]);
        if (rc) return rc;
        nwork += shards[k].nframes > 0;
    }
    std::vector<int> rcs((size_t)nctx, MRGINGHAM_AMD_OK);
    std::vector<char> started((size_t)nctx, 0);
    const int root_device = root->device;
    size_t off = 0;  // frames in front of shard k
    for (int k = 0; k < nctx; ++k) {
        mrgingham_amd_ctx* ctx = ctxs[k];
        const int B = shards[k].nframes;
        if (B == 0) continue;
        double* dst_p = d_points + off * points_pitch * 2;
        signed char* dst_l = d_levels + off * points_pitch;
        int32_t* dst_n = d_npoints + off;
        off += (size_t)B;
        const mrgingham_amd_frames* sh = &shards[k];
        int* out = &rcs[(size_t)k];
        if (nwork == 1) {
            *out = chain_multi_shard(ctx, root_device, sh, start_level, dst_p, dst_l, dst_n, points_pitch);
        } else {
            ctx->submit_pool.start(1, [=] { *out = chain_multi_shard(ctx, root_device, sh, start_level, dst_p, dst_l, dst_n, points_pitch); });
            started[(size_t)k] = 1;
        }
    }
    int rc = MRGINGHAM_AMD_OK;
    for (int k = 0; k < nctx; ++k) {
        if (started[(size_t)k]) ctxs[k]->submit_pool.wait();
        if (rcs[(size_t)k] && !rc) rc = rcs[(size_t)k];
    }
    (void)hipSetDevice(root_device);
    return rc;
}

/* The one exchange of the path for a host that runs ONE PROCESS PER GPU (rccl.h:745): ncclGather of this
 * rank's packed corner lists to `root`, on `stream`, behind the context's most recent call.  RCCL is not linked: the
 * communicator was made by the RCCL the host process runs on, and its ncclGather is the one that has to be called -- looked
 * up in the process (dlsym), then in librccl.so.1 / librccl.so. */
int mrgingham_amd_packed_layout(int nframes, int points_pitch, size_t* off_levels, size_t* off_npoints, size_t* bytes) {
    if (nframes < 0 || points_pitch <= 0) return MRGINGHAM_AMD_ERR_ARG;
    const size_t np = (size_t)nframes * points_pitch;
    const size_t o_lv = np * 16, o_np = (o_lv + np + 7) / 8 * 8;
    if (off_levels) *off_levels = o_lv;
    if (off_npoints) *off_npoints = o_np;
    if (bytes) *bytes = (o_np + (size_t)nframes * 4 + 7) / 8 * 8;  // (a multiple of 8: rank blocks of the gathered buffer stay aligned)
    return MRGINGHAM_AMD_OK;
}

int mrgingham_amd_gather_rccl(mrgingham_amd_ctx* ctx, void* nccl_comm, int root, const void* d_packed, size_t bytes,
                              void* d_gathered, void* stream) {
    if (!ctx) return MRGINGHAM_AMD_ERR_ARG;
    if (!nccl_comm || !d_packed || bytes == 0 || root < 0) return fail(ctx, MRGINGHAM_AMD_ERR_ARG, "gather_rccl: NULL communicator / buffer, or nothing to send");
    // RCCL's ncclGather, looked up in the copy of RCCL the HOST has loaded (the one that made `nccl_comm`): no header and no
    // link dependency, and never a second copy -- a communicator handed to another instance of the library is undefined
    // behaviour.  First among the global symbols (a C host linked with -lrccl), then in an already-loaded librccl that was
    // opened RTLD_LOCAL (Python / PyTorch's bundled copy): dlopen(RTLD_NOLOAD) finds it without loading anything.  A
    // failed lookup is not remembered (the host may load RCCL later).
    using gather_fn = int (*)(const void*, void*, size_t, int /* ncclDataType_t */, int, void* /* ncclComm_t */, hipStream_t);
    using errstr_fn = const char* (*)(int);
    constexpr int kNcclSuccess = 0, kNcclUint8 = 1;  // nccl.h: ncclSuccess, ncclUint8 (stable since NCCL 2.0)
    static std::atomic<gather_fn> gather_cached{nullptr};
    static std::atomic<errstr_fn> errstr_cached{nullptr};
    gather_fn gather = gather_cached.load(std::memory_order_acquire);
    if (!gather) {
        void* f = dlsym(RTLD_DEFAULT, "ncclGather");
        void* e = dlsym(RTLD_DEFAULT, "ncclGetErrorString");
        if (!f)
            for (const char* name : {"librccl.so.1", "librccl.so"}) {
                void* h = dlopen(name, RTLD_NOW | RTLD_NOLOAD);
                if (h && (f = dlsym(h, "ncclGather"))) {
                    e = dlsym(h, "ncclGetErrorString");
                    break;  // (the handle is kept: the library stays mapped as long as this one uses its function)
                }
                if (h) dlclose(h);
            }
        if (!f)
            return fail(ctx, MRGINGHAM_AMD_ERR_DEVICE, "gather_rccl: RCCL is not loaded in this process (no ncclGather among the global symbols, "
                                                       "no librccl.so mapped): the host that made the communicator must have it loaded");
        gather = (gather_fn)f;
        errstr_cached.store((errstr_fn)e, std::memory_order_release);
        gather_cached.store(gather, std::memory_order_release);
    }
    const errstr_fn errstr = errstr_cached.load(std::memory_order_acquire);
    const CallerDevice keep;  // (the caller's current device is put back)
    MRG_HIP_CHECK(hipSetDevice(ctx->device));
    const int rc = mrgingham_amd_stream_wait(ctx, stream);  // the gather starts behind the chain that fills d_packed, on the device
    if (rc) return rc;
    const int r = gather(d_packed, d_gathered, bytes, kNcclUint8, root, nccl_comm, (hipStream_t)stream);
    if (r != kNcclSuccess)
        return fail(ctx, MRGINGHAM_AMD_ERR_DEVICE, "ncclGather failed: %s", errstr ? errstr(r) : "(no error text)");
    return MRGINGHAM_AMD_OK;
}

int mrgingham_amd_sync_multi(mrgingham_amd_ctx* const* ctxs, int nctx) {
    if (!ctxs || nctx <= 0) return MRGINGHAM_AMD_ERR_ARG;
    const CallerDevice keep;  // (the contexts live on several devices: the caller's current one is put back)
    int rc = MRGINGHAM_AMD_OK;
    for (int k = 0; k < nctx; ++k) {
        mrgingham_amd_ctx* ctx = ctxs[k];
        if (!ctx) return MRGINGHAM_AMD_ERR_ARG;
        const int r = mrgingham_amd_sync(ctx);
        if (r && !rc) rc = r;
        if (ctx->mg_stream) {
            MRG_HIP_CHECK(hipSetDevice(ctx->device));
            MRG_HIP_CHECK(hipStreamSynchronize(ctx->mg_stream));
        }
        ctx->mg_pending = false;
    }
    return rc;
}

/* Device-side alternative to mrgingham_amd_sync_multi: `stream` (a hipStream_t of any device, normally the first
 * context's) waits for the chains and the gathers of the most recent mrgingham_amd_chain_multi. */
int mrgingham_amd_stream_wait_multi(mrgingham_amd_ctx* const* ctxs, int nctx, void* stream) {
    if (!ctxs || nctx <= 0) return MRGINGHAM_AMD_ERR_ARG;
    const CallerDevice keep;
    for (int k = 0; k < nctx; ++k) {
        mrgingham_amd_ctx* ctx = ctxs[k];
        if (!ctx) return MRGINGHAM_AMD_ERR_ARG;
        if (ctx->mg_pending) {
            MRG_HIP_CHECK(hipStreamWaitEvent((hipStream_t)stream, ctx->mg_done, 0));
        } else {
            const int r = mrgingham_amd_stream_wait(ctx, stream);
            if (r) return r;
        }
    }
    return MRGINGHAM_AMD_OK;
}

// Upload one host frame as a dense device image; fills `fr`.
static int upload_frame(mrgingham_amd_ctx* ctx, const void* host, int rows, int cols, int stride,
                        mrgingham_amd_frames* fr) {
    int rc;
    if ((rc = ensure(ctx, ctx->io_frame, (size_t)rows * cols + 64))) return rc;
    // stream-ordered on streams[0]: the kernels that read it are queued on the same stream
    // (a dense frame as ONE copy: the 2-D form goes through a slower path of the runtime even when the rows are contiguous)
    if (rows > 0 && cols > 0) {
        if (stride == cols)
            MRG_HIP_CHECK(hipMemcpyAsync(ctx->io_frame.p, host, (size_t)rows * cols, hipMemcpyHostToDevice, ctx->pix));
        else
            MRG_HIP_CHECK(copy_rows_async(ctx->io_frame.p, cols, host, stride, cols, rows, hipMemcpyHostToDevice,
                                           ctx->pix));
    }
    fr->frames = (const uint8_t*)ctx->io_frame.p;
    fr->frame_pitch = (int64_t)rows * cols;
    fr->nframes = 1;
    fr->width = cols;
    fr->height = rows;
    fr->stride = cols;
    return 0;
}

// The reference's --debug dumps of one detector / refinement pass (find_chessboard_corners.cc:282-315,
// :453-459, :513-541): the level image, the ChESS response normalised to 0..255 (raw, and with the
// negatives clamped), and a self-plotting vnlog of the corners.  Same file names, same messages.  The
// response PNGs follow cv::normalize(.., 0, 255, NORM_MINMAX) on CV_16S (single-precision scale and
// shift, round half to even) and imwrite's saturating conversion to 8 bit.
static void write_debug_dumps(mrgingham_amd_ctx* ctx, const mrgingham_amd_frames* fr1, int level, bool refinement,
                              const char* debug_image_filename, const double* pts_xy, int npts) {
    int w, h;
    if (level_dims(fr1->width, fr1->height, level, &w, &h) || w <= 0 || h <= 0) return;
    const size_t n = (size_t)w * h;
    char name[300];
    std::vector<uint8_t> img8(n);
    std::vector<int16_t> resp(n);
    if (ensure(ctx, ctx->dbg_img, n + 64) || ensure(ctx, ctx->dbg_resp, n * 2 + 64)) return;
    if (!refinement) {  // apply_image_pyramid_scaling dumps once per detector call (:453-459)
        if (mrgingham_amd_decimate_batch(ctx, fr1, level, (uint8_t*)ctx->dbg_img.p, ctx->pix) ||
            hipMemcpyAsync(img8.data(), ctx->dbg_img.p, n, hipMemcpyDeviceToHost, ctx->pix) != hipSuccess ||
            hipStreamSynchronize(ctx->pix) != hipSuccess)
            return;
        snprintf(name, sizeof(name), "/tmp/mrgingham-scaled-processed-level%d.png", level);
        if (write_png_gray8(name, img8.data(), w, h)) fprintf(stderr, "Wrote scaled,processed image to %s\n", name);
    }
    for (int positive = 0; positive < 2; ++positive) {
        if (mrgingham_amd_chess_response_batch(ctx, fr1, level, positive, (int16_t*)ctx->dbg_resp.p, ctx->pix) ||
            hipMemcpyAsync(resp.data(), ctx->dbg_resp.p, n * 2, hipMemcpyDeviceToHost, ctx->pix) != hipSuccess ||
            hipStreamSynchronize(ctx->pix) != hipSuccess)
            return;
        int lo = 32767, hi = -32768;
        for (int16_t v : resp) { lo = v < lo ? v : lo; hi = v > hi ? v : hi; }
        const double scale = 255.0 * (hi - lo > 2.220446049250313e-16 ? 1.0 / (double)(hi - lo) : 0.0);
        const double shift = 0.0 - (double)lo * scale;
        const float a = (float)scale, b = (float)shift;
        for (size_t i = 0; i < n; ++i) {
            const float prod = (float)resp[i] * a;
            const float r = rintf(prod + b);
            img8[i] = (uint8_t)(r < 0.f ? 0.f : (r > 255.f ? 255.f : r));
        }
        snprintf(name, sizeof(name), "/tmp/mrgingham-chess-response%s-level%d%s.png", refinement ? "-refinement" : "", level,
                 positive ? "-positive" : "");
        if (write_png_gray8(name, img8.data(), w, h))
            fprintf(stderr, positive ? "Wrote positive-only, normalized ChESS response to %s\n"
                                     : "Wrote a normalized ChESS response to %s\n", name);
    }
    if (refinement) snprintf(name, sizeof(name), "/tmp/mrgingham-1-corners-refinement-level%d.vnl", level);
    else snprintf(name, sizeof(name), "/tmp/mrgingham-1-corners.vnl");
    fprintf(stderr, "Writing self-plotting corner dump to %s\n", name);
    FILE* fp = fopen(name, "w");
    if (!fp) return;
    if (debug_image_filename)
        fprintf(fp, "#!/usr/bin/feedgnuplot --dom --with 'points pt 7 ps 2' --square --image %s\n", debug_image_filename);
    else
        fprintf(fp, "#!/usr/bin/feedgnuplot --dom --square --set 'yr [:] rev'\n");
    fprintf(fp, "# x y\n");
    for (int i = 0; i < npts; ++i) fprintf(fp, "%f %f\n", pts_xy[2 * i], pts_xy[2 * i + 1]);
    fclose(fp);
}

// Every candidate of ONE frame that already lives on the device (dense or strided), with the retry of
// the reference-symbol wrappers: a frame whose hot list or candidate table overflows the default
// capacity is re-run with one table entry per pixel.  Returns false on a device / argument error.
static bool detect_one_frame_all(mrgingham_amd_ctx* ctx, const mrgingham_amd_frames* fr1, int level,
                                 std::vector<int32_t>& xy, int32_t* count_out, bool debug = false,
                                 const char* debug_image_filename = nullptr) {
    const int saved_shift = ctx->cap_shift;
    bool ok = false;
    int32_t count = 0;
    for (int attempt = 0; attempt < 4 && !ok; ++attempt) {
        if (ensure_level(ctx, level, 1, fr1->width, fr1->height, 0) || ensure_points(ctx, 1, 1)) break;
        const int cap = ctx->lvs[0][level].cand_cap;
        if (ensure(ctx, ctx->io_out, (size_t)cap * 8 + 64) || ensure(ctx, ctx->io_counts, 64)) break;
        if (mrgingham_amd_detect_batch(ctx, fr1, level, (int32_t*)ctx->io_out.p, cap, (int32_t*)ctx->io_counts.p)) break;
        // The count and the first candidates follow the search on its own stream into page-locked memory: one wait for
        // that stream instead of a full synchronisation with its status read-back and two blocking copies (3 x 15-20 us
        // of a 0.4 ms call).  A frame whose tables overflowed says so in its count (-1): only then the status words are
        // read, the tables grow and the call is made again.
        constexpr int kFast = 4000;  // candidates that travel with the count
        if (!ctx->io_res_pin && hipHostMalloc(&ctx->io_res_pin, 64 + (size_t)kFast * 8, hipHostMallocDefault) != hipSuccess) {
            ctx->io_res_pin = nullptr;
            break;
        }
        int32_t* pin_count = (int32_t*)ctx->io_res_pin;
        int32_t* pin_xy = pin_count + 16;
        hipStream_t cc = ctx->ccs[ctx->cur];
        const int nfast = cap < kFast ? cap : kFast;
        if (hipMemcpyAsync(pin_count, ctx->io_counts.p, sizeof(int32_t), hipMemcpyDeviceToHost, cc) != hipSuccess ||
            hipMemcpyAsync(pin_xy, ctx->io_out.p, (size_t)nfast * 8, hipMemcpyDeviceToHost, cc) != hipSuccess ||
            hipStreamSynchronize(cc) != hipSuccess)
            break;
        count = *pin_count;
        if (count < 0) {
            const int rc = mrgingham_amd_sync(ctx);
            if (rc == MRGINGHAM_AMD_ERR_CAPACITY && attempt < 3) {
                // the tables have grown to what the frame asked for (mrgingham_amd_sync); the last retry takes a
                // table entry for every pixel (adversarial texture)
                if (attempt == 2) ctx->cap_shift = 0;
                continue;
            }
            break;  // (a count of -1 with nothing to grow: a device error)
        }
        xy.resize((size_t)count * 2);
        if (count > 0) memcpy(xy.data(), pin_xy, (size_t)(count < nfast ? count : nfast) * 8);
        if (count > nfast &&
            hipMemcpy(xy.data() + (size_t)nfast * 2, (const int32_t*)ctx->io_out.p + (size_t)nfast * 2, (size_t)(count - nfast) * 8,
                      hipMemcpyDeviceToHost) != hipSuccess)
            break;
        ok = true;
    }
    ctx->cap_shift = saved_shift;
    *count_out = count;
    if (ok && debug) {
        // the dump lists the corners in full-resolution pixels (:346-348); from the *1000 integers here,
        // i.e. to three decimals
        std::vector<double> p((size_t)(count > 0 ? count : 0) * 2);
        for (size_t i = 0; i < p.size(); ++i) p[i] = (double)xy[i] / kGridScale;
        write_debug_dumps(ctx, fr1, level, false, debug_image_filename, p.data(), count > 0 ? count : 0);
    }
    return ok;
}

void mrgingham_ChESS_response_5(int16_t* response, const uint8_t* image, int w, int h, int stride) {
    if (w < 15 || h < 15) return;  // no interior: the reference's loops do not execute (ChESS.c:62-63)
    CallerDevice caller_device_;
    mrgingham_amd_ctx* ctx = thread_ctx();
    if (!ctx || !response || !image) {
        fprintf(stderr, "mrgingham_amd: mrgingham_ChESS_response_5: no device context; response not written\n");
        return;
    }
    hipSetDevice(ctx->device);
    mrgingham_amd_frames fr;
    if (upload_frame(ctx, image, h, w, stride, &fr)) return;
    if (ensure(ctx, ctx->io_out, (size_t)w * h * 2 + 64)) return;
    if (mrgingham_amd_chess_response_batch(ctx, &fr, 0, 0, (int16_t*)ctx->io_out.p, ctx->pix)) return;
    // interior only, like the reference: the 7-pixel frame of `response` is not touched
    hipError_t e = hipSuccess;
    const size_t bytes = (size_t)w * h * 2;
    {
        // The strided copy of the interior into pageable memory goes through a slow path of the runtime (2.4 ms per 12 MP
        // frame, all of it this copy) that also serialises the threads of a process (hipMemcpy2DAsync: sixteen workers
        // of the command-line tool ran at an eighth of their rate behind one such copy per image).  Instead: whole rows
        // in plain copies into page-locked staging of the context, at the speed of the link, and -- for large frames
        // (12 MP: 25 MB back) in four chunks -- a few host threads that move the interior of each row block into the
        // caller's array as soon as the copy that carries it has landed.
        const bool small = bytes < (4u << 20);
        const int kChunks = small ? 1 : 4;
        if (bytes > ctx->io_pin_bytes) {
            if (ctx->io_pin) hipHostFree(ctx->io_pin);
            ctx->io_pin = nullptr;
            ctx->io_pin_bytes = 0;
            if (hipHostMalloc(&ctx->io_pin, bytes + bytes / 8, hipHostMallocDefault) != hipSuccess) ctx->io_pin = nullptr;
            else ctx->io_pin_bytes = bytes + bytes / 8;
        }
        for (int c = 0; c < kChunks; ++c)
            if (!ctx->io_ev[c]) hipEventCreateWithFlags(&ctx->io_ev[c], hipEventDisableTiming);
        if (!ctx->io_pin || !ctx->io_ev[kChunks - 1]) {
            fprintf(stderr, "mrgingham_amd: ChESS response failed: no page-locked staging\n");
            return;
        }
        const int rows_per = (h + kChunks - 1) / kChunks;
        for (int c = 0; c < kChunks && e == hipSuccess; ++c) {
            const int y0 = c * rows_per, y1 = y0 + rows_per < h ? y0 + rows_per : h;
            if (y1 > y0)
                e = hipMemcpyAsync((char*)ctx->io_pin + (size_t)y0 * w * 2, (const char*)ctx->io_out.p + (size_t)y0 * w * 2,
                                   (size_t)(y1 - y0) * w * 2, hipMemcpyDeviceToHost, ctx->pix);
            if (e == hipSuccess) e = hipEventRecord(ctx->io_ev[c], ctx->pix);
        }
        if (e == hipSuccess) {
            std::atomic<int> next{0};
            std::atomic<int> failed{0};
            const int16_t* pin = (const int16_t*)ctx->io_pin;
            const int device = ctx->device;
            hipEvent_t* evs = ctx->io_ev;
            constexpr int kBlock = 32;  // rows per work item
            const int nblocks = (h - 2 * kMargin + kBlock - 1) / kBlock;
            auto mover = [&]() {
                hipSetDevice(device);
                int waited = -1;  // chunks known to have landed
                for (int b; (b = next.fetch_add(1)) < nblocks;) {
                    const int ya = kMargin + b * kBlock, yb = ya + kBlock < h - kMargin ? ya + kBlock : h - kMargin;
                    const int need = (yb - 1) / rows_per;
                    while (waited < need) {
                        if (hipEventSynchronize(evs[waited + 1]) != hipSuccess) { failed.store(1); return; }
                        ++waited;
                    }
                    for (int y = ya; y < yb; ++y)
                        memcpy(response + (size_t)y * w + kMargin, pin + (size_t)y * w + kMargin, (size_t)(w - 2 * kMargin) * 2);
                }
            };
            int nthreads = (int)std::thread::hardware_concurrency();
            nthreads = nthreads > 8 ? 8 : (nthreads < 1 ? 1 : nthreads);
            if (small) mover();  // (a few hundred KB: the calling thread)
            else ctx->pool.run(nthreads, mover);
            if (failed.load()) e = hipErrorUnknown;
            if (e == hipSuccess) e = hipStreamSynchronize(ctx->pix);
        }
    }
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

// find_blobs_from_image_array (find_blobs.cc:14-46) on a frame that lives on the device as `fr` (one frame) and
// on the host as h_img: candidates as (x, y) * 1000 ints.
static bool blobs_on_device(mrgingham_amd_ctx* ctx, const mrgingham_amd_frames* fr, const uint8_t* h_img, int h_stride,
                            std::vector<int32_t>& xy) {
    if (ensure(ctx, ctx->blob_scratch, blob_scratch_bytes(fr->width, fr->height, nullptr))) return false;
    std::string err;
    auto nodes = [&](size_t bytes) -> void* { return ensure(ctx, ctx->blob_nodes, bytes) ? nullptr : ctx->blob_nodes.p; };
    auto outs = [&](size_t bytes) -> void* { return ensure(ctx, ctx->blob_out, bytes) ? nullptr : ctx->blob_out.p; };
    if (!blob_detect(fr->frames, fr->stride, h_img, h_stride, fr->width, fr->height, ctx->blob_scratch.p, nodes, outs, ctx->pix, xy,
                     err)) {
        fail(ctx, MRGINGHAM_AMD_ERR_CAPACITY, "%s", err.c_str());
        return false;
    }
    return true;
}

bool find_chessboard_corners_from_image_array_C(int Nrows, int Ncols, int stride, char* imagebuffer,
                                                int image_pyramid_level, bool doblobs, bool debug,
                                                bool (*add_points)(int* xy, int N, double scale, void* cookie),
                                                void* cookie) {
    if (Nrows < 0 || Ncols < 0 || stride < Ncols || !imagebuffer || !add_points) return false;
    if (doblobs) {  // bridge.cc:50-55: the blob detector, level 0 only; always "found", possibly with 0 points
        if (image_pyramid_level != 0) return false;
        CallerDevice caller_device_;
        mrgingham_amd_ctx* bctx = thread_ctx();
        if (!bctx) return false;
        hipSetDevice(bctx->device);
        mrgingham_amd_frames bfr;
        std::vector<int32_t> bxy;
        if (upload_frame(bctx, imagebuffer, Nrows, Ncols, stride, &bfr) ||
            !blobs_on_device(bctx, &bfr, (const uint8_t*)imagebuffer, stride, bxy))
            return false;
        int32_t none[2] = {0, 0};
        return (*add_points)(bxy.empty() ? none : bxy.data(), (int)(bxy.size() / 2), 1. / kGridScale, cookie);
    }
    if (!check_level_and_layout(__func__, Nrows, Ncols, stride, image_pyramid_level)) return false;
    CallerDevice caller_device_;
    mrgingham_amd_ctx* ctx = thread_ctx();
    if (!ctx) return false;
    hipSetDevice(ctx->device);
    std::vector<int32_t> xy;
    int32_t count = 0;
    mrgingham_amd_frames fr;
    const bool ok = upload_frame(ctx, imagebuffer, Nrows, Ncols, stride, &fr) == 0 &&
                    detect_one_frame_all(ctx, &fr, image_pyramid_level, xy, &count, debug, nullptr);
    if (!ok || count <= 0) return false;  // bridge.cc:61: nothing found -> false, add_points not called
    return (*add_points)(xy.data(), (int)count, 1. / kGridScale, cookie);  // bridge.cc:66-69
}

// Refinement of host-side points against a frame that already lives on the device (one frame).
static int refine_on_device(mrgingham_amd_ctx* ctx, const mrgingham_amd_frames* fr, double* points_xy,
                            signed char* level, int Npoints, int image_pyramid_level, bool debug = false,
                            const char* debug_image_filename = nullptr) {
    std::vector<signed char> level_before;
    if (debug) level_before.assign(level, level + Npoints);
    const int saved_shift = ctx->cap_shift;
    int32_t nrefined = 0;
    bool ok = false;
    for (int attempt = 0; attempt < 4; ++attempt) {
        // layout of io_out: points | levels | npoints | nrefined
        const size_t o_lv = (size_t)Npoints * 16, o_np = o_lv + (((size_t)Npoints + 7) & ~(size_t)7), o_nr = o_np + 8;
        if (ensure(ctx, ctx->io_out, o_nr + 8)) break;
        char* base = (char*)ctx->io_out.p;
        // ONE block up and ONE block down (points | levels | npoints | nrefined through a host copy of the same layout):
        // every blocking copy of a few hundred bytes costs 15-20 us, and there were three each way
        std::vector<char>& blk = ctx->io_host_block;
        blk.assign(o_nr + 8, 0);
        memcpy(blk.data(), points_xy, (size_t)Npoints * 16);
        memcpy(blk.data() + o_lv, level, (size_t)Npoints);
        const int32_t np = Npoints;
        memcpy(blk.data() + o_np, &np, 4);
        if (hipMemcpy(base, blk.data(), o_nr + 8, hipMemcpyHostToDevice) != hipSuccess) break;
        if (mrgingham_amd_refine_batch(ctx, fr, image_pyramid_level, (double*)base, (signed char*)(base + o_lv),
                                       (const int32_t*)(base + o_np), Npoints, (int32_t*)(base + o_nr)))
            break;
        const int rc = mrgingham_amd_sync(ctx);
        if (rc == MRGINGHAM_AMD_ERR_CAPACITY && attempt < 3) {  // (the upload above restores the points)
            if (attempt == 2) ctx->cap_shift = 0;
            continue;
        }
        if (rc) break;
        if (hipMemcpy(blk.data(), base, o_nr + 8, hipMemcpyDeviceToHost) != hipSuccess) break;
        memcpy(&nrefined, blk.data() + o_nr, 4);
        memcpy(points_xy, blk.data(), (size_t)Npoints * 16);
        memcpy(level, blk.data() + o_lv, (size_t)Npoints);
        ok = true;
        break;
    }
    ctx->cap_shift = saved_shift;
    if (ok && debug) {  // the points refined by this pass, in index order (:390-392)
        std::vector<double> p;
        for (int i = 0; i < Npoints; ++i)
            if (level[i] != level_before[i]) { p.push_back(points_xy[2 * i]); p.push_back(points_xy[2 * i + 1]); }
        write_debug_dumps(ctx, fr, image_pyramid_level, true, debug_image_filename, p.data(), (int)(p.size() / 2));
    }
    return ok && nrefined > 0 ? nrefined : 0;
}

int refine_chessboard_corners_from_image_array_C(int Nrows, int Ncols, int stride, char* imagebuffer,
                                                 double* points_xy, signed char* level, int Npoints,
                                                 int image_pyramid_level, bool debug) {
    if (Nrows < 0 || Ncols < 0 || stride < Ncols || !imagebuffer || Npoints < 0) return 0;
    if (Npoints > 0 && (!points_xy || !level)) return 0;
    if (!check_level_and_layout(__func__, Nrows, Ncols, stride, image_pyramid_level)) return 0;
    if (Npoints == 0) return 0;
    CallerDevice caller_device_;
    mrgingham_amd_ctx* ctx = thread_ctx();
    if (!ctx) return 0;
    hipSetDevice(ctx->device);
    mrgingham_amd_frames fr;
    if (upload_frame(ctx, imagebuffer, Nrows, Ncols, stride, &fr)) return 0;
    return refine_on_device(ctx, &fr, points_xy, level, Npoints, image_pyramid_level, debug, nullptr);
}

/* C face of mrgingham::find_grid_from_points (mrgingham.hh:83-87; find_grid.cc:1216-1445): host only. */
bool mrgingham_amd_find_grid_from_points(const int* xy_scaled, int npoints, int gridn, double* xy_out) {
    if (!xy_scaled || !xy_out || npoints < 0 || gridn < 2) return false;
    std::vector<PointI> pts((size_t)npoints);
    for (int i = 0; i < npoints; ++i) pts[i] = PointI{xy_scaled[2 * i], xy_scaled[2 * i + 1]};
    std::vector<PointD> out;
    if (!find_grid_from_points(out, pts, gridn) || (int)out.size() != gridn * gridn) return false;
    memcpy(xy_out, out.data(), sizeof(double) * 2 * out.size());
    return true;
}

bool mrgingham_amd_find_grid_from_points_traced(const int* xy_scaled, int npoints, int gridn, double* xy_out,
                                                int debug, int debug_sequence_x, int debug_sequence_y) {
    mrg::g_grid_debug = debug != 0;
    mrg::g_grid_debug_sequence = {debug_sequence_x >= 0 && debug_sequence_y >= 0, debug_sequence_x, debug_sequence_y};
    const bool ok = mrgingham_amd_find_grid_from_points(xy_scaled, npoints, gridn, xy_out);
    mrg::g_grid_debug_sequence = {false, 0, 0};
    mrg::g_grid_debug = false;
    return ok;
}

/* Test hook: the same with the visiting order perturbed (grid.h, GridPerturbation). */
bool mrgingham_amd_find_grid_from_points_perturbed(const int* xy_scaled, int npoints, int gridn, double* xy_out,
                                                   unsigned ring_seed, int last_match) {
    g_grid_perturbation = GridPerturbation{ring_seed, last_match != 0};
    const bool ok = mrgingham_amd_find_grid_from_points(xy_scaled, npoints, gridn, xy_out);
    g_grid_perturbation = GridPerturbation{0u, false};
    return ok;
}

// mrgingham::find_chessboard_from_image_array (mrgingham.cc:38-140) on ONE frame that already lives
// on the device (dense, stride == width): detector and refinement on the GPU, grid finder on the
// host.  Returns the level the grid was found at, or -1.  `lv` receives the per-corner refinement
// level; without do_refine nothing is refined and every entry is the found level.
static int find_board_on_device(mrgingham_amd_ctx* ctx, const char* who, const mrgingham_amd_frames* fr, int gridn,
                                int image_pyramid_level, bool do_refine, std::vector<PointD>& board,
                                std::vector<signed char>& lv, bool debug = false,
                                const char* debug_image_filename = nullptr) {
    const int Nrows = fr->height, Ncols = fr->width;
    const int N = gridn * gridn;
    if (!debug && ctx->fb_pipeline && image_pyramid_level <= kMaxLevel) {
        // one frame through the pipelined batch detector (find_boards_submit / _collect below): the candidates of levels
        // 3, 2 and 1 in ONE device pass instead of a round trip per level, the refinement of every level in one more --
        // same boards (the pipelined detector equals the level-by-level schedule frame for frame, tests/test_gpu_board.py)
        board.assign((size_t)N, PointD{0., 0.});
        lv.assign((size_t)N, 0);
        signed char found_level = -1;
        const int ticket = fb_submit(ctx, fr, gridn, image_pyramid_level, &board[0].x, &found_level, 1, do_refine, lv.data());
        if (ticket < 0 || mrgingham_amd_find_boards_collect(ctx, ticket) != 0) return -1;
        return found_level;
    }
    std::vector<int32_t> xy;
    bool found = false;
    // image_pyramid_level >= 0: that level only; < 0: 3, 2, 1, 0 until a grid is found (mrgingham.cc:116-139)
    const int first = image_pyramid_level >= 0 ? image_pyramid_level : 3;
    const int last = image_pyramid_level >= 0 ? image_pyramid_level : 0;
    int level = first;
    for (; level >= last && !found; --level) {
        if (!check_level_and_layout(who, Nrows, Ncols, fr->stride, level)) continue;
        int32_t count = 0;
        const bool ok = detect_one_frame_all(ctx, fr, level, xy, &count, debug, debug_image_filename);
        if (!ok || count < N) continue;
        std::vector<PointI> cand((size_t)count);
        for (int i = 0; i < count; ++i) cand[i] = PointI{xy[2 * i], xy[2 * i + 1]};
        board.clear();
        mrg::g_grid_debug = debug;  // the reference hands its debug flag to the grid finder as well (mrgingham.cc:50-52)
        found = find_grid_from_points(board, cand, gridn) && (int)board.size() == N;  // mrgingham.cc:51
        mrg::g_grid_debug = false;
        if (found) break;
    }
    if (!found) return -1;
    lv.assign((size_t)N, (signed char)level);
    // refine towards level 0 while something still refines (mrgingham.cc:81-99)
    if (do_refine)
        for (int l = level - 1; l >= 0; --l)
            if (refine_on_device(ctx, fr, &board[0].x, lv.data(), N, l, debug, debug_image_filename) <= 0) break;
    return level;
}

/* Replaces find_chessboard_from_image_array_C (mrgingham_pywrap_cplusplus_bridge.h:25-42, .cc:72-138),
 * i.e. mrgingham::find_chessboard_from_image_array with refinement on (mrgingham.cc:38-140): detector
 * and refinement on the GPU, grid finder on the host. */
bool find_chessboard_from_image_array_C(int Nrows, int Ncols, int stride, char* imagebuffer, const int gridn,
                                        int image_pyramid_level, bool doblobs, bool debug, int debug_sequence_x,
                                        int debug_sequence_y,
                                        bool (*add_points)(double* xy, int N, void* cookie), void* cookie) {
    // bridge.cc:97-104: both coordinates >= 0 switch the grid finder's sequence trace on (stderr)
    struct TraceScope {
        TraceScope(int x, int y) { mrg::g_grid_debug_sequence = {x >= 0 && y >= 0, x, y}; }
        ~TraceScope() { mrg::g_grid_debug_sequence = {false, 0, 0}; }
    } trace_scope(debug_sequence_x, debug_sequence_y);
    if (Nrows < 0 || Ncols < 0 || stride < Ncols || !imagebuffer || !add_points || gridn < 2) return false;
    if (doblobs) {  // bridge.cc:104-113: find_circle_grid_from_image_array = blobs + grid finder, no refinement
        if (image_pyramid_level != 0) return false;
        CallerDevice caller_device_;
        mrgingham_amd_ctx* bctx = thread_ctx();
        if (!bctx) return false;
        hipSetDevice(bctx->device);
        mrgingham_amd_frames bfr;
        std::vector<int32_t> bxy;
        if (upload_frame(bctx, imagebuffer, Nrows, Ncols, stride, &bfr) ||
            !blobs_on_device(bctx, &bfr, (const uint8_t*)imagebuffer, stride, bxy))
            return false;
        std::vector<PointI> cand(bxy.size() / 2);
        for (size_t i = 0; i < cand.size(); ++i) cand[i] = PointI{bxy[2 * i], bxy[2 * i + 1]};
        std::vector<PointD> grid;
        if (!find_grid_from_points(grid, cand, gridn) || (int)grid.size() != gridn * gridn) return false;
        return (*add_points)(&grid[0].x, gridn * gridn, cookie);
    }
    if (image_pyramid_level > 10) {
        fprintf(stderr, "mrgingham_amd: %s(): Got an unreasonable image_pyramid_level = %d. Sorry.\n", __func__,
                image_pyramid_level);
        return false;
    }
    CallerDevice caller_device_;
    mrgingham_amd_ctx* ctx = thread_ctx();
    if (!ctx) return false;
    hipSetDevice(ctx->device);
    mrgingham_amd_frames fr;
    if (upload_frame(ctx, imagebuffer, Nrows, Ncols, stride, &fr)) return false;
    std::vector<PointD> board;
    std::vector<signed char> lv;
    if (find_board_on_device(ctx, __func__, &fr, gridn, image_pyramid_level, true, board, lv, debug, nullptr) < 0)
        return false;
    static_assert(sizeof(PointD) == 2 * sizeof(double), "add_points() takes interleaved doubles");
    return (*add_points)(&board[0].x, gridn * gridn, cookie);  // bridge.cc:133-137
}

/* The reference's file entry points: find_chessboard_corners_from_image_file
 * (find_chessboard_corners.cc:623-648) and find_chessboard_from_image_file (mrgingham.cc:145-170) are
 * cv::imread(GRAYSCALE) followed by the array functions.  Here the file is decoded by csrc/image_io
 * (binary PGM, non-interlaced PNG; 16-bit samples are reduced to their high byte, as cv::imread without
 * IMREAD_ANYDEPTH does) -- same results as the array
 * functions on the decoded pixels, same "Couldn't open image" failure. */
static bool load_gray8(const char* who, const char* filename, mrg::Image& im, std::vector<uint8_t>& tmp,
                       const uint8_t** px) {
    if (!filename || !mrg::read_image(filename, im)) {
        fprintf(stderr, "mrgingham_amd: %s(): Couldn't open image '%s'. Sorry.\n", who, filename ? filename : "(null)");
        return false;
    }
    if (im.depth == 16) {
        mrg::to_8bit_imread(im, tmp);  // cv::imread(GRAYSCALE) keeps the high byte; the CLI's own path rescales
        *px = tmp.data();
    } else {
        *px = im.px8.data();
    }
    return true;
}

int mrgingham_amd_read_image(const char* filename, int cli_scaling, uint8_t* out, size_t out_capacity, int* width,
                             int* height, int* depth) {
    mrg::Image im;
    if (!filename || !mrg::read_image(filename, im)) return -1;
    if (width) *width = im.w;
    if (height) *height = im.h;
    if (depth) *depth = im.depth;
    const size_t n = (size_t)im.w * im.h;
    if (!out) return 0;
    if (out_capacity < n) return -2;
    if (im.depth == 16) {
        std::vector<uint8_t> tmp;
        if (cli_scaling) mrg::to_8bit(im, tmp);
        else mrg::to_8bit_imread(im, tmp);
        memcpy(out, tmp.data(), n);
    } else {
        memcpy(out, im.px8.data(), n);
    }
    return 0;
}

bool find_chessboard_corners_from_image_file_C(const char* filename, int image_pyramid_level, bool debug,
                                               bool (*add_points)(int* xy, int N, double scale, void* cookie),
                                               void* cookie) {
    mrg::Image im;
    std::vector<uint8_t> tmp;
    const uint8_t* px = nullptr;
    if (!load_gray8(__func__, filename, im, tmp, &px)) return false;
    return find_chessboard_corners_from_image_array_C(im.h, im.w, im.w, (char*)px, image_pyramid_level, false, debug,
                                                      add_points, cookie);
}

bool find_chessboard_from_image_file_C(const char* filename, const int gridn, int image_pyramid_level, bool debug,
                                       bool (*add_points)(double* xy, int N, void* cookie), void* cookie) {
    mrg::Image im;
    std::vector<uint8_t> tmp;
    const uint8_t* px = nullptr;
    if (!load_gray8(__func__, filename, im, tmp, &px)) return false;
    return find_chessboard_from_image_array_C(im.h, im.w, im.w, (char*)px, gridn, image_pyramid_level, false, debug, -1,
                                              -1, add_points, cookie);
}

/* The preprocessing alone, host image in, host image out (dense width x height bytes): what the
 * Python recipe in find_board.docstring:8-10 does with cv2 before calling find_board.  Returns 0, or
 * -2 on an argument / device error. */
int mrgingham_amd_preprocess_image(const uint8_t* image, int width, int height, int stride, int do_clahe,
                                   int blur_radius, uint8_t* out) {
    if (!image || !out || width <= 0 || height <= 0 || stride < width || blur_radius < 0) return -2;
    CallerDevice caller_device_;
    mrgingham_amd_ctx* ctx = thread_ctx();
    if (!ctx) return -2;
    hipSetDevice(ctx->device);
    mrgingham_amd_frames fr;
    if (upload_frame(ctx, image, height, width, stride, &fr)) return -2;
    if (ensure(ctx, ctx->pre_out, (size_t)width * height + 64)) return -2;
    if (mrgingham_amd_preprocess_batch(ctx, &fr, do_clahe, blur_radius, (uint8_t*)ctx->pre_out.p, ctx->pix)) return -2;
    if (hipMemcpyAsync(out, ctx->pre_out.p, (size_t)width * height, hipMemcpyDeviceToHost, ctx->pix) != hipSuccess ||
        hipStreamSynchronize(ctx->pix) != hipSuccess)
        return -2;
    return 0;
}

/* What one worker of the reference CLI does with one decoded 8-bit image
 * (mrgingham-from-image.cc:71-111 and :160-171): [normalize + CLAHE] -> box blur ->
 * find_chessboard_from_image_array.  The frame is uploaded once; preprocessing, detector and
 * refinement run on the device, the grid finder on the host.  Returns the level the board was found
 * at (>= 0), -1 when no board was found, -2 on an argument / device error. */
int mrgingham_amd_preprocess_image16(const uint16_t* image, int width, int height, int stride, int do_clahe,
                                     int blur_radius, uint8_t* out) {
    if (!image || !out || width <= 0 || height <= 0 || stride < width || blur_radius < 0 || width > 32767 || height > 32767)
        return -2;
    CallerDevice caller_device_;
    mrgingham_amd_ctx* ctx = thread_ctx();
    if (!ctx) return -2;
    hipSetDevice(ctx->device);
    const size_t npx = (size_t)width * height;
    if (ensure(ctx, ctx->io_frame16, npx * 2 + 64) || ensure(ctx, ctx->pre_tmp, npx + 64) || ensure(ctx, ctx->pre_out, npx + 64) ||
        ensure(ctx, ctx->pre16_scratch, preprocess16_scratch_bytes(1, width, height)))
        return -2;
    if (copy_rows_async(ctx->io_frame16.p, (size_t)width * 2, image, (size_t)stride * 2, (size_t)width * 2, height,
                         hipMemcpyHostToDevice, ctx->pix) != hipSuccess)
        return -2;
    uint8_t* eight = (uint8_t*)(blur_radius > 0 ? ctx->pre_tmp.p : ctx->pre_out.p);
    if (!launch_preprocess16((const uint16_t*)ctx->io_frame16.p, (long long)npx, 1, width, height, width, do_clahe != 0,
                             8.0, eight, ctx->pre16_scratch.p, ctx->pix))
        return -2;
    if (blur_radius > 0) {
        const mrgingham_amd_frames fr{eight, (int64_t)npx, 1, width, height, width};
        if (mrgingham_amd_box_blur_batch(ctx, &fr, blur_radius, (uint8_t*)ctx->pre_out.p, ctx->pix)) return -2;
    }
    if (hipMemcpyAsync(out, ctx->pre_out.p, npx, hipMemcpyDeviceToHost, ctx->pix) != hipSuccess ||
        hipStreamSynchronize(ctx->pix) != hipSuccess)
        return -2;
    return 0;
}

int mrgingham_amd_process_image_ex(const void* image, int bits, int width, int height, int stride,
                                   const mrgingham_amd_cli_options* o, double* xy_out, signed char* levels_out) {
    if (!image || !o || (bits != 8 && bits != 16) || width <= 0 || height <= 0 || stride < width || o->gridn < 2 ||
        !xy_out || o->blur_radius < 0 || width > 32767 || height > 32767)
        return -2;
    if (o->image_pyramid_level > 10) {
        fprintf(stderr, "mrgingham_amd: %s(): Got an unreasonable image_pyramid_level = %d. Sorry.\n", __func__,
                o->image_pyramid_level);
        return -2;
    }
    struct TraceScope {  // --debug-sequence X,Y of the command-line tool (mrgingham-from-image.cc:262-276)
        TraceScope(int x, int y) { mrg::g_grid_debug_sequence = {x >= 0 && y >= 0, x, y}; }
        ~TraceScope() { mrg::g_grid_debug_sequence = {false, 0, 0}; }
    } trace_scope(o->debug_sequence_x, o->debug_sequence_y);
    CallerDevice caller_device_;
    mrgingham_amd_ctx* ctx = thread_ctx();
    if (!ctx) return -2;
    hipSetDevice(ctx->device);
    const size_t npx = (size_t)width * height;
    mrgingham_amd_frames fr;
    if (bits == 8) {
        if (upload_frame(ctx, image, height, width, stride, &fr)) return -2;
        if (o->do_clahe || o->blur_radius > 0) {
            if (ensure(ctx, ctx->pre_out, npx + 64)) return -2;
            if (mrgingham_amd_preprocess_batch(ctx, &fr, o->do_clahe, o->blur_radius, (uint8_t*)ctx->pre_out.p, ctx->pix))
                return -2;
            fr.frames = (const uint8_t*)ctx->pre_out.p;  // same stream as the detector's pixel kernels
        }
    } else {
        // mrgingham-from-image.cc:85-92: [normalize to 0..65535 + CLAHE on 16 bits] -> convertTo(CV_8U, 255/65535)
        if (ensure(ctx, ctx->io_frame16, npx * 2 + 64) || ensure(ctx, ctx->pre_tmp, npx + 64) ||
            ensure(ctx, ctx->pre_out, npx + 64) ||
            ensure(ctx, ctx->pre16_scratch, preprocess16_scratch_bytes(1, width, height)))
            return -2;
        if (copy_rows_async(ctx->io_frame16.p, (size_t)width * 2, image, (size_t)stride * 2, (size_t)width * 2, height,
                             hipMemcpyHostToDevice, ctx->pix) != hipSuccess)
            return -2;
        uint8_t* eight = (uint8_t*)(o->blur_radius > 0 ? ctx->pre_tmp.p : ctx->pre_out.p);
        if (!launch_preprocess16((const uint16_t*)ctx->io_frame16.p, (long long)npx, 1, width, height, width,
                                 o->do_clahe != 0, 8.0, eight, ctx->pre16_scratch.p, ctx->pix))
            return -2;
        fr = mrgingham_amd_frames{eight, (int64_t)npx, 1, width, height, width};
        if (o->blur_radius > 0) {
            if (mrgingham_amd_box_blur_batch(ctx, &fr, o->blur_radius, (uint8_t*)ctx->pre_out.p, ctx->pix)) return -2;
            fr.frames = (const uint8_t*)ctx->pre_out.p;
        }
    }
    if (o->debug) {  // mrgingham-from-image.cc:113-148: /tmp/<basename without extension>_preprocessed.png
        const char* fn = o->filename ? o->filename : "image";
        const char* slash = strrchr(fn, '/');
        std::string base = slash ? slash + 1 : fn;
        const size_t dot = base.rfind('.');
        if (dot != std::string::npos) base.resize(dot);
        const std::string outname = "/tmp/" + base + "_preprocessed.png";
        std::vector<uint8_t> host(npx);
        if (copy_rows_async(host.data(), width, fr.frames, fr.stride, width, height, hipMemcpyDeviceToHost, ctx->pix) ==
                hipSuccess &&
            hipStreamSynchronize(ctx->pix) == hipSuccess && write_png_gray8(outname.c_str(), host.data(), width, height))
            fprintf(stderr, "Wrote preprocessed image to %s\n", outname.c_str());
    }
    std::vector<PointD> board;
    std::vector<signed char> lv;
    if (o->do_blobs) {
        // mrgingham-from-image.cc:153-160: find_circle_grid_from_image_array on the preprocessed image, "level" 0
        std::vector<uint8_t> host(npx);
        std::vector<int32_t> bxy;
        if (copy_rows_async(host.data(), width, fr.frames, fr.stride, width, height, hipMemcpyDeviceToHost, ctx->pix) !=
                hipSuccess ||
            hipStreamSynchronize(ctx->pix) != hipSuccess || !blobs_on_device(ctx, &fr, host.data(), width, bxy))
            return -2;
        std::vector<PointI> cand(bxy.size() / 2);
        for (size_t i = 0; i < cand.size(); ++i) cand[i] = PointI{bxy[2 * i], bxy[2 * i + 1]};
        if (!find_grid_from_points(board, cand, o->gridn) || (int)board.size() != o->gridn * o->gridn) return -1;
        memcpy(xy_out, &board[0].x, sizeof(double) * 2 * (size_t)o->gridn * o->gridn);
        if (levels_out) memset(levels_out, 0, (size_t)o->gridn * o->gridn);
        return 0;
    }
    const int level = find_board_on_device(ctx, __func__, &fr, o->gridn, o->image_pyramid_level, o->do_refine != 0,
                                           board, lv, o->debug != 0, o->filename);
    if (level < 0) return -1;
    memcpy(xy_out, &board[0].x, sizeof(double) * 2 * (size_t)o->gridn * o->gridn);
    if (levels_out) memcpy(levels_out, lv.data(), (size_t)o->gridn * o->gridn);
    return level;
}

int mrgingham_amd_process_image(const uint8_t* image, int width, int height, int stride, int do_clahe,
                                int blur_radius, int gridn, int image_pyramid_level, int do_refine, double* xy_out,
                                signed char* levels_out) {
    mrgingham_amd_cli_options o{};
    o.do_clahe = do_clahe;
    o.blur_radius = blur_radius;
    o.gridn = gridn;
    o.image_pyramid_level = image_pyramid_level;
    o.do_refine = do_refine;
    return mrgingham_amd_process_image_ex(image, 8, width, height, stride, &o, xy_out, levels_out);
}

// grid-finder threads of the find_boards calls: <= 0 = one per core the process may use, at most 32 (the grid finder
// takes ~0.1 ms per frame and level: a few dozen threads cover a batch)
static int fb_threads(int nthreads) {
    if (nthreads <= 0) {
        nthreads = (int)std::thread::hardware_concurrency();
        if (nthreads > 32) nthreads = 32;
    }
    return nthreads > 0 ? nthreads : 1;
}

// The level search of mrgingham_amd_find_boards_batch, SYNCHRONOUS form: per level from `first` down to `last` one
// batched device pass over the frames still open (`open0`, ascending; the others must have h_found_level >= 0
// already), the grid finder on host threads, the boards found at the level refined densely level by level.  The
// pipelined form (find_boards_submit / _collect below) uses it for what its first pass leaves open, and option
// "find_boards_pipeline" 0 for everything.
// `h_levels` (may be NULL): per frame the gridn^2 refinement levels of its corners (what the reference's
// refinement_level array holds, mrgingham.cc:81-99); `do_refine` false: the boards stay as the grid finder made them.
static int find_boards_sync_levels(mrgingham_amd_ctx* ctx, const mrgingham_amd_frames* fr, int gridn, int first, int last,
                                   double* h_boards, signed char* h_found_level, int nthreads, std::vector<int> open,
                                   bool do_refine = true, signed char* h_levels = nullptr) {
    int rc = 0;
    const int B = fr->nframes, N = gridn * gridn;
    const int cap = 4 * N + 64;  // candidates kept per frame for the grid finder
    nthreads = fb_threads(nthreads);

    DevBuf &d_xy = ctx->fb_xy, &d_cnt = ctx->fb_cnt, &d_pts = ctx->fb_pts, &d_lv = ctx->fb_lv, &d_np = ctx->fb_np;
    if ((rc = ensure(ctx, d_xy, (size_t)B * cap * 8)) || (rc = ensure(ctx, d_cnt, (size_t)B * 4)) ||
        (rc = ensure(ctx, d_pts, (size_t)B * N * 16)) || (rc = ensure(ctx, d_lv, (size_t)B * N)) ||
        (rc = ensure(ctx, d_np, (size_t)B * 4)))
        return rc;
    std::vector<int32_t> h_xy((size_t)B * cap * 2), h_cnt(B), h_np(B, 0);
    std::vector<signed char> h_lv((size_t)B * N, 0);
    std::vector<double> h_pts((size_t)B * N * 2);

    // A dense copy of a few frames of the batch, so that a late level only runs on the frames that
    // still need it (one straggler must not cost the whole batch another two ChESS passes).
    const size_t frame_bytes = (size_t)fr->width * fr->height;
    auto gather = [&](DevBuf& buf, const std::vector<int>& idx, mrgingham_amd_frames* sub) -> int {
        int r = ensure(ctx, buf, frame_bytes * idx.size() + 64);
        if (r) return r;
        for (size_t k = 0; k < idx.size(); ++k)
            MRG_HIP_CHECK(copy_rows_async((char*)buf.p + k * frame_bytes, fr->width,
                                          fr->frames + (size_t)idx[k] * fr->frame_pitch, fr->stride, fr->width,
                                          fr->height, hipMemcpyDeviceToDevice, ctx->pix));
        *sub = mrgingham_amd_frames{(const uint8_t*)buf.p, (int64_t)frame_bytes, (int)idx.size(), fr->width, fr->height,
                                    fr->width};
        return 0;
    };

#ifdef MRG_EXPERIMENT
    static const bool dbg_t = getenv("MRG_DBG_FB") != nullptr;
#else
    constexpr bool dbg_t = false;
#endif
    auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t_prev = now();
    auto lap = [&](const char* what, int L, int n) { if (dbg_t) { const double t = now(); fprintf(stderr, "  [fb] L%d %-14s %3d frames %7.3f ms\n", L, what, n, t - t_prev); t_prev = t; } };

    std::vector<int> cur_idx(B);             // original index of every frame of the batch the detector runs on
    for (int f = 0; f < B; ++f) cur_idx[f] = f;
    mrgingham_amd_frames cur = *fr, rsub;

    for (int L = first; L >= last && !open.empty(); --L) {
        // (a) candidates at level L of the frames still open (compacted once at most half are left)
        if (open.size() * 2 <= cur_idx.size()) {
            if ((rc = gather(ctx->fb_frames, open, &cur))) break;
            cur_idx = open;
        }
        const int nb = (int)cur_idx.size();
        if ((rc = mrgingham_amd_detect_batch(ctx, &cur, L, (int32_t*)d_xy.p, cap, (int32_t*)d_cnt.p))) break;
        rc = mrgingham_amd_sync(ctx);
        if (rc == MRGINGHAM_AMD_ERR_CAPACITY) rc = 0;  // the frames concerned report count -1: handled below
        if (rc) break;
        if (hipMemcpy(h_cnt.data(), d_cnt.p, (size_t)nb * 4, hipMemcpyDeviceToHost) != hipSuccess ||
            hipMemcpy(h_xy.data(), d_xy.p, (size_t)nb * cap * 8, hipMemcpyDeviceToHost) != hipSuccess) {
            rc = fail(ctx, MRGINGHAM_AMD_ERR_DEVICE, "candidate download failed");
            break;
        }
        // Frames with more candidates than the batch buffer keeps (clutter), or whose component tables
        // overflowed (dense texture): the reference runs the grid finder on ALL candidates
        // (mrgingham.cc:50-51), so these are re-run one by one with exact capacity (and the
        // one-entry-per-pixel retry), on the calling thread's single-frame context.
        std::vector<std::vector<int32_t>> big(nb);
        for (int k = 0; k < nb && !rc; ++k) {
            if (h_found_level[cur_idx[k]] >= 0 || (h_cnt[k] >= 0 && h_cnt[k] <= cap)) continue;
            mrgingham_amd_ctx* one = same_device_ctx(ctx);
            if (!one) { rc = fail(ctx, MRGINGHAM_AMD_ERR_DEVICE, "no single-frame context"); break; }
            const mrgingham_amd_frames f1{cur.frames + (size_t)k * cur.frame_pitch, cur.frame_pitch, 1, cur.width,
                                          cur.height, cur.stride};
            int32_t n1 = 0;
            if (!detect_one_frame_all(one, &f1, L, big[k], &n1)) {
                rc = fail(ctx, MRGINGHAM_AMD_ERR_DEVICE, "frame %d, level %d: full-capacity detect failed", cur_idx[k], L);
                break;
            }
            h_cnt[k] = n1;
        }
        if (rc) break;
        lap("detect+D2H", L, nb);
        // (b) grid finder on host threads (mrgingham.cc:51), for the frames still without a board
        std::vector<char> found_now(nb, 0);
        std::atomic<int> next{0};
        auto worker = [&]() {
            for (int k; (k = next.fetch_add(1)) < nb;) {
                const int f = cur_idx[k];
                if (h_found_level[f] >= 0) continue;
                const int n = h_cnt[k];
                if (n < N) continue;
                const int32_t* src = big[k].empty() ? &h_xy[(size_t)k * cap * 2] : big[k].data();
                std::vector<PointI> cand((size_t)n);
                for (int i = 0; i < n; ++i) cand[i] = PointI{src[2 * i], src[2 * i + 1]};
                std::vector<PointD> board;
                if (find_grid_from_points(board, cand, gridn) && (int)board.size() == N) {
                    memcpy(h_boards + (size_t)f * N * 2, board.data(), sizeof(double) * 2 * N);
                    found_now[k] = 1;
                }
            }
        };
        ctx->pool.run(nthreads < nb ? nthreads : nb, worker);
        std::vector<int> found_pos;  // positions within the current batch
        for (int k = 0; k < nb; ++k)
            if (found_now[k]) {
                h_found_level[cur_idx[k]] = (signed char)L;
                found_pos.push_back(k);
                if (h_levels) memset(h_levels + (size_t)cur_idx[k] * N, L, (size_t)N);
            }
        lap("grid finder", L, (int)found_pos.size());
        if (found_pos.empty()) continue;
        {
            std::vector<int> still;
            for (int f : open)
                if (h_found_level[f] < 0) still.push_back(f);
            open.swap(still);
        }
        if (L == 0 || !do_refine) continue;
        // (c) refine the boards found at this level down to level 0 (mrgingham.cc:81-99): on the current
        // batch with zero points for the other frames, or on a dense copy of just those frames
        const mrgingham_amd_frames* rb = &cur;
        std::vector<int> ridx;  // position in the refine batch -> original frame
        if (found_pos.size() * 2 <= (size_t)nb) {
            for (int k : found_pos) ridx.push_back(cur_idx[k]);
            if ((rc = gather(ctx->fb_frames2, ridx, &rsub))) break;
            rb = &rsub;
        } else {
            ridx = cur_idx;
        }
        const int nr = (int)ridx.size();
        for (int k = 0; k < nr; ++k) {
            const int f = ridx[k];
            const bool is_new = h_found_level[f] == L;
            h_np[k] = is_new ? N : 0;
            if (is_new) {
                memset(h_lv.data() + (size_t)k * N, L, (size_t)N);
                memcpy(h_pts.data() + (size_t)k * N * 2, h_boards + (size_t)f * N * 2, sizeof(double) * 2 * N);
            }
        }
        if (hipMemcpy(d_pts.p, h_pts.data(), (size_t)nr * N * 16, hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(d_lv.p, h_lv.data(), (size_t)nr * N, hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(d_np.p, h_np.data(), (size_t)nr * 4, hipMemcpyHostToDevice) != hipSuccess) {
            rc = fail(ctx, MRGINGHAM_AMD_ERR_DEVICE, "board upload failed");
            break;
        }
        for (int l = L - 1; l >= 0 && !rc; --l)  // (refining past "nothing refined" is a no-op, mrgingham.cc:97-98)
            rc = mrgingham_amd_refine_batch(ctx, rb, l, (double*)d_pts.p, (signed char*)d_lv.p, (const int32_t*)d_np.p,
                                            N, nullptr);
        if (!rc) rc = mrgingham_amd_sync(ctx);
        if (rc == MRGINGHAM_AMD_ERR_CAPACITY) {
            // a frame of the refine batch overflowed the default tables at some level: refine the boards
            // found at this level one frame at a time (that path retries with one entry per pixel)
            rc = 0;
            mrgingham_amd_ctx* one = same_device_ctx(ctx);
            if (!one) { rc = fail(ctx, MRGINGHAM_AMD_ERR_DEVICE, "no single-frame context"); break; }
            for (int k = 0; k < nr; ++k) {
                if (!h_np[k]) continue;
                const mrgingham_amd_frames f1{rb->frames + (size_t)k * rb->frame_pitch, rb->frame_pitch, 1, rb->width,
                                              rb->height, rb->stride};
                double* bp = h_boards + (size_t)ridx[k] * N * 2;  // still the unrefined grid
                std::vector<signed char> lv1((size_t)N, (signed char)L);
                for (int l = L - 1; l >= 0; --l)
                    if (refine_on_device(one, &f1, bp, lv1.data(), N, l) <= 0) break;
                if (h_levels) memcpy(h_levels + (size_t)ridx[k] * N, lv1.data(), (size_t)N);
            }
            lap("refine 1-by-1", L, nr);
            continue;
        }
        if (rc) break;
        if (hipMemcpy(h_pts.data(), d_pts.p, (size_t)nr * N * 16, hipMemcpyDeviceToHost) != hipSuccess ||
            (h_levels && hipMemcpy(h_lv.data(), d_lv.p, (size_t)nr * N, hipMemcpyDeviceToHost) != hipSuccess)) {
            rc = fail(ctx, MRGINGHAM_AMD_ERR_DEVICE, "board download failed");
            break;
        }
        for (int k = 0; k < nr; ++k)
            if (h_np[k]) {
                memcpy(h_boards + (size_t)ridx[k] * N * 2, h_pts.data() + (size_t)k * N * 2, sizeof(double) * 2 * N);
                if (h_levels) memcpy(h_levels + (size_t)ridx[k] * N, h_lv.data() + (size_t)k * N, (size_t)N);
            }
        lap("refine+D2H", L, nr);
    }
    return rc;
}

/* ------------------------------------------------------------------------ */
/* The full detector over a batch, pipelined                                 */
/* ------------------------------------------------------------------------ */
// What mrgingham::find_chessboard_from_image_array does per frame (mrgingham.cc:106-140) is a chain of dependent
// steps that alternate between device and host: candidates at level 3 -> grid finder -> (none: level 2 -> grid
// finder ...) -> refinement of the found board level by level.  One batch at a time that leaves the device idle
// while the host threads run the grid finder and the host idle during the device passes (round 3: 3.8 ms per 64
// frames of 4096x3072 against 0.98 ms for the chain).  Here a batch is a JOB in three parts:
//   A  (device, queued by submit)  all level images in one pass over the frames, the responses + candidates of
//      levels 3, 2 AND 1 in one grid (levels 2 and 1 speculatively: together a third of a level-0 pass; 12 MP boards
//      are found at level 2, and the one frame in fifty that needs level 1 would otherwise hold up its whole batch),
//      candidates to pinned host memory;
//   H  (host, run inside the NEXT submit or by collect)  grid finder per frame, level 3 first, then 2, then 1
//      (mrgingham.cc:127-138) on the context's host threads -- started before that submit queues its own part A,
//      joined after it; the boards that were found go back to the device;
//   B  (device, queued by H on the job's component stream)  refinement of the found boards down to level 0
//      (mrgingham.cc:81-99) with the sparse schedule -- response only in the cells around the corners, frames it
//      cannot take repeated densely on the device (queue_sparse_levels) --, boards to pinned host memory.
// A job owns one scratch set from A to the end of B (B reads A's level images; its level sizes stay with that set, so
// jobs of different frame sizes can be in flight), so up to `scratch sets` jobs are in flight; submit completes the job
// that still holds the set it is about to take.  Frames without a board at levels 3-1 (no board in view, or one that
// only shows at full resolution) finish through the synchronous level search above on the single-frame context of the
// same device, which leaves the jobs in flight alone.  Results are the synchronous dense schedule's, double for double.

static int fb_complete(mrgingham_amd_ctx* ctx, mrgingham_amd_ctx::BoardsJob& job);
// Results of jobs that were completed before anybody collected them wait in done_tickets.  A caller that never collects
// (it may: the outputs are complete by then) must not make the list grow for ever: beyond 1024 entries the oldest go, and
// a _collect of such a ticket reports "no such ticket".
static void fb_remember(mrgingham_amd_ctx* ctx, int ticket, int status) {
    ctx->done_tickets.emplace_back(ticket, status);
    if (ctx->done_tickets.size() > 1024) ctx->done_tickets.erase(ctx->done_tickets.begin(), ctx->done_tickets.begin() + 512);
}
// phase clock of the find_boards calls (a dozen clock reads per batch; mrgingham_amd_find_boards_stats)
static double fb_now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define FB_LAP(i) do { const double t_ = fb_now(); ctx->fb_prof[i] += t_ - fb_t; fb_t = t_; } while (0)
#define FB_T0 double fb_t = fb_now()

static size_t fb_align(size_t v) { return (v + 255) & ~(size_t)255; }
struct FbPinned { int32_t *cnt, *xy; double* pts; signed char* lv; int32_t *np, *st; size_t bytes; };
static FbPinned fb_layout(void* base, int nlev, int B, int cap, int N) {
    FbPinned L;
    size_t o = 0;
    char* b = (char*)base;
    L.cnt = (int32_t*)(b + o); o += fb_align((size_t)nlev * B * 4);
    L.xy = (int32_t*)(b + o); o += fb_align((size_t)nlev * B * cap * 8);
    L.pts = (double*)(b + o); o += fb_align((size_t)B * N * 16);
    L.lv = (signed char*)(b + o); o += fb_align((size_t)B * N);
    L.np = (int32_t*)(b + o); o += fb_align((size_t)B * 4);
    L.st = (int32_t*)(b + o); o += fb_align((size_t)(kMaxLevel + 1) * B * 4);  // status words of the refinement, [level][frame]
    L.bytes = o;
    return L;
}

// part H, first half: waits for part A, deals with the frames whose candidate lists did not fit, and STARTS the grid
// finder (mrgingham.cc:51) on the context's host threads -- level by level per frame.  The caller may do something
// else before fb_host_end (submit queues the next batch's device passes there).
static void fb_grid_worker(mrgingham_amd_ctx::BoardsJob* job) {
    const int B = job->fr.nframes, N = job->gridn * job->gridn, cap = job->cap, nlev = job->nlev;
    const FbPinned pin = fb_layout(job->pin, nlev, B, cap, N);
    std::vector<PointI> cand;
    std::vector<PointD> board;
    const GridPhaseClock c0 = g_grid_clock;
    struct Leave {   // this thread's share of the batch's grid-finder time into the context's totals
        mrgingham_amd_ctx* ctx; GridPhaseClock c0;
        ~Leave() {
            if (!ctx) return;
            const GridPhaseClock& c = g_grid_clock;
            std::lock_guard<std::mutex> lk(ctx->fb_stat_mu);
            ctx->fb_grid.graph_t += c.graph_t - c0.graph_t; ctx->fb_grid.adjacency_t += c.adjacency_t - c0.adjacency_t;
            ctx->fb_grid.sequences_t += c.sequences_t - c0.sequences_t; ctx->fb_grid.cycles_t += c.cycles_t - c0.cycles_t;
            ctx->fb_grid.calls += c.calls - c0.calls; ctx->fb_grid.found += c.found - c0.found;
        }
    } leave{job->owner, c0};
    for (int k; (k = job->next.fetch_add(1)) < B;) {
        for (int li = 0; li < nlev; ++li) {
            const int n = pin.cnt[(size_t)li * B + k];
            if (n < N) continue;
            const std::vector<int32_t>& bg = job->big[(size_t)li * B + k];
            const int32_t* src = bg.empty() ? pin.xy + ((size_t)li * B + k) * cap * 2 : bg.data();
            cand.resize((size_t)n);
            for (int i = 0; i < n; ++i) cand[i] = PointI{src[2 * i], src[2 * i + 1]};
            board.clear();
            if (find_grid_from_points(board, cand, job->gridn) && (int)board.size() == N) {
                memcpy(job->h_boards + (size_t)k * N * 2, board.data(), sizeof(double) * 2 * N);
                job->h_found[k] = (signed char)job->levs[li];
                if (job->h_levels) memset(job->h_levels + (size_t)k * N, job->levs[li], (size_t)N);
                break;
            }
        }
    }
}
static int fb_host_begin(mrgingham_amd_ctx* ctx, mrgingham_amd_ctx::BoardsJob& job) {
    const int B = job.fr.nframes, N = job.gridn * job.gridn, cap = job.cap, nlev = job.nlev;
    const mrgingham_amd_frames* fr = &job.fr;
    job.state = 2;
    job.refine_queued = false;
    job.grid_running = false;
    MRG_HIP_CHECK(hipEventSynchronize(job.ev_a));
    {
        float ms = 0.f;
        if (job.ev_a0 && hipEventElapsedTime(&ms, job.ev_a0, job.ev_a) == hipSuccess) ctx->fb_dev_ms[0] += ms;
    }
    const FbPinned pin = fb_layout(job.pin, nlev, B, cap, N);
    int rc = 0;
    // Frames with more candidates than the batch buffer keeps (clutter), or whose component tables overflowed (dense
    // texture, count -1): the reference runs the grid finder on ALL candidates (mrgingham.cc:50-51), so these are
    // re-run one by one with exact capacity on the single-frame context of this device; the tables of the level grow
    // for the batches to come.
    job.big.assign((size_t)nlev * B, std::vector<int32_t>());
    bool overflowed = false;
    for (int li = 0; li < nlev && !rc; ++li)
        for (int k = 0; k < B && !rc; ++k) {
            int32_t& c = pin.cnt[(size_t)li * B + k];
            if (c >= 0 && c <= cap) continue;
            overflowed |= c < 0;
            mrgingham_amd_ctx* one = same_device_ctx(ctx);
            if (!one) { rc = fail(ctx, MRGINGHAM_AMD_ERR_DEVICE, "no single-frame context"); break; }
            const mrgingham_amd_frames f1{fr->frames + (size_t)k * fr->frame_pitch, fr->frame_pitch, 1, fr->width, fr->height,
                                          fr->stride};
            int32_t n1 = 0;
            if (!detect_one_frame_all(one, &f1, job.levs[li], job.big[(size_t)li * B + k], &n1))
                rc = fail(ctx, MRGINGHAM_AMD_ERR_DEVICE, "frame %d, level %d: full-capacity detect failed", k, job.levs[li]);
            c = n1;
        }
    if (rc) return rc;
    if (overflowed)
        for (int li = 0; li < nlev; ++li) {
            int grew = 0;
            harvest_status(ctx, job.set, job.levs[li], &grew, true);
        }
    const int nthreads = fb_threads(job.nthreads);
    job.next.store(0);
    job.nworkers = (nthreads < B ? nthreads : B) - 1;  // + the calling thread, in fb_host_end
    job.owner = ctx;
    ctx->fb_threads_used = job.nworkers + 1;
    mrgingham_amd_ctx::BoardsJob* jp = &job;
    if (job.nworkers > 0) {
        ctx->pool.start(job.nworkers, [jp] { fb_grid_worker(jp); });
        job.grid_running = true;
    }
    return 0;
}

// part H, second half: joins the grid finder and queues part B -- the boards found above level 0, refined level by level
// (mrgingham.cc:81-99) on the job's component stream.
static int fb_host_end(mrgingham_amd_ctx* ctx, mrgingham_amd_ctx::BoardsJob& job) {
    const int B = job.fr.nframes, N = job.gridn * job.gridn, cap = job.cap, nlev = job.nlev;
    const mrgingham_amd_frames* fr = &job.fr;
    FB_T0;
    fb_grid_worker(&job);
    if (job.grid_running) {
        ctx->pool.wait();
        job.grid_running = false;
    }
    FB_LAP(3);
    const FbPinned pin = fb_layout(job.pin, nlev, B, cap, N);
    int rc = 0;
    int top = 0, nref = 0;
    for (int k = 0; k < B && job.do_refine; ++k) {
        const int L = job.h_found[k];
        pin.np[k] = L >= 1 ? N : 0;
        if (L < 1) continue;
        ++nref;
        top = L > top ? L : top;
        memset(pin.lv + (size_t)k * N, L, (size_t)N);
        memcpy(pin.pts + (size_t)k * N * 2, job.h_boards + (size_t)k * N * 2, sizeof(double) * 2 * N);
    }
    if (nref > 0) {
        const int saved = ctx->cur;
        ctx->cur = job.set;  // (the helpers below address the current set)
        hipStream_t cc = cur_cc(ctx);
        const size_t pb = (size_t)B * N * 16, lb = (size_t)B * N;
        // boards | levels | point counts: one block on both sides
        char* const d_pts = (char*)job.d_pts.p;
        char* const d_lv = d_pts + fb_align(pb);
        char* const d_np = d_lv + fb_align(lb);
        char* const d_pts0 = (char*)job.d_pts0.p;
        const bool sparse = ctx->cc_lds && !ctx->use_v0 && top <= kRefineLevelsMax &&
                            (ctx->sparse_refine == 2 ||
                             (ctx->sparse_refine == 1 && (long long)fr->width * fr->height * B >= kSparsePaysPixels));
        hipEventRecord(job.ev_b0, cc);
        hipError_t e = hipMemcpyAsync(d_pts, pin.pts, fb_align(pb) + fb_align(lb) + (size_t)B * 4, hipMemcpyHostToDevice, cc);
        if (e == hipSuccess && sparse)  // (only the dense repeat of a sparse refinement goes back to them)
            e = hipMemcpyAsync(d_pts0, d_pts, fb_align(pb) + lb, hipMemcpyDeviceToDevice, cc);
        if (e == hipSuccess) {
            auto& ps = ctx->pts[job.set];
            RefineIO io{(double*)d_pts, (signed char*)d_lv, (const int32_t*)d_np, N, nullptr,
                        (int32_t*)ps.leader.p, (int32_t*)ps.need.p, (int32_t*)ps.nseeds.p, (uint32_t*)ps.seeds.p,
                        (int32_t*)ps.sroot.p};
            const SparseRestore src{nullptr, 0, 0, (const double*)d_pts0, (const signed char*)(d_pts0 + fb_align(pb))};
            rc = queue_sparse_levels(ctx, fr, top, io, src, !sparse);
            job.top = top;
            e = hipMemcpyAsync(pin.pts, d_pts, job.h_levels ? fb_align(pb) + lb : pb, hipMemcpyDeviceToHost, cc);
            for (int L = 0; L < top && e == hipSuccess; ++L)  // (a frame whose tables overflowed at a level was not refined there)
                e = hipMemcpyAsync(pin.st + (size_t)L * B, status_of(ctx, L), (size_t)B * 4, hipMemcpyDeviceToHost, cc);
            if (e == hipSuccess) e = hipEventRecord(job.ev_b, cc);
            end_op(ctx);
            job.refine_queued = true;
        }
        ctx->cur = saved;
        if (e != hipSuccess) return fail_hip(ctx, e, "find_boards refinement", __FILE__, __LINE__);
        if (rc) return rc;
    }
    FB_LAP(4);
    return 0;
}

// the rest of a job: wait for part B, boards into the caller's array, then the frames still open (level_arg < 0 only)
static int fb_finish(mrgingham_amd_ctx* ctx, mrgingham_amd_ctx::BoardsJob& job) {
    const int B = job.fr.nframes, N = job.gridn * job.gridn;
    int rc = 0;
    FB_T0;
    if (job.refine_queued) {
        MRG_HIP_CHECK(hipEventSynchronize(job.ev_b));
        FB_LAP(5);
        {
            float ms = 0.f;
            if (job.ev_b0 && hipEventElapsedTime(&ms, job.ev_b0, job.ev_b) == hipSuccess) ctx->fb_dev_ms[1] += ms;
        }
        const FbPinned pin = fb_layout(job.pin, job.nlev, B, job.cap, N);
        bool overflowed = false;
        for (int k = 0; k < B && !rc; ++k) {
            const int Lf = job.h_found[k];
            if (Lf < 1) continue;
            int bad = 0;
            for (int L = 0; L < Lf; ++L) bad |= pin.st[(size_t)L * B + k] & (kStatusHotOverflow | kStatusCandOverflow);
            if (!bad) {
                memcpy(job.h_boards + (size_t)k * N * 2, pin.pts + (size_t)k * N * 2, sizeof(double) * 2 * N);
                if (job.h_levels) memcpy(job.h_levels + (size_t)k * N, pin.lv + (size_t)k * N, (size_t)N);
                continue;
            }
            // The component tables of a level overflowed for this frame (dense texture): it was not refined there.  Its
            // board -- still the grid finder's in h_boards -- is refined on the single-frame context, which retries with
            // a table entry per pixel; the tables of this context grow for the batches to come.
            overflowed = true;
            mrgingham_amd_ctx* one = same_device_ctx(ctx);
            if (!one) { rc = fail(ctx, MRGINGHAM_AMD_ERR_DEVICE, "no single-frame context"); break; }
            const mrgingham_amd_frames f1{job.fr.frames + (size_t)k * job.fr.frame_pitch, job.fr.frame_pitch, 1, job.fr.width,
                                          job.fr.height, job.fr.stride};
            std::vector<signed char> lv1((size_t)N, (signed char)Lf);
            for (int l = Lf - 1; l >= 0; --l)
                if (refine_on_device(one, &f1, job.h_boards + (size_t)k * N * 2, lv1.data(), N, l) <= 0) break;
            if (job.h_levels) memcpy(job.h_levels + (size_t)k * N, lv1.data(), (size_t)N);
        }
        if (overflowed)
            for (int L = 0; L < job.top; ++L) {
                int grew = 0;
                harvest_status(ctx, job.set, L, &grew, true);
            }
        job.refine_queued = false;
        FB_LAP(6);
    }
    const int lowest = job.levs[job.nlev - 1];
    std::vector<int> open;
    if (job.level_arg < 0 && lowest > 0)
        for (int k = 0; k < B; ++k)
            if (job.h_found[k] < 0) open.push_back(k);
    job.state = 0;  // the set is this job's no longer
    if (!open.empty()) {
        // what is left (no board in view, or one that only shows at full resolution): level by level, synchronously, on
        // the single-frame context of this device -- its own streams and scratch, so the jobs in flight here stay so
        mrgingham_amd_ctx* one = same_device_ctx(ctx);
        if (!one) return fail(ctx, MRGINGHAM_AMD_ERR_DEVICE, "no single-frame context");
        rc = find_boards_sync_levels(one, &job.fr, job.gridn, lowest - 1, 0, job.h_boards, job.h_found, job.nthreads, open,
                                     job.do_refine, job.h_levels);
        if (rc) ctx->err = one->err;
    }
    return rc;
}

static int fb_abandon(mrgingham_amd_ctx* ctx, mrgingham_amd_ctx::BoardsJob& job, int rc) {
    if (job.grid_running) ctx->pool.wait();
    job.grid_running = false;
    hipStreamSynchronize(ctx->ccs[job.set]);  // nothing of this job may stay queued behind an error
    job.state = 0;
    job.refine_queued = false;
    return rc;
}
static int fb_complete(mrgingham_amd_ctx* ctx, mrgingham_amd_ctx::BoardsJob& job) {
    int rc = 0;
    if (job.state == 1) {
        rc = fb_host_begin(ctx, job);
        if (!rc) rc = fb_host_end(ctx, job);
    }
    if (rc) return fb_abandon(ctx, job, rc);
    return fb_finish(ctx, job);
}

}  // extern "C"
static void fb_drain(mrgingham_amd_ctx* ctx) {
    for (auto& j : ctx->jobs)
        if (j.state != 0) {
            const int ticket = j.ticket;
            fb_remember(ctx, ticket, fb_complete(ctx, j));
        }
}
extern "C" {

}  // extern "C"
// (the public entry + what the single-image wrappers need on top of it: no refinement, the corners' refinement levels)
static int fb_submit(mrgingham_amd_ctx* ctx, const mrgingham_amd_frames* fr, int gridn, int image_pyramid_level,
                     double* h_boards, signed char* h_found_level, int nthreads, bool do_refine, signed char* h_levels) {
    int rc = validate_frames(ctx, fr);
    if (rc) return rc;
    if (gridn < 2 || image_pyramid_level > kMaxLevel || !h_boards || !h_found_level)
        return fail(ctx, MRGINGHAM_AMD_ERR_ARG, "bad gridn / level / NULL outputs");
    const int B = fr->nframes, N = gridn * gridn;
    const int ticket = ctx->next_ticket++ & 0x3fffffff;
    FB_T0;
    for (int f = 0; f < B; ++f) h_found_level[f] = -1;
    if (B == 0) {
        fb_remember(ctx, ticket, 0);
        return ticket;
    }
    MRG_HIP_CHECK(hipSetDevice(ctx->device));
    if (!ctx->fb_pipeline) {  // option "find_boards_pipeline" 0: the synchronous dense schedule, at once
        for (auto& j : ctx->jobs)
            if (j.state != 0) fb_remember(ctx, j.ticket, fb_complete(ctx, j));
        const int first = image_pyramid_level >= 0 ? image_pyramid_level : 3;
        const int last = image_pyramid_level >= 0 ? image_pyramid_level : 0;
        std::vector<int> open(B);
        for (int f = 0; f < B; ++f) open[f] = f;
        fb_remember(ctx, ticket, find_boards_sync_levels(ctx, fr, gridn, first, last, h_boards, h_found_level, nthreads, open,
                                                               do_refine, h_levels));
        return ticket;
    }
    // levels searched in the first pass: the one asked for, or 3, 2 and 1 (levels 2 and 1 speculatively: together they
    // cost the device a third of a level-0 pass, 12 MP boards are found at level 2, and the one frame in fifty that
    // needs level 1 would otherwise hold up its whole batch); level 0 only for what is still open after them
    const int top = image_pyramid_level >= 0 ? image_pyramid_level : 3;
    const int nlev = image_pyramid_level >= 0 ? 1 : 3;
    const int cap = 4 * N + 64;  // candidates kept per frame for the grid finder
    // the refinement takes the sparse schedule where it pays: such a context keeps three scratch sets (choose_sets)
    if (ctx->sparse_refine && top >= 1 && ctx->cc_lds && !ctx->use_v0) ctx->sparse_seen = true;
    {   // a change of the rotation (another batch shape) synchronises and may free a set: no job may be in flight then
        const double per_set = 5.0 * (double)B * fr->width * fr->height;
        const double mx = per_set > ctx->max_set_bytes ? per_set : ctx->max_set_bytes;
        const int want = ctx->nsets_fixed ? ctx->nsets : (3.0 * mx <= (ctx->sparse_seen ? 16e9 : 8e9) ? 3 : 2);
        if (want != ctx->nsets)
            for (auto& j : ctx->jobs)
                if (j.state != 0) fb_remember(ctx, j.ticket, fb_complete(ctx, j));
    }
    if ((rc = choose_sets(ctx, fr))) return rc;
    // the set this job is going to take may still belong to an earlier one: that one is completed first
    {
        auto& occupant = ctx->jobs[(ctx->cur + 1) % ctx->nsets];
        if (occupant.state != 0) fb_remember(ctx, occupant.ticket, fb_complete(ctx, occupant));
    }
    // Level scratch of THAT set alone (the other sets belong to jobs in flight, possibly of another frame size: a stream
    // of mixed resolutions keeps every job's level sizes with its own set).  Buffers only ever grow; a buffer that has
    // to grow synchronises the device first, which the jobs in flight survive.
    {
        const int target = (ctx->cur + 1) % ctx->nsets;
        for (int L = 0; L <= top; ++L)
            if ((rc = ensure_level_set(ctx, target, L, B, fr->width, fr->height, N))) return rc;
    }
    if ((rc = ensure_points(ctx, B, N))) return rc;
    if (!ctx->sparse_stat.p) {
        if ((rc = ensure(ctx, ctx->sparse_stat, 256))) return rc;
        MRG_HIP_CHECK(hipMemset(ctx->sparse_stat.p, 0, 256));
    }
    {   // everything the job allocates BEFORE the scratch rotation moves (begin_op): an allocation that fails returns with
        // the context as it was -- no set taken, nothing queued -- and the call can simply be made again
        auto& nj = ctx->jobs[(ctx->cur + 1) % ctx->nsets];
        if ((rc = ensure(ctx, nj.d_cnt, fb_align((size_t)nlev * B * 4) + (size_t)nlev * B * cap * 8)) ||
            (rc = ensure(ctx, nj.d_pts, fb_align((size_t)B * N * 16) + fb_align((size_t)B * N) + (size_t)B * 4)) ||
            (rc = ensure(ctx, nj.d_pts0, fb_align((size_t)B * N * 16) + (size_t)B * N)))
            return rc;
        const size_t need = fb_layout(nullptr, nlev, B, cap, N).bytes;
        if (need > nj.pin_bytes) {
            if (nj.pin) hipHostFree(nj.pin);
            nj.pin = nullptr;
            nj.pin_bytes = 0;
            MRG_HIP_CHECK(hipHostMalloc(&nj.pin, need + need / 4, hipHostMallocDefault));
            nj.pin_bytes = need + need / 4;
        }
        if (!nj.ev_a) MRG_HIP_CHECK(hipEventCreate(&nj.ev_a));
        if (!nj.ev_b) MRG_HIP_CHECK(hipEventCreate(&nj.ev_b));
        if (!nj.ev_a0) MRG_HIP_CHECK(hipEventCreate(&nj.ev_a0));
        if (!nj.ev_b0) MRG_HIP_CHECK(hipEventCreate(&nj.ev_b0));
    }
    begin_op(ctx, top);
    auto& job = ctx->jobs[ctx->cur];
    job.set = ctx->cur;
    job.ticket = ticket;
    job.fr = *fr;
    job.gridn = gridn;
    job.level_arg = image_pyramid_level;
    job.nthreads = nthreads;
    job.nlev = nlev;
    for (int li = 0; li < 3; ++li) job.levs[li] = top - li;
    job.cap = cap;
    job.h_boards = h_boards;
    job.h_found = h_found_level;
    job.h_levels = h_levels;
    job.do_refine = do_refine;
    job.refine_queued = false;
    // The host part of the job before this one runs inside this call: its grid-finder threads are started first when
    // its candidates have already arrived (the steady state), so that they work while this thread queues the device
    // passes below; otherwise after them.
    mrgingham_amd_ctx::BoardsJob* prev = nullptr;
    for (auto& other : ctx->jobs)
        if (&other != &job && other.state == 1) prev = &other;
    bool prev_begun = false;
    FB_LAP(0);
    if (prev && hipEventQuery(prev->ev_a) == hipSuccess) {
        const int r = fb_host_begin(ctx, *prev);
        if (r) {
            fb_remember(ctx, prev->ticket, fb_abandon(ctx, *prev, r));
            prev = nullptr;
        }
        prev_begun = true;
    }
    FB_LAP(1);
    order_after_previous(ctx, {}, {});
    hipEventRecord(job.ev_a0, ctx->pix);
    // part A: level images of every level up to the top in one pass (the refinement's variance windows and cells read
    // them too), the responses of the levels searched, their candidates
    queue_level_images(ctx, fr, top, true);
    LevelBatch lbs[3];
    bool merged = false;
    if (job.nlev >= 2 && !ctx->use_v0 && ctx->multi_level) {
        LevelBatch mlb[3];
        CompTables mt[3];
        for (int k = 0; k < job.nlev; ++k) {  // largest level first
            mlb[k] = level_batch_of(ctx, fr, job.levs[job.nlev - 1 - k]);
            mt[k] = tables_of(ctx, job.levs[job.nlev - 1 - k]);
        }
        if (chess_multi_ok(mlb, job.nlev, B) && launch_chess_multi(mlb, mt, job.nlev, B, ctx->pix, ctx->chess_seg)) {
            merged = true;
            for (int k = 0; k < job.nlev; ++k) lbs[job.nlev - 1 - k] = mlb[k];
            hipEventRecord(ctx->ev_pix[top], ctx->pix);
            for (int li = 0; li < job.nlev; ++li)
                if (B > ctx->pending_frames[ctx->cur][job.levs[li]]) ctx->pending_frames[ctx->cur][job.levs[li]] = B;
        }
    }
    if (!merged)
        for (int li = 0; li < job.nlev; ++li) lbs[li] = queue_level_chess(ctx, fr, job.levs[li]);
    hipStream_t cc = cur_cc(ctx);
    hipError_t e = hipStreamWaitEvent(cc, ctx->ev_pix[merged ? top : job.levs[job.nlev - 1]], 0);
    {   // the candidates of every level searched in this pass: one grid per kernel, not one per level
        CompTables dts[3];
        DetectOut douts[3];
        for (int li = 0; li < job.nlev; ++li) {
            dts[li] = tables_of(ctx, job.levs[li]);
            douts[li] = DetectOut{(int32_t*)((char*)job.d_cnt.p + fb_align((size_t)job.nlev * B * 4)) + (size_t)li * B * cap * 2, cap,
                                  (int32_t*)job.d_cnt.p + (size_t)li * B};
        }
        launch_cc_detect_levels(lbs, dts, job.levs, douts, job.nlev, B, cc);
    }
    const FbPinned pin = fb_layout(job.pin, job.nlev, B, cap, N);
    if (e == hipSuccess)  // counts | candidates: one block on both sides
        e = hipMemcpyAsync(pin.cnt, job.d_cnt.p, fb_align((size_t)job.nlev * B * 4) + (size_t)job.nlev * B * cap * 8, hipMemcpyDeviceToHost, cc);
    if (e == hipSuccess) e = hipEventRecord(job.ev_a, cc);
    end_op(ctx);
    if (e == hipSuccess) e = hipGetLastError();
    if (e == hipSuccess) job.state = 1;
    FB_LAP(2);
    ++ctx->fb_prof_n;
    // ... and while the device works on that: the (rest of the) host part of the job before this one
    if (prev) {
        int r = prev_begun ? 0 : fb_host_begin(ctx, *prev);
        if (!r) r = fb_host_end(ctx, *prev);
        if (r) fb_remember(ctx, prev->ticket, fb_abandon(ctx, *prev, r));
    }
    if (e != hipSuccess) return fail_hip(ctx, e, "find_boards first pass", __FILE__, __LINE__);
    return ticket;
}

extern "C" {

int mrgingham_amd_find_boards_submit(mrgingham_amd_ctx* ctx, const mrgingham_amd_frames* fr, int gridn,
                                     int image_pyramid_level, double* h_boards, signed char* h_found_level, int nthreads) {
    return fb_submit(ctx, fr, gridn, image_pyramid_level, h_boards, h_found_level, nthreads, true, nullptr);
}

int mrgingham_amd_find_boards_collect(mrgingham_amd_ctx* ctx, int ticket) {
    if (!ctx) return MRGINGHAM_AMD_ERR_ARG;
    MRG_HIP_CHECK(hipSetDevice(ctx->device));
    for (auto& j : ctx->jobs)
        if (j.state != 0 && j.ticket == ticket) return fb_complete(ctx, j);
    for (size_t i = 0; i < ctx->done_tickets.size(); ++i)
        if (ctx->done_tickets[i].first == ticket) {
            const int rc = ctx->done_tickets[i].second;
            ctx->done_tickets.erase(ctx->done_tickets.begin() + (long)i);
            return rc;
        }
    return fail(ctx, MRGINGHAM_AMD_ERR_ARG, "find_boards_collect: no such ticket (%d)", ticket);
}

int mrgingham_amd_find_boards_batch(mrgingham_amd_ctx* ctx, const mrgingham_amd_frames* fr, int gridn,
                                    int image_pyramid_level, double* h_boards, signed char* h_found_level,
                                    int nthreads) {
    const int ticket = mrgingham_amd_find_boards_submit(ctx, fr, gridn, image_pyramid_level, h_boards, h_found_level, nthreads);
    if (ticket < 0) return ticket;
    return mrgingham_amd_find_boards_collect(ctx, ticket);
}

}  // extern "C"
