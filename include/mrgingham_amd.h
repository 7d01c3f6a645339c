/*
 * mrgingham_amd.h -- C-ABI of the MI355X-native chessboard-corner candidate path.
 *
 * One shared library (mrgingham_amd/libmrgingham_amd.so, HIP for gfx950) exports
 *
 *   (1) the reference's own C symbols for this path, same names, arguments and
 *       error behaviour, so the reference's callers bind to it unchanged;
 *   (2) a batch API over frames that already live in HBM, which is what the
 *       single-frame symbols are thin wrappers of, and what bench.py measures.
 *
 * Plain pointers and sizes only: no torch, OpenCV or C++ types cross this
 * boundary.  INTEGRATION.md shows the reference-side bindings.  "file:line"
 * citations are into the upstream dkogan/mrgingham tree.
 *
 * There is no CPU fallback anywhere behind this header: with no usable HIP
 * device every entry point fails loudly (message on stderr, false / 0 / NULL /
 * negative status), it never computes on the host.
 */
#pragma once
#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* History of the boundary (mrgingham_amd_abi_version() reports what the LIBRARY was built as; compare with this macro):
 *   1  rounds 1-3.
 *   2  round 4: MRGINGHAM_AMD_ERR_SPARSE (-4) left the status enum -- value -4 stays RESERVED, it is never returned and
 *      will not be reused --; option "sparse_refine" defaults to 1 (a context that runs sparse chains keeps a third scratch
 *      set, up to 16 GB: INTEGRATION.md 5e); find_boards_submit / _collect, chain_multi, host_alloc / _register,
 *      set_wait_policy, set_thread_device and the file entry points were added.
 *   3  round 5: mrgingham_amd_find_boards_stats, _grid_clock, _packed_layout, _gather_rccl added; the packed corner block
 *      is a multiple of 8 bytes; the reference-symbol wrappers restore the caller's current HIP device.
 *   4  round 6: mrgingham_amd_sclk_mhz added; options "chess_seg" / "chess16_seg" are per context (they were process-wide) and
 *      mean balanced segments, option "preprocess_fused"; _gather_rccl, _chain_multi, _sync_multi, _stream_wait_multi restore
 *      the caller's current HIP device; _gather_rccl needs no RCCL header or library at build time and never loads a second RCCL. */
#define MRGINGHAM_AMD_ABI_VERSION 4

/* ------------------------------------------------------------------------ */
/* (1) Reference symbols                                                    */
/* ------------------------------------------------------------------------ */

/* Replaces mrgingham_ChESS_response_5 (ChESS.h:31-34, ChESS.c:55-106).
 * HOST pointers, caller-owned.  `image` is w x h bytes with `stride` bytes per
 * row, `response` is a dense w x h int16 image.  Exactly like the reference,
 * only the interior [7,w-7) x [7,h-7) is written; the 7-pixel frame of
 * `response` is left untouched.  No return value; a device failure prints a
 * message on stderr and leaves `response` unwritten. */
void mrgingham_ChESS_response_5(int16_t* response, const uint8_t* image, int w, int h, int stride);

/* Replaces find_chessboard_corners_from_image_array_C
 * (mrgingham_pywrap_cplusplus_bridge.h:10-23, .cc:28-70), the extern "C" face
 * of mrgingham::find_chessboard_corners_from_image_array
 * (find_chessboard_corners.hh:12-30, .cc:568-587).
 * HOST image buffer (Nrows x Ncols bytes, `stride` bytes per row).  On success
 * calls add_points(xy, N, 1/1000., cookie) once with N > 0 interleaved
 * (x,y)*1000 ints in the reference's order (valid only during the callback)
 * and returns its result.  Returns false without calling add_points when
 * nothing was found or on an error (message on stderr):
 *   - image_pyramid_level outside [0,10]            (find_chessboard_corners.cc:433-441)
 *   - level 0 and stride != Ncols (non-continuous)  (find_chessboard_corners.cc:461-466)
 * doblobs (bridge.cc:50-55): the blob detector, find_blobs_from_image_array (find_blobs.cc:14-46: a
 * cv::SimpleBlobDetector with minArea 20, maxArea 80000, minDistBetweenBlobs 5, dark blobs), at level 0
 * only (any other level: false); like the reference it always "finds", i.e. add_points is called even with
 * N = 0.  debug: the reference's /tmp/mrgingham-* dumps of the pass
 * (find_chessboard_corners.cc:282-315, :453-459, :513-541). */
bool find_chessboard_corners_from_image_array_C(int Nrows, int Ncols, int stride, char* imagebuffer,
                                                int image_pyramid_level, bool doblobs, bool debug,
                                                bool (*add_points)(int* xy, int N, double scale, void* cookie),
                                                void* cookie);

/* C face of mrgingham::refine_chessboard_corners_from_image_array
 * (find_chessboard_corners.hh:51-72, .cc:591-619), which the reference only
 * exposes with std::vector / cv::Mat arguments.  points_xy: Npoints interleaved
 * doubles (full-resolution pixel coordinates), updated in place; level[i] is
 * the pyramid level point i was computed at and is lowered to
 * image_pyramid_level for each point that gets refined.  Returns the number of
 * refined points (0 on error, like the reference). */
int refine_chessboard_corners_from_image_array_C(int Nrows, int Ncols, int stride, char* imagebuffer,
                                                 double* points_xy, signed char* level, int Npoints,
                                                 int image_pyramid_level, bool debug);

/* Replaces find_chessboard_from_image_array_C (mrgingham_pywrap_cplusplus_bridge.h:25-42,
 * .cc:72-138) = mrgingham::find_chessboard_from_image_array with refinement on
 * (mrgingham.hh:28-56, mrgingham.cc:38-140): image_pyramid_level >= 0 uses that level,
 * < 0 tries levels 3, 2, 1, 0 until the grid finder succeeds; the gridn x gridn corners are then
 * refined towards level 0 and handed to add_points(xy, gridn*gridn, cookie) in board order.
 * The detector and the refinement run on the GPU, the grid finder (find_grid.cc) on the host.
 * Returns false when no board is found or on an error; doblobs (level 0 only, bridge.cc:104-113) is
 * find_circle_grid_from_image_array: blob detector + grid finder, no refinement;
 * debug writes the detector's / refinement's dumps (see above); debug_sequence_x, _y both >= 0 (bridge.cc:97-104)
 * make the grid finder trace, on stderr, the sequences it tries from the candidate nearest to that pixel
 * (find_grid.cc:247-306, :515-553: "Looking at sequences from", "Considering connection ...", "rejecting" /
 * "accepting"). */
bool find_chessboard_from_image_array_C(int Nrows, int Ncols, int stride, char* imagebuffer, const int gridn,
                                        int image_pyramid_level, bool doblobs, bool debug, int debug_sequence_x,
                                        int debug_sequence_y,
                                        bool (*add_points)(double* xy, int N, void* cookie), void* cookie);

/* C face of mrgingham::find_grid_from_points (mrgingham.hh:83-87, find_grid.cc:1216-1445), host
 * only: npoints interleaved (x,y)*1000 candidates in, gridn*gridn interleaved (x,y) corners out
 * (rows top to bottom, each left to right).  false when no grid is found. */
bool mrgingham_amd_find_grid_from_points(const int* xy_scaled, int npoints, int gridn, double* xy_out);

/* The same with the reference's debug arguments (mrgingham::find_grid_from_points(..., debug, debug_sequence),
 * mrgingham.hh:83-87).  debug != 0: the finder writes its self-plotting vnlog dumps /tmp/mrgingham-2-voronoi.vnl,
 * -3-candidates(.vnl, -detailed.vnl), -4-outer-edges(.vnl, -detailed.vnl), -5-outer-edge-cycles,
 * -6-identified-outer-edge-cycle and says on stderr what it found or why it gave up (find_grid.cc:385-779, :1229-1442).
 * debug_sequence_x, _y >= 0 name a pixel; the sequences tried from the candidate nearest to it are reported on
 * stderr (find_grid.cc:247-306, :515-553). */
bool mrgingham_amd_find_grid_from_points_traced(const int* xy_scaled, int npoints, int gridn, double* xy_out,
                                                int debug, int debug_sequence_x, int debug_sequence_y);

/* TEST HOOK: the same with the parts of the visiting order that cannot be checked against the
 * reference's boost::polygon graph perturbed -- ring_seed != 0 starts every site's neighbour ring at a
 * pseudo-random position, last_match != 0 takes the last neighbour that continues a sequence instead of
 * the first (find_grid.cc:216-222).  tests/test_grid.py asserts the results do not change. */
bool mrgingham_amd_find_grid_from_points_perturbed(const int* xy_scaled, int npoints, int gridn, double* xy_out,
                                                   unsigned ring_seed, int last_match);

/* ------------------------------------------------------------------------ */
/* (2) Batch API over device-resident frames                                */
/* ------------------------------------------------------------------------ */

typedef struct mrgingham_amd_ctx mrgingham_amd_ctx;

/* A batch of equally-sized 8-bit frames in DEVICE memory:
 * frame f, row y starts at frames + f*frame_pitch + y*stride. */
typedef struct {
    const uint8_t* frames;
    int64_t frame_pitch; /* bytes between consecutive frames */
    int nframes;
    int width, height;
    int stride; /* bytes between consecutive rows */
} mrgingham_amd_frames;

enum {
    MRGINGHAM_AMD_OK = 0,
    MRGINGHAM_AMD_ERR_ARG = -1,      /* bad argument (level, sizes, NULL) */
    MRGINGHAM_AMD_ERR_DEVICE = -2,   /* HIP error; see mrgingham_amd_last_error */
    MRGINGHAM_AMD_ERR_CAPACITY = -3, /* an output capacity given by the caller was too small */
    /* -4 is reserved (MRGINGHAM_AMD_ERR_SPARSE of ABI 1: never returned since ABI 2) */
};

/* One context = one device, two HIP streams (the HBM-bound pixel kernels of a
 * batch run back to back on one, the latency-bound component kernels underneath
 * them on the other) and the per-level scratch buffers (level images, responses,
 * component tables), grown on demand and reused.  Not thread-safe: use one
 * context per host thread. */
mrgingham_amd_ctx* mrgingham_amd_create(int device_ordinal);
void mrgingham_amd_destroy(mrgingham_amd_ctx* ctx);
const char* mrgingham_amd_last_error(const mrgingham_amd_ctx* ctx);
int mrgingham_amd_abi_version(void);
/* Identity of the ChESS kernel sources the library was built from (16 hex digits of a SHA-256 over chess.hip and the
 * headers it includes): bench.py replays counter evidence collected in an earlier run (profiles/chess_l0_*.json) only
 * when that run used a library with the same id. */
const char* mrgingham_amd_kernel_id(void);
/* Number of usable HIP devices (0 = none: every entry point will fail, there is no CPU path). */
int mrgingham_amd_device_count(void);

/* How many frames the sparse refinement (option "sparse_refine") handed back to the dense kernels since the last call
 * of this function (they are repeated inside the call that met them; the outputs do not depend on it).  Synchronises. */
int mrgingham_amd_sparse_fallbacks(mrgingham_amd_ctx* ctx);

/* ---- several GPUs ---------------------------------------------------------------------------------------------
 * The reference-symbol wrappers of section (1) (and mrgingham_amd_process_image*, the file entry points) work on the
 * CALLING THREAD's context.  Which device that context lives on: MRGINGHAM_AMD_DEVICE if the variable is set (every
 * thread on that device); otherwise the k-th thread that calls into the library gets device k modulo the number of
 * devices -- the reference parallelises over worker threads with image i on worker i % N (mrgingham-from-image.cc:50,
 * :374-379), and mapped this way its workers spread over the GPUs of a node by themselves; a thread can also choose:
 * mrgingham_amd_set_thread_device (before its first call, or later: its context is then rebuilt on the new device). */
int mrgingham_amd_device_for_thread(int thread_index, int ndevices, const char* env_value);  /* the policy, as a function */
int mrgingham_amd_set_thread_device(int device_ordinal);
int mrgingham_amd_thread_device(void);  /* the device of the calling thread's context (created if need be); -1: none */

/* Host memory the device can read directly (page-locked, usable with every device): a frame handed to the wrappers
 * out of such memory is uploaded at the speed of the link, with no staging copy and no pinning on the fly (12 MB:
 * ~0.25 ms instead of ~0.45).  _register does the same for memory the caller already owns (it must stay allocated
 * until _unregister). */
void* mrgingham_amd_host_alloc(size_t bytes);
void mrgingham_amd_host_free(void* p);
int mrgingham_amd_host_register(void* p, size_t bytes);
int mrgingham_amd_host_unregister(void* p);

/* How the calling threads of this PROCESS wait for the device (hipSetDeviceFlags on every device): 0 = the runtime's
 * choice, 1 = spin, 2 = yield the core while waiting, 3 = block.  The runtime's default spins, which is the lowest
 * latency while every waiting thread has a core of its own -- and a cliff when it has not: a spinning thread burns the
 * time slice the thread that feeds the device is waiting for (32 callers on 16 cores, 4096 small images: 10-13 s
 * spinning, 4-6 s blocking, 1.4 s with four callers).  Better than any policy is FEWER callers per GPU -- the
 * command-line tool hands the images of all its workers to eight device threads per GPU --; a host that cannot
 * arrange that asks for 3.  Call it before the process creates its first context.  Returns 0, MRGINGHAM_AMD_ERR_ARG,
 * or MRGINGHAM_AMD_ERR_DEVICE when a device refused (the HIP runtime of the process is already running with another
 * policy: e.g. inside a PyTorch process). */
int mrgingham_amd_set_wait_policy(int policy);

/* Contiguous shards of `total` frames over n contexts / ranks: shard k = frames [*first, *first + *count), the first
 * total % n shards one frame longer (the split bench.py and mrgingham_amd/parallel.py use). */
int mrgingham_amd_shard_range(int total, int k, int n, int* first, int* count);

/* mrgingham_amd_chain_batch over several contexts -- one per device of a node -- in ONE call: context k takes
 * shards[k], a batch in the memory of ITS device (shards[k].nframes may be 0), and the corner lists of every shard
 * arrive in d_points / d_levels / d_npoints: buffers on the device of ctxs[0], laid out like mrgingham_amd_chain_batch's
 * for the sum of the shards' frames, shard after shard (frame-major).  A shard on the first context's device writes its
 * block in place; any other writes into buffers of its own context and the block travels device to device behind its
 * chain (a peer copy: xGMI between the GPUs of a node) -- the one exchange of the path.  Asynchronous like chain_batch;
 * consecutive calls need their own output buffers.  mrgingham_amd_sync_multi waits for every context and every gather
 * and returns the first error (MRGINGHAM_AMD_ERR_CAPACITY: as mrgingham_amd_sync -- make the call again);
 * mrgingham_amd_stream_wait_multi makes `stream` wait for them on the device instead.  Call these from one thread. */
int mrgingham_amd_chain_multi(mrgingham_amd_ctx* const* ctxs, int nctx, const mrgingham_amd_frames* shards, int start_level,
                              double* d_points, signed char* d_levels, int32_t* d_npoints, int points_pitch);
int mrgingham_amd_sync_multi(mrgingham_amd_ctx* const* ctxs, int nctx);
int mrgingham_amd_stream_wait_multi(mrgingham_amd_ctx* const* ctxs, int nctx, void* stream);

/* ---- one process per GPU: the exchange over RCCL ---------------------------------------------------------------
 * A host that runs one process (rank) per GPU calls mrgingham_amd_chain_batch on its shard with the three outputs laid
 * out in ONE device block -- mrgingham_amd_packed_layout: points at offset 0, levels at *off_levels, counts at
 * *off_npoints, *bytes in all (the layout of mrgingham_amd/parallel.py) -- and then mrgingham_amd_gather_rccl: ONE
 * ncclGather (rccl.h:745) of that block to rank `root` on `stream` (a hipStream_t of the context's device), queued behind
 * the chain on the device, asynchronous.  d_gathered (root only, may be NULL elsewhere): world x bytes, rank after rank,
 * i.e. frame-major over the global batch when ranks own contiguous shards (mrgingham_amd_shard_range).  nccl_comm is the
 * caller's ncclComm_t (this header names no RCCL type); the library does not link RCCL, it calls the ncclGather of the RCCL
 * the process already runs on.  Returns 0, MRGINGHAM_AMD_ERR_ARG, or MRGINGHAM_AMD_ERR_DEVICE (no RCCL in the process / the
 * collective failed: mrgingham_amd_last_error has RCCL's text). */
int mrgingham_amd_packed_layout(int nframes, int points_pitch, size_t* off_levels, size_t* off_npoints, size_t* bytes);
int mrgingham_amd_gather_rccl(mrgingham_amd_ctx* ctx, void* nccl_comm, int root, const void* d_packed, size_t bytes,
                              void* d_gathered, void* stream);

/* Size of pyramid level `level` of a width x height frame: what
 * cv::resize(.., 1/2^level, 1/2^level) produces (find_chessboard_corners.cc:449-450). */
int mrgingham_amd_level_dims(int width, int height, int level, int* w, int* h);

/* Dense ChESS response of every frame at pyramid level `level` (0 = the frame
 * itself): d_response receives nframes dense w_L x h_L int16 images back to
 * back.  clamp = 0: the raw reference response in the interior (ChESS.c:104),
 * zeros in the 7-pixel frame.  clamp = 1: negatives replaced by 0, i.e. the
 * buffer the reference's component search starts from
 * (find_chessboard_corners.cc:506-529).  `stream` is a hipStream_t, used as given
 * (NULL = HIP's default stream); the call is asynchronous on it. */
int mrgingham_amd_chess_response_batch(mrgingham_amd_ctx* ctx, const mrgingham_amd_frames* frames, int level,
                                       int clamp, int16_t* d_response, void* stream);

/* Pyramid level images alone (find_chessboard_corners.cc:445-452): d_out
 * receives nframes dense w_L x h_L byte images. */
int mrgingham_amd_decimate_batch(mrgingham_amd_ctx* ctx, const mrgingham_amd_frames* frames, int level,
                                 uint8_t* d_out, void* stream);

/* 3x3.. box blur the reference CLI applies before detection
 * (mrgingham-from-image.cc:106-111): d_out receives nframes dense byte images. */
int mrgingham_amd_box_blur_batch(mrgingham_amd_ctx* ctx, const mrgingham_amd_frames* frames, int radius,
                                 uint8_t* d_out, void* stream);

/* The contrast preprocessing the reference CLI applies to an 8-bit frame before
 * detection (mrgingham-from-image.cc:38-45, :71-79, :106-111): when do_clahe,
 * cv::normalize(0, 255, NORM_MINMAX) then CLAHE (clip limit 8, 8x8 tiles); then a
 * (2*blur_radius+1)^2 box blur (0 = none).  d_out receives nframes dense byte
 * images and must not alias the input.  OpenCV arithmetic: parity unpinned. */
int mrgingham_amd_preprocess_batch(mrgingham_amd_ctx* ctx, const mrgingham_amd_frames* frames, int do_clahe,
                                   int blur_radius, uint8_t* d_out, void* stream);

/* find_chessboard_corners_from_image_array for every frame of the batch at one
 * pyramid level.  Device outputs: d_xy holds nframes blocks of
 * capacity_per_frame interleaved (x,y)*1000 int pairs in the reference's
 * order, d_counts[f] the number of candidates of frame f (may exceed
 * capacity_per_frame: then only the first capacity_per_frame were stored).
 * Asynchronous on the context's streams; mrgingham_amd_sync() waits. */
int mrgingham_amd_detect_batch(mrgingham_amd_ctx* ctx, const mrgingham_amd_frames* frames, int level,
                               int32_t* d_xy, int capacity_per_frame, int32_t* d_counts);

/* refine_chessboard_corners_from_image_array for every frame at one level.
 * d_points: nframes blocks of points_pitch interleaved (x,y) doubles;
 * d_levels: nframes blocks of points_pitch signed chars; d_npoints[f] points
 * are live in frame f.  Updated in place; d_nrefined[f] (may be NULL) receives
 * the reference's return value. */
int mrgingham_amd_refine_batch(mrgingham_amd_ctx* ctx, const mrgingham_amd_frames* frames, int level,
                               double* d_points, signed char* d_levels, const int32_t* d_npoints,
                               int points_pitch, int32_t* d_nrefined);

/* The reference's found-frame schedule for image_pyramid_level < 0 with the
 * grid finder left on the host (mrgingham.cc:50, :81-99): detect at
 * start_level, take every candidate as a corner ((double)x/1000,
 * find_grid.cc:353-354), then refine through start_level-1 .. 0.  Outputs as
 * mrgingham_amd_refine_batch; d_npoints[f] receives the candidate count. */
int mrgingham_amd_chain_batch(mrgingham_amd_ctx* ctx, const mrgingham_amd_frames* frames, int start_level,
                              double* d_points, signed char* d_levels, int32_t* d_npoints, int points_pitch);

/* The connected-component stage ALONE, on caller-supplied responses: what
 * process_connected_components (find_chessboard_corners.cc:284-397) does with the buffer the
 * reference hands it, for the rule tests (running-maximum order dependence, thresholds 15 / 120,
 * N >= 2, margin columns, variance window, response mutation between refined points) that a natural
 * image only meets by chance.  d_response: nframes dense w x h int16 images (device), used the way
 * the reference's buffer arrives there: negatives count as 0 (:527-529) and everything outside
 * [7,w-7) x [7,h-7) as 0 (:506; the ChESS pass never writes there) -- the input is not modified.
 * d_level_image: nframes dense w x h byte images, the image the variance test reads (:50-88).
 * `level` only scales the output coordinates (:319, :346, :369, :390).  Exactly one mode:
 *   detect  d_xy / capacity_per_frame / d_counts as mrgingham_amd_detect_batch, d_points NULL;
 *   refine  d_points / d_levels / d_npoints / points_pitch / d_nrefined as
 *           mrgingham_amd_refine_batch, d_xy NULL.
 * Asynchronous like the other batch calls. */
int mrgingham_amd_cc_on_response_batch(mrgingham_amd_ctx* ctx, const int16_t* d_response,
                                       const uint8_t* d_level_image, int nframes, int w, int h, int level,
                                       int32_t* d_xy, int capacity_per_frame, int32_t* d_counts,
                                       double* d_points, signed char* d_levels, const int32_t* d_npoints,
                                       int points_pitch, int32_t* d_nrefined);

/* C faces of find_chessboard_corners_from_image_file (find_chessboard_corners.hh:32-44, .cc:623-648)
 * and find_chessboard_from_image_file (mrgingham.hh:77-83, mrgingham.cc:145-170): the image file is
 * decoded (binary PGM or non-interlaced PNG; the reference uses cv::imread) and handed to the array
 * functions above.  false when the file cannot be read or nothing is found. */
bool find_chessboard_corners_from_image_file_C(const char* filename, int image_pyramid_level, bool debug,
                                               bool (*add_points)(int* xy, int N, double scale, void* cookie),
                                               void* cookie);
bool find_chessboard_from_image_file_C(const char* filename, const int gridn, int image_pyramid_level, bool debug,
                                       bool (*add_points)(double* xy, int N, void* cookie), void* cookie);

/* The image decoder behind the two functions above and the command-line tool, on its own (host only, no
 * device needed): binary PGM (8 / 16 bit) and non-interlaced PNG.  `out` (may be NULL: sizes only)
 * receives width*height grey bytes.  16-bit samples: cli_scaling = 0 keeps the high byte, what
 * cv::imread(IMREAD_GRAYSCALE) gives the file entry points; 1 rescales by 255/65535 with rounding, what
 * the CLI's convertTo does (mrgingham-from-image.cc:85-92).  Returns 0; -1 unreadable, unsupported or
 * malformed file (never throws, never reads or writes out of bounds on a crafted file; sides above
 * 32767 are rejected); -2 out_capacity too small (sizes are still reported). */
int mrgingham_amd_read_image(const char* filename, int cli_scaling, uint8_t* out, size_t out_capacity, int* width,
                             int* height, int* depth);

/* The same preprocessing for one HOST image (out: dense width x height bytes, host): what the Python
 * recipe of find_board.docstring:8-10 does with cv2 before find_board.  Uses the calling thread's
 * context.  Returns 0, or -2 on an argument or device error. */
int mrgingham_amd_preprocess_image(const uint8_t* image, int width, int height, int stride, int do_clahe,
                                   int blur_radius, uint8_t* out);

/* The 16-bit branch of the tool's preprocessing for one HOST image (mrgingham-from-image.cc:85-111; `stride`
 * in elements): out receives width*height bytes.  Returns 0, or -2. */
int mrgingham_amd_preprocess_image16(const uint16_t* image, int width, int height, int stride, int do_clahe,
                                     int blur_radius, uint8_t* out);

/* What one worker of the reference CLI does with one decoded 8-bit image
 * (mrgingham-from-image.cc:71-111, :160-171): [normalize + CLAHE(8)] -> box blur of
 * blur_radius -> find_chessboard_from_image_array(gridn, image_pyramid_level), with the
 * per-corner refinement when do_refine.  Host image in, gridn*gridn corners (xy_out,
 * interleaved doubles) and their refinement levels (levels_out, may be NULL) out.  Uses the
 * calling thread's context.  Returns the level the board was found at, -1 if there is
 * none, -2 on an argument or device error. */
int mrgingham_amd_process_image(const uint8_t* image, int width, int height, int stride, int do_clahe,
                                int blur_radius, int gridn, int image_pyramid_level, int do_refine,
                                double* xy_out, signed char* levels_out);

/* The same with the tool's whole option set, and for 16-bit images (mrgingham-from-image.cc:85-92:
 * cv::normalize to 0..65535 and CLAHE on 16 bits when do_clahe, then convertTo(CV_8U, 255/65535)).
 * bits = 8 or 16; `stride` in ELEMENTS.  debug != 0 writes the reference's dumps: the preprocessed
 * image as /tmp/<basename of filename>_preprocessed.png (:113-148) and, per detector / refinement pass,
 * the level image, the normalised ChESS responses and a self-plotting corner vnlog
 * (find_chessboard_corners.cc:282-315, :453-459, :513-541) -- same names, same messages on stderr. */
typedef struct mrgingham_amd_cli_options {
    int do_clahe, blur_radius, gridn, image_pyramid_level, do_refine, do_blobs, debug;
    int debug_sequence_x, debug_sequence_y; /* both >= 0: the grid finder's sequence trace on stderr (--debug-sequence) */
    const char* filename;                   /* names the debug dumps only; may be NULL */
} mrgingham_amd_cli_options;
int mrgingham_amd_process_image_ex(const void* image, int bits, int width, int height, int stride,
                                   const mrgingham_amd_cli_options* options, double* xy_out,
                                   signed char* levels_out);

/* Batch form of the full detector (what find_chessboard_from_image_array_C does per frame, for
 * image_pyramid_level < 0 the reference's default "first level of 3,2,1,0 at which the grid finder
 * succeeds", mrgingham.cc:116-139): the GPU detects candidates for the whole batch (levels 3, 2 and 1 in one pass; what
 * is still open after them level by level), up to `nthreads` host threads (<= 0: all cores, at most 32) run the grid
 * finder on the frames that have no board yet, and the boards found are refined to level 0 on the GPU.  Synchronous.
 * h_boards (HOST): nframes x gridn*gridn x 2 doubles, board order; h_found_level[f] (HOST): the
 * level frame f's grid was found at, or -1 (then its h_boards block is left untouched). */
int mrgingham_amd_find_boards_batch(mrgingham_amd_ctx* ctx, const mrgingham_amd_frames* frames, int gridn,
                                    int image_pyramid_level, double* h_boards, signed char* h_found_level,
                                    int nthreads);

/* The same in two halves, so that consecutive batches overlap: _submit queues the device passes of a batch (level
 * images, candidates of levels 3, 2 and 1 -- or of the one level asked for) and returns a ticket (>= 0; < 0: an error code)
 * without waiting for them; _collect(ticket) returns when h_boards / h_found_level of that batch are complete (status as
 * mrgingham_amd_find_boards_batch, which is _submit + _collect).  Between the two, the library runs the host part of
 * a batch -- the grid finder on the candidates, level 3 first, then 2, then 1 (mrgingham.cc:127-138) -- inside the NEXT
 * _submit, after that call has queued its own device passes, and queues the refinement of the boards it found
 * (mrgingham.cc:81-99; with option "sparse_refine": the response only around the corners) on a stream of its own.  So
 *     t0 = submit(batch 0); t1 = submit(batch 1); loop: t(n+2) = submit(batch n+2); collect(t(n)); ...
 * keeps the device, the host threads and the refinement busy at the same time (64 frames of 4096x3072 on one
 * MI355X: see DESIGN.md 4.5).  Frames that show no board at levels 3, 2 and 1 are finished at level 0 inside
 * _collect.  Results do not depend on how calls are interleaved: they are the synchronous dense schedule's (option
 * "find_boards_pipeline" 0), double for double.
 * The frames, h_boards and h_found_level of a batch must stay valid and untouched until its _collect returns.  As many
 * batches as the context has scratch sets (2 or 3, option "scratch_sets") can be in flight; a further _submit completes
 * the oldest one first (its _collect then returns at once).  One thread per context, as for every other call.  Any
 * other batch call or mrgingham_amd_set_option on the same context completes the batches in flight first (they stay
 * collectable), so mixing calls is safe but gives the overlap away; mrgingham_amd_destroy abandons what is in flight. */
int mrgingham_amd_find_boards_submit(mrgingham_amd_ctx* ctx, const mrgingham_amd_frames* frames, int gridn,
                                     int image_pyramid_level, double* h_boards, signed char* h_found_level,
                                     int nthreads);
int mrgingham_amd_find_boards_collect(mrgingham_amd_ctx* ctx, int ticket);

/* Where the find_boards calls of this context spent their HOST time since the last reset, and what their grid-finder
 * threads did (the reference's find_grid_from_points, mrgingham.cc:51, is the host part of the product call; on a busy
 * batch it is what bounds it).  out[0 .. n) receives up to MRGINGHAM_AMD_FB_STATS doubles:
 *   [0] batches submitted   [1] host threads of the most recent batch (grid-finder workers + the calling thread)
 *   host milliseconds, totals over the batches:
 *   [2] submit: checks + scratch   [3] submit: the previous batch's host part begun (waits for its first device pass)
 *   [4] submit: this batch's device passes queued   [5] host part: grid finder joined (wall time the caller waits)
 *   [6] host part: refinement queued   [7] collect: wait for the refinement   [8] collect: boards copied
 *   grid finder, summed over the threads:
 *   [9] calls   [10] calls that found a grid   microseconds in [11] neighbour graph (sort, Delaunay, site rings)
 *   [12] adjacency lists   [13] sequence-candidate search   [14] outer edges, 4-cycles, rows
 *   device milliseconds (hipEvents), totals over the batches:
 *   [15] first pass: from its first kernel on the pixel stream to the candidates on the host (runs behind the pass of the
 *        batch before it, so in a full pipeline this is the pass's own time)   [16] refinement: boards up, levels, boards down
 * Completes the batches in flight first (they stay collectable).  Returns MRGINGHAM_AMD_FB_STATS, or an error code. */
#define MRGINGHAM_AMD_FB_STATS 17
int mrgingham_amd_find_boards_stats(mrgingham_amd_ctx* ctx, double* out, int n, int reset);
/* The same clock of the CALLING thread's own mrgingham_amd_find_grid_from_points* calls (host only, no device):
 * out6 = calls, found, then the four microsecond sums. */
int mrgingham_amd_grid_clock(double* out6, int reset);

/* TEST HOOK: which implementation of the component search handled each frame of the most recent call at
 * `level`: h_paths[f] = 1 out of LDS, 0 the global-memory kernels (hot pixels that cannot be cut into bands
 * of at most 2048, more than 512 multi-pixel components per band / points, or more LIFO demand than the LDS
 * tables hold), 2 = a refinement the LDS kernel began band by band and the global-memory kernel finished.
 * Synchronises. */
int mrgingham_amd_debug_paths(mrgingham_amd_ctx* ctx, int level, int nframes, int32_t* h_paths);

/* TEST HOOK: with option "cc_lds" = 1 | 512 the LDS refinement kernel leaves a phase clock of the first frame of
 * the call: h_ticks12[0..7] = 100 MHz ticks at start, bands planned, hot list loaded + labelled, seeds found,
 * groups formed, LIFO demand known + neighbour table built, fills done (all of the first band), end;
 * [8..11] = hot pixels, points, bands, level.  Synchronises.  (tools/cc_phases.py) */
int mrgingham_amd_debug_refine_clock(mrgingham_amd_ctx* ctx, long long* h_ticks12);

/* Device memory the context currently holds (level scratch of both sets, point scratch, staging). */
long long mrgingham_amd_scratch_bytes(const mrgingham_amd_ctx* ctx);

/* How the most recent mrgingham_amd_chain_batch call was launched (for benchmarks that price the level-0
 * kernel): *fused_pyramid = 1 when the level-0 response kernel also wrote the level images 1..3 (frames of
 * whole 16 x 8 blocks, option "fuse_pyramid"), 0 when a separate pyramid kernel did; *merged_levels = number
 * of levels whose responses shared one launch (0 = one launch per level; -1 = the call ran with option
 * "sparse_refine", where the timed launch is the one that writes the level images).  Either pointer may be NULL. */
int mrgingham_amd_chain_info(const mrgingham_amd_ctx* ctx, int* fused_pyramid, int* merged_levels);

/* Tunables outside the reference's surface.  Known names:
 *   "fuse_pyramid"        1 (default): chain calls on frames of whole 16 x 8 blocks take the level images
 *                         1..3 out of the level-0 response kernel; 0: separate pyramid kernel
 *   "scratch_sets"        how many calls' component searches may be in flight while the pixel kernels of the next call
 *                         run (each set is a full copy of the level scratch): 0 (default) = chosen per batch shape --
 *                         three while three sets stay below 8 GB (small frames, whose search chain is longer than two
 *                         steps of the pixel kernels; 16 GB once option "sparse_refine" has been on), two otherwise --,
 *                         2 or 3 = fixed.  Synchronises.
 *   "hot_capacity_shift"  per-frame capacity of the hot-pixel / component tables of a pyramid level is
 *                         (level width * height) >> shift entries, at least 4096 (default 7: 0.35 bytes of tables per
 *                         pixel; 0 = one entry per pixel).  Derived limits at shift > 0: candidates per frame
 *                         capacity / 16 + 1024, LIFO words 1.25 * capacity + 16384.  A frame that exceeds one of
 *                         them makes the call fail with MRGINGHAM_AMD_ERR_CAPACITY at the next mrgingham_amd_sync --
 *                         nothing is written for that frame (detect: count -1; refine: its points keep their values
 *                         and levels, except on a LIFO overflow in the middle of a frame, where the points refined so
 *                         far stay refined) -- and the tables of that level GROW to what the frame asked for, so
 *                         the same call succeeds when it is made again (refinement is idempotent: points already at
 *                         the level are skipped).  Setting the option resets what has grown
 *                         ("hot_capacity_shift_temporary": the same without that reset -- for a caller's own last-resort
 *                         retry at shift 0 and the way back).
 *   "multi_level_launch"  chain_batch: 0 = one ChESS launch per pyramid level, 1 (default) = levels 3..1 in one
 *                         launch, 2 = all levels in one launch
 *   "cc_lds"              1 (default) = component search out of LDS: frames with at most 2048 hot pixels in one
 *                         pass, frames with up to 16384 in bands of rows separated by three rows without a hot
 *                         pixel, refinement of frames with more than that on the hot pixels in the cells around
 *                         the points (the global-memory kernels take what is left), 0 = global-memory kernels only;
 *                         1 | 256 = neither bands nor cells (test hook; results are the same).  Any other value is
 *                         refused.
 *   "sparse_subsets"      1 .. 4 (default 2): workgroups per frame of the refinement kernel of a sparse level.  A frame with
 *                         at least 128 points to refine at a level (a 14x14 board) is cut into that many subsets of points
 *                         that are at least 64 pixels apart, refined side by side (64 x 4096x3072, 14x14: 0.55 -> 0.46 ms per
 *                         sparse step; a 10x10 board stays with one workgroup, which is faster there).  Results do not depend on it.
 *   "sparse_refine"       1 (default), 0, 2: chain_batch computes the response of the levels BELOW the start level only in
 *                         the 16 x 16 cells around the points it refines there (all level images and the start level's
 *                         response stay whole-frame): 2 = always, 1 = for calls of at least 96 Mi frame pixels (smaller
 *                         calls are faster dense), 0 = never (the dense per-level response of the reference, what bench.py's
 *                         `value` is measured with).  Same outputs on every frame: a frame the sparse kernels cannot take
 *                         -- a component that leaves the cells around its point, more than 512 points -- is repeated densely
 *                         by the library, on the device, inside the same call (mrgingham_amd_sparse_fallbacks counts them).
 *   "find_boards_pipeline" 1 (default): mrgingham_amd_find_boards_batch / _submit / _collect as described there; 0: the
 *                         synchronous schedule (one level at a time for the whole batch, dense refinement) -- same results
 *   "chess_seg", "chess16_seg"  rows per workgroup of the ChESS kernels of THIS context (chess_v1* / chess_v16; 0 = cost model,
 *                         the default): the frame is cut into ceil(height / value) row segments of equal height (whole
 *                         8- / 16-row groups); results do not depend on it
 *   "chess_variant"       the response without a hot list: 0 (default) = chess_v16_kernel where it pays, 1 = chess_v1 always,
 *                         16 = chess_v16 wherever it can run (widths that are multiples of 16); same results
 *   "preprocess_fused"    1 (default): mrgingham_amd_preprocess_batch(do_clahe, blur_radius 1) -- the reference tool's default
 *                         chain -- blends the CLAHE tile LUTs and blurs in ONE pass over the frame where the geometry allows
 *                         (rows of 16-byte multiples, tiles at least 34 rows high); 0: always two kernels.  Same bytes.
 * Builds made with -DMRG_EXPERIMENT (make -C mrgingham_amd/csrc EXPERIMENT=1 -> libmrgingham_amd_experiment.so)
 * additionally accept the timing ablations and phase clocks of tools/ ("cc_lds" bits 2, 4, 8, 16, 128, 512,
 * "cc_schedule", "chess_stage", "chess_multi_min_blocks", "chess_v0" = the reference-shaped ChESS kernel as an on-device
 * cross-check, the MRGINGHAM_AMD_PYR_SKIP / _CC_LDS_PAD / _CC_CUS /
 * _PIX_COMPLEMENT / _CHESS_V0 environment variables): some of them produce wrong results on purpose, which is why the
 * shipped library has none of them and reads no environment variable except MRGINGHAM_AMD_DEVICE. */
int mrgingham_amd_set_option(mrgingham_amd_ctx* ctx, const char* name, int value);

/* Wait for everything queued on the context's streams; returns the first
 * asynchronous error.  MRGINGHAM_AMD_ERR_CAPACITY here means a frame had more
 * hot pixels (or candidates) than the component tables of its level hold (dense
 * texture): the tables have grown by the time this returns, make the same call
 * again (see "hot_capacity_shift").  The reference-symbol wrappers in section (1)
 * and mrgingham_amd_find_boards_batch do that themselves. */
int mrgingham_amd_sync(mrgingham_amd_ctx* ctx);

/* Device-side alternative to mrgingham_amd_sync for pipelines: makes `stream` (a hipStream_t,
 * NULL = the default stream) wait for the most recently queued detect / refine / chain call,
 * without blocking the host.  Consecutive calls overlap (the pixel kernels of call N+1 run while
 * the component kernels of call N finish, on alternating scratch sets), so give each call in
 * flight its own output buffers. */
int mrgingham_amd_stream_wait(mrgingham_amd_ctx* ctx, void* stream);

/* The other direction: the NEXT detect / refine / chain call queued on the context starts only after
 * everything queued so far on `stream` (e.g. the host-to-device copy of its frames on a copy stream)
 * has completed.  Does not block the host. */
int mrgingham_amd_after_stream(mrgingham_amd_ctx* ctx, void* stream);

/* Average duration in milliseconds of the dominant kernel (the level-0 ChESS
 * response kernel) over the launches issued since the last call, measured with
 * hipEvents on the streams the kernel ran on; the number of launches is stored
 * in *nlaunches.  Timing is off by default: enable with
 * mrgingham_amd_set_kernel_timing(ctx, 1); 2 = only the engine-clock probe below (no events on the streams), 0 = off. */
void mrgingham_amd_set_kernel_timing(mrgingham_amd_ctx* ctx, int enable);
double mrgingham_amd_chess_kernel_ms(mrgingham_amd_ctx* ctx, int* nlaunches);

/* The engine clock the level-0 response kernels (and the fused blend + blur kernel of mrgingham_amd_preprocess_batch) ACTUALLY ran at, in MHz, averaged over the launches issued since the last
 * call while kernel timing was enabled (1 or 2): workgroup 0 of each launch reads the shader-cycle counter (s_memtime) and the
 * constant-rate counter (s_memrealtime, hipDeviceAttributeWallClockRate) at its start and end and adds the two differences
 * to a pair of device counters -- four scalar instructions in one workgroup of the launch.  A VALU-bound kernel scales with
 * this clock, so a benchmark line that carries it can tell a slow box from a regression.  0 when nothing was probed.
 * Synchronises the device. */
double mrgingham_amd_sclk_mhz(mrgingham_amd_ctx* ctx);

#ifdef __cplusplus
}
#endif
