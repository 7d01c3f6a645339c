// Launchers of the HIP kernels (host side declarations).
#pragma once
#include <functional>
#include <string>
#include <vector>

#include "common.h"

namespace mrg {

struct PyramidOut {
    uint8_t* out[3];  // dense level images of levels 1..3 (NULL = not wanted), frames back to back
    int w[3], h[3];
};

// chess.hip
#ifdef MRG_EXPERIMENT
void launch_chess_v0(const LevelBatch& lb, const CompTables& t, int frame0, int nframes, bool clamp, bool hot,
                     hipStream_t s);
#endif
// `seg_rows` (here and below): 0 = the launcher picks the row segments, > 0 = the caller's segment height (the
// per-context options "chess_seg" / "chess16_seg")
void launch_chess(const LevelBatch& lb, const CompTables& t, int frame0, int nframes, bool clamp, bool hot,
                  hipStream_t s, int seg_rows = 0);

// sparse refinement: the response + hot list in a per-frame list of cells only (cells listed by launch_sparse_cells, cc.hip)
void launch_chess_cells(const LevelBatch& lb, const CompTables& t, const uint32_t* cell_list, const int32_t* cell_cnt,
                        int list_pitch, int frame0, int nframes, hipStream_t s);

// level 0 of a chain with the level images 1..3 produced by the same kernel (out of its LDS ring)
bool chess_pyramid_ok(const LevelBatch& lb, int nframes);
bool launch_chess_pyramid(const LevelBatch& lb, const CompTables& t, const PyramidOut& po, int nframes, hipStream_t s,
                          int seg_rows = 0);

bool chess_multi_ok(const LevelBatch* lbs, int n, int nframes);
bool launch_chess_multi(const LevelBatch* lbs, const CompTables* ts, int n, int nframes, hipStream_t s, int seg_rows = 0);
extern int chess_stage_override;
extern int chess_multi_min_blocks;
extern int pyramid_lds_pad;

// chess16.hip: the response with sixteen pixels per lane (widths that are multiples of 16; no hot list)
bool chess16_ok(const LevelBatch& lb);
bool chess16_pays(const LevelBatch& lb, int nframes);
void launch_chess16(const LevelBatch& lb, int frame0, int nframes, bool clamp, hipStream_t s, int seg_rows = 0);
#ifdef MRG_EXPERIMENT
extern int chess16_pair;
void launch_chess16_hot(const LevelBatch& lb, const CompTables& t, int frame0, int nframes, hipStream_t s);
bool launch_chess16_pyramid(const LevelBatch& lb, const CompTables& t, const PyramidOut& po, int nframes, hipStream_t s);
bool chess16_multi_ok(const LevelBatch* lbs, int n, int nframes);
bool launch_chess16_multi(const LevelBatch* lbs, const CompTables* ts, int n, int nframes, hipStream_t s);
#endif

// decimate.hip
struct FrameBatch {
    const uint8_t* frames;
    long long frame_pitch;
    int width, height, stride;
};
void launch_decimate(const FrameBatch& in, int level, uint8_t* out, long long out_pitch, int ow, int oh, int frame0,
                     int nframes, hipStream_t s);
void launch_pyramid(const FrameBatch& in, const PyramidOut& po, int top, int nframes, hipStream_t s, bool gentle = false);
void launch_box_blur(const FrameBatch& in, int radius, uint8_t* out, int frame0, int nframes, hipStream_t s);

// preprocess.hip: cv::normalize(0..255, NORM_MINMAX) + CLAHE (8x8 tiles), mrgingham-from-image.cc:71-79
size_t clahe_scratch_bytes(int nframes);
// `blur3`: followed by cv::blur(3x3), in the same pass where clahe_blur3_fused() says so, else through `tmp`
extern int clahe_hist_copies;
bool clahe_blur3_fused(const FrameBatch& in, const uint8_t* out);
bool launch_clahe(const FrameBatch& in, int nframes, double clip_limit, bool do_normalize, uint8_t* out,
                  void* scratch, hipStream_t s, bool blur3 = false, uint8_t* tmp = nullptr, unsigned long long* clk = nullptr);

// preprocess16.hip: the CLI's 16-bit branch (normalize to 0..65535, CLAHE on 16 bits, convertTo 8 bit)
size_t preprocess16_scratch_bytes(int nframes, int w, int h);
bool launch_preprocess16(const uint16_t* frames, long long pitch, int nframes, int w, int h, int stride, bool do_clahe,
                         double clip_limit, uint8_t* out8, void* scratch, hipStream_t s);

// blobs.hip: cv::SimpleBlobDetector as find_blobs.cc:14-46 configures it (device border following, host filters)
struct BlobScratchLayout {
    int wpr;
    size_t o_counters, o_bits, o_wordpre, o_rowcnt, o_rowoff;
};
size_t blob_scratch_bytes(int w, int h, BlobScratchLayout* lay);
bool blob_detect(const uint8_t* d_img, int d_stride, const uint8_t* h_img, int h_stride, int w, int h, void* scratch,
                 const std::function<void*(size_t)>& node_scratch, const std::function<void*(size_t)>& out_scratch, hipStream_t s,
                 std::vector<int32_t>& xy_out, std::string& err);

// cc.hip
struct DetectOut {
    int32_t* xy;      // [nframes*capacity*2]
    int capacity;
    int32_t* counts;  // [nframes]
    // optional (chain): the candidates again as corners for the refinement below
    double* points = nullptr;       // [nframes*points_pitch*2]
    signed char* levels = nullptr;  // [nframes*points_pitch]
    int32_t* npoints = nullptr;     // [nframes]
    int points_pitch = 0;
};
struct RefineIO {
    double* points;          // [nframes*pitch*2]
    signed char* levels;     // [nframes*pitch]
    const int32_t* npoints;  // [nframes]
    int pitch;
    int32_t* nrefined;       // [nframes] or NULL
    // scratch, per frame `pitch` entries unless noted
    int32_t* leader;
    int32_t* need;
    int32_t* nseeds;
    uint32_t* seeds;         // [nframes*pitch*9]
    int32_t* sroot;          // [nframes*pitch*9] root of each seed's super-component
    // sparse refinement (lds_path bit kLdsPathSparse): the cells whose response was computed (sparse_cells_kernel)
    uint32_t* cell_list = nullptr;        // [nframes*list_pitch]
    const int32_t* cell_cnt = nullptr;    // [nframes*kCellHdr] of this level (common.h)
    int list_pitch = 0;
    // the refinement kernel of level L lists the cells of level L - 1 itself when it is done, into next_list (the OTHER of two
    // buffers: a frame may have several workgroups, and one may be done while the others still read cell_list); next_cnt =
    // NULL at level 0
    int32_t* next_cnt = nullptr;
    uint32_t* next_list = nullptr;
    int subsets = 1;  // workgroups per frame of the sparse refinement kernel (cc.hip, "Several workgroups"): 1 .. 4
    int next_w = 0, next_h = 0;
    long long next_max_items = 0;
};
void launch_hot_from_response(const int16_t* src, const LevelBatch& lb, const CompTables& t, int frame0, int nframes,
                              hipStream_t s);
// several levels of the same frames in one grid per kernel (the first pass of the full detector)
constexpr int kDetectLevelsMax = 3;
struct DetectLevels {
    LevelBatch lb[kDetectLevelsMax];
    CompTables t[kDetectLevelsMax];
    DetectOut out[kDetectLevelsMax];
    int level[kDetectLevelsMax];
};
void launch_cc_detect_levels(const LevelBatch* lbs, const CompTables* ts, const int* levels, const DetectOut* outs, int nlevels,
                             int nframes, hipStream_t s);
void launch_cc_detect(const LevelBatch& lb, const CompTables& t, int level, const DetectOut& out, int frame0,
                      int nframes, hipStream_t s);
void launch_cc_refine(const LevelBatch& lb, const CompTables& t, int level, const RefineIO& io, int frame0,
                      int nframes, hipStream_t s);
// sparse refinement: the cells (squares of 16 pixels; 32, 64 ... when the box around the points has more than 40 960 of them) around the points to refine at
// `level`, per frame: cell_list[frame * list_pitch + k], k < cell_cnt[frame]
// cnt_all / nframes_all: the cell-list headers of every level, [level][frame][kCellHdr]: those of the levels below `level` are zeroed
void launch_sparse_cells(const LevelBatch& lb, const CompTables& t, int level, const RefineIO& io, uint32_t* cell_list,
                         int32_t* cell_cnt, int list_pitch, int frame0, int nframes, hipStream_t s, int32_t* cnt_all, int nframes_all);
constexpr int kLdsPathSparse = 1024;  // CompTables::lds_path bit: the dense response only holds those cells
// Sparse refinement, the frames it could not take (kStatusSparse in their level-0 status word): REPEATED DENSELY,
// on the device, in three small launches (api.hip, queue_sparse_levels):
//   launch_sparse_flag_list          the frames as a list (list[0] = how many, list[1 ..] = which);
//   launch_chess / launch_chess_multi with CompTables::only = that list: their dense responses + hot lists;
//   launch_cc_refine_flagged_levels  per listed frame: its points back to what they were before the first sparse
//                                    level -- the detection's candidates (xy != NULL: (double)xy / 1000 at `level`,
//                                    find_grid.cc:353-354) or a copy the caller kept (pts0 / lv0) --, the refinement
//                                    of levels nlevels-1 .. 0 with the global-memory kernel's body (lbs / ts indexed
//                                    by level), the flag cleared from the status words of those levels (words of
//                                    consecutive levels are level_stride apart) and the frame counted in *counter.
struct SparseRestore {
    const int32_t* xy;  // [nframes * xy_pitch * 2] or NULL
    int xy_pitch, level;
    const double* pts0;  // [nframes * pitch * 2]
    const signed char* lv0;
};
constexpr int kRefineLevelsMax = 4;
void launch_sparse_flag_list(const int32_t* status_level0, int nframes, int32_t* list, hipStream_t s);
void launch_cc_refine_flagged_levels(const LevelBatch* lbs, const CompTables* ts, int nlevels, const RefineIO& io,
                                     const SparseRestore& restore, const int32_t* list, int32_t* status_level0,
                                     int level_stride, int32_t* counter, hipStream_t s);

}  // namespace mrg
