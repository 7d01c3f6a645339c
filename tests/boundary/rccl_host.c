/* BOUNDARY example ("one process per GPU, C++ host"): what a rank of a host that runs one process per GPU
 * does with this library and RCCL -- its shard of the frames through mrgingham_amd_chain_batch into ONE packed block, then
 * ONE ncclGather of that block to rank 0 (mrgingham_amd_gather_rccl).  Plain C against include/mrgingham_amd.h, the HIP
 * runtime API and rccl.h; tests/test_c_client.py compiles it (syntax and types; it is run where ranks and GPUs exist). */
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>
#include <stdint.h>
#include <stdio.h>
#include "mrgingham_amd.h"

/* rank `rank` of `world`; `comm` from ncclCommInitRank; d_frames: this rank's `count` frames of W x H bytes on `gpu`;
 * returns 0 and, on rank 0, *d_gathered_out = world blocks of *block_bytes (rank after rank = frame-major) */
int run_rank(int rank, int world, int gpu, ncclComm_t comm, const uint8_t* d_frames, int total_frames, int W, int H,
             int points_pitch, void** d_gathered_out, size_t* block_bytes) {
    int first = 0, count = 0;
    size_t o_lv = 0, o_np = 0, bytes = 0;
    char* d_pack = NULL;
    char* d_gathered = NULL;
    hipStream_t stream;
    mrgingham_amd_frames fr;
    mrgingham_amd_ctx* ctx = mrgingham_amd_create(gpu);
    if (!ctx) return -1;
    if (mrgingham_amd_shard_range(total_frames, rank, world, &first, &count) != 0) return -2;
    /* equal blocks from every rank: lay the block out for the longest shard (the first ones are one frame longer) */
    if (mrgingham_amd_packed_layout((total_frames + world - 1) / world, points_pitch, &o_lv, &o_np, &bytes) != 0) return -3;
    if (hipSetDevice(gpu) != hipSuccess || hipStreamCreate(&stream) != hipSuccess) return -4;
    if (hipMalloc((void**)&d_pack, bytes) != hipSuccess || hipMemset(d_pack, 0, bytes) != hipSuccess) return -5;
    if (rank == 0 && hipMalloc((void**)&d_gathered, (size_t)world * bytes) != hipSuccess) return -6;
    fr.frames = d_frames; fr.frame_pitch = (int64_t)W * H; fr.nframes = count; fr.width = W; fr.height = H; fr.stride = W;
    if (mrgingham_amd_chain_batch(ctx, &fr, 3, (double*)d_pack, (signed char*)(d_pack + o_lv), (int32_t*)(d_pack + o_np),
                                  points_pitch) != 0 ||
        mrgingham_amd_gather_rccl(ctx, comm, 0, d_pack, bytes, d_gathered, stream) != 0) {
        fprintf(stderr, "rank %d: %s\n", rank, mrgingham_amd_last_error(ctx));
        return -7;
    }
    if (hipStreamSynchronize(stream) != hipSuccess) return -8;
    if (mrgingham_amd_sync(ctx) != 0) return -9;   /* status words: MRGINGHAM_AMD_ERR_CAPACITY -> make the call again */
    *d_gathered_out = d_gathered;
    *block_bytes = bytes;
    (void)first;
    hipFree(d_pack);
    mrgingham_amd_destroy(ctx);
    return 0;
}
