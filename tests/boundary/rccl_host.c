/* BOUNDARY example ("one process per GPU, C++ host"): what a rank of a host that runs one process per GPU
 * does with this library and RCCL -- its shard of the frames through mrgingham_amd_chain_batch into ONE packed block, then
 * ONE ncclGather of that block to rank 0 (mrgingham_amd_gather_rccl).  Plain C against include/mrgingham_amd.h, the HIP
 * runtime API and rccl.h; tests/test_c_client.py compiles it on the CPU (syntax and types) and, under -m gpu, builds it with
 * main() below and runs it: ONE rank on a communicator made with ncclCommInitRank, the gathered block against a second
 * chain_batch of the same frames. */
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
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

/* usage: rccl_host <frames.raw> <nframes> <W> <H> <points_pitch> <gathered.out>
 * frames.raw: nframes x H x W bytes.  A world of ONE rank (the communicator is a real one, made the way a launcher's rank
 * makes it: ncclGetUniqueId on rank 0, ncclCommInitRank everywhere); the gathered block is written to gathered.out and
 * compared here, byte for byte over the live entries, with what a plain chain_batch of the same frames returns. */
int main(int argc, char** argv) {
    int n, W, H, pitch, rc, f, bad = 0;
    size_t o_lv = 0, o_np = 0, bytes = 0, block = 0, frame_bytes;
    uint8_t* h_frames;
    uint8_t* d_frames = NULL;
    char* h_gathered;
    char* h_plain;
    char* d_plain = NULL;
    void* d_gathered = NULL;
    ncclUniqueId id;
    ncclComm_t comm;
    mrgingham_amd_ctx* ctx;
    mrgingham_amd_frames fr;
    FILE* fp;
    if (argc != 7) { fprintf(stderr, "usage: %s frames.raw nframes W H points_pitch gathered.out\n", argv[0]); return 2; }
    n = atoi(argv[2]); W = atoi(argv[3]); H = atoi(argv[4]); pitch = atoi(argv[5]);
    frame_bytes = (size_t)W * H;
    h_frames = (uint8_t*)malloc(frame_bytes * n);
    fp = fopen(argv[1], "rb");
    if (!fp || !h_frames || fread(h_frames, frame_bytes, n, fp) != (size_t)n) { fprintf(stderr, "cannot read %s\n", argv[1]); return 2; }
    fclose(fp);
    if (hipSetDevice(0) != hipSuccess || hipMalloc((void**)&d_frames, frame_bytes * n) != hipSuccess ||
        hipMemcpy(d_frames, h_frames, frame_bytes * n, hipMemcpyHostToDevice) != hipSuccess) return 3;
    if (ncclGetUniqueId(&id) != ncclSuccess || ncclCommInitRank(&comm, 1, id, 0) != ncclSuccess) {
        fprintf(stderr, "no one-rank communicator\n");
        return 4;
    }
    rc = run_rank(0, 1, 0, comm, d_frames, n, W, H, pitch, &d_gathered, &block);
    if (rc != 0) { fprintf(stderr, "run_rank %d\n", rc); return 5; }
    h_gathered = (char*)malloc(block);
    if (hipMemcpy(h_gathered, d_gathered, block, hipMemcpyDeviceToHost) != hipSuccess) return 6;
    /* the same frames through chain_batch alone, same layout */
    if (mrgingham_amd_packed_layout(n, pitch, &o_lv, &o_np, &bytes) != 0 || bytes != block) return 7;
    ctx = mrgingham_amd_create(0);
    if (!ctx || hipMalloc((void**)&d_plain, bytes) != hipSuccess || hipMemset(d_plain, 0, bytes) != hipSuccess) return 8;
    fr.frames = d_frames; fr.frame_pitch = (int64_t)frame_bytes; fr.nframes = n; fr.width = W; fr.height = H; fr.stride = W;
    if (mrgingham_amd_chain_batch(ctx, &fr, 3, (double*)d_plain, (signed char*)(d_plain + o_lv), (int32_t*)(d_plain + o_np),
                                  pitch) != 0 || mrgingham_amd_sync(ctx) != 0) {
        fprintf(stderr, "%s\n", mrgingham_amd_last_error(ctx));
        return 9;
    }
    h_plain = (char*)malloc(bytes);
    if (hipMemcpy(h_plain, d_plain, bytes, hipMemcpyDeviceToHost) != hipSuccess) return 10;
    for (f = 0; f < n; f++) {
        int32_t np_g, np_p;
        memcpy(&np_g, h_gathered + o_np + 4 * (size_t)f, 4);
        memcpy(&np_p, h_plain + o_np + 4 * (size_t)f, 4);
        if (np_g != np_p || np_g < 0 || np_g > pitch) { bad++; continue; }
        if (memcmp(h_gathered + (size_t)f * pitch * 16, h_plain + (size_t)f * pitch * 16, (size_t)np_g * 16) != 0) bad++;
        if (memcmp(h_gathered + o_lv + (size_t)f * pitch, h_plain + o_lv + (size_t)f * pitch, (size_t)np_g) != 0) bad++;
        printf("frame %d points %d\n", f, (int)np_g);
    }
    fp = fopen(argv[6], "wb");
    if (!fp || fwrite(h_gathered, 1, block, fp) != block) return 11;
    fclose(fp);
    printf("block_bytes %lu off_levels %lu off_npoints %lu\n", (unsigned long)block, (unsigned long)o_lv, (unsigned long)o_np);
    printf("gathered_equals_chain %d\n", bad == 0);
    ncclCommDestroy(comm);
    mrgingham_amd_destroy(ctx);
    hipFree(d_plain); hipFree(d_gathered); hipFree(d_frames);
    free(h_plain); free(h_gathered); free(h_frames);
    return bad == 0 ? 0 : 1;
}
