// mrgingham-amd-from-image: the reference's command-line tool (mrgingham-from-image.cc) over
// libmrgingham_amd.so.  Same options, same vnlog output ("# filename x y level", one record per
// corner, "filename - - -" when no board is found), same worker model (--jobs N threads, image i
// goes to worker i % N, mrgingham-from-image.cc:50), same failure behaviour (an unreadable image is
// reported and ends that worker, :58-68).  Row (f)-3 of the scope table: I/O around the hot path.
//
// Image decoding: the reference uses cv::imread; OpenCV is not available to this build, so this
// file reads binary PGM (P5, 8 or 16 bit) and non-interlaced PNG (8 or 16 bit; grey, grey+alpha,
// RGB, RGBA, 8-bit palette) with zlib.  Colour is reduced to grey with the fixed-point BT.601
// weights (4899 R + 9617 G + 1868 B + 8192) >> 14 -- OpenCV's own conversion depends on its
// codec build, so byte-identity with cv::imread on colour files is not claimed.
//
// Preprocessing (normalize + CLAHE(8), box blur) and detection run on the GPU through
// mrgingham_amd_process_image_ex, for 8- and 16-bit images alike (16 bit: normalize to 0..65535, CLAHE on
// 16 bits, convertTo(CV_8U, 255/65535), :85-92).
#include <getopt.h>
#include <glob.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <chrono>
#include <cmath>
#include <string>
#include <vector>

#include "mrgingham_amd.h"
#include "../csrc/image_io.h"

namespace {

using mrg::Image;
using mrg::read_image;

struct Options {
    glob_t globbed;
    int jobs = 1, blur_radius = 1, gridn = 10, level = -1;
    int gpus = 0;  // --gpus: 0 = not given (the library's default: worker k on device k % devices, or MRGINGHAM_AMD_DEVICE)
    bool doclahe = true, do_refine = true, debug = false, doblobs = false;
    int debug_sequence_x = -1, debug_sequence_y = -1;
} opt;

const char* kUsage =
    "Usage: %s [--gridn N] [--noclahe] [--blur radius] [--level l] [--no-refine] [--jobs N]\n"
    "          [--gpus N|all] [--debug] [--debug-sequence x,y] imageglobs...\n"
    "\n"
    "Finds the chessboard in every image (binary PGM or PNG) and writes a vnlog table\n"
    "\n"
    "  # filename x y level\n"
    "\n"
    "with one record per corner, or 'filename - - -' when no board was found.  The corner\n"
    "candidates, the contrast preprocessing and the refinement run on an AMD GPU\n"
    "(libmrgingham_amd); the options mean what they mean in mrgingham-from-image:\n"
    "\n"
    "  --gridn N       corners per side of the board (default 10)\n"
    "  --noclahe       skip the normalize + CLAHE(8) step\n"
    "  --blur R        box blur radius applied before detection (default 1, 0 = none)\n"
    "  --level L       pyramid level to search at; default -1 = try 3, 2, 1, 0 in turn\n"
    "  --no-refine     keep the corners of the level the board was found at\n"
    "  --jobs N, -j N  worker threads (image i is handled by worker i mod N)\n"
    "  --gpus N|all    GPUs to use: worker k works on device k mod N (default: every GPU of the node,\n"
    "                  or the one named by MRGINGHAM_AMD_DEVICE); --jobs several times the GPU count keeps\n"
    "                  each GPU busy while other workers read and decode\n"
    "  --blobs         find a grid of dark circles instead of a chessboard (no --level, no refinement)\n"
    "  --debug         one image only: write the preprocessed image, the level images, the ChESS\n"
    "                  responses and the corner vnlogs to /tmp like the reference does\n"
    "  --debug-sequence x,y   accepted (the grid finder's own dumps are not produced)\n";

// At most kDeviceSlots threads per GPU call into the library.  With --jobs up to that many per GPU the workers call it
// themselves; with more, the workers only read, decode and print, and hand their images to kDeviceSlots DEVICE THREADS
// per GPU, asleep on a condition variable while they wait.  More callers per GPU add no throughput (the device is one),
// every caller costs a context (streams, scratch: ~20 ms of start-up each), and every waiting caller spins for the device:
// once they outnumber the cores the spinners burn the time slices of the threads that feed the device (4096 small images
// on 16 cores: 1.4 s with 4 workers, 2-4 s with 16, 10-13 s with 32 before this; decoding PNGs is what more workers are
// for).
constexpr int kDeviceSlots = 8;
struct Request {
    const void* px;
    int depth, w, h;
    const mrgingham_amd_cli_options* o;
    double* xy;
    signed char* lv;
    int level;
    bool done;
    pthread_cond_t cv;
};
struct DeviceQueue {
    pthread_mutex_t mu = PTHREAD_MUTEX_INITIALIZER;
    pthread_cond_t work = PTHREAD_COND_INITIALIZER;
    std::vector<Request*> q;  // (a handful of entries: at most one per worker)
    bool stop = false;
    int device = -1;  // -1: wherever the library puts the thread (MRGINGHAM_AMD_DEVICE)
};
std::vector<DeviceQueue> device_queues;  // one per GPU in use (main)
bool hand_off = false;                   // more workers than device slots

int process(const void* px, int depth, int w, int h, const mrgingham_amd_cli_options* o, double* xy, signed char* lv) {
    return mrgingham_amd_process_image_ex(px, depth, w, h, w, o, xy, lv);
}
void* device_thread(void* arg) {
    DeviceQueue& Q = *(DeviceQueue*)arg;
    if (Q.device >= 0) mrgingham_amd_set_thread_device(Q.device);
    pthread_mutex_lock(&Q.mu);
    while (true) {
        while (Q.q.empty() && !Q.stop) pthread_cond_wait(&Q.work, &Q.mu);
        if (Q.q.empty()) break;
        Request* r = Q.q.front();
        Q.q.erase(Q.q.begin());
        pthread_mutex_unlock(&Q.mu);
        const int level = process(r->px, r->depth, r->w, r->h, r->o, r->xy, r->lv);
        pthread_mutex_lock(&Q.mu);
        r->level = level;
        r->done = true;
        pthread_cond_signal(&r->cv);
    }
    pthread_mutex_unlock(&Q.mu);
    return nullptr;
}

void* worker(void* arg) {
    const int ijob = (int)(intptr_t)arg;
    const int N = opt.gridn * opt.gridn;
    std::vector<double> xy((size_t)N * 2);
    std::vector<signed char> lv((size_t)N);
    Image im;  // reused: its buffers keep their pages from image to image
    // worker k works on device k mod N (N = --gpus, or every GPU of the node; MRGINGHAM_AMD_DEVICE pins all to one)
    DeviceQueue& Q = device_queues[(size_t)ijob % device_queues.size()];
    if (!hand_off && Q.device >= 0) mrgingham_amd_set_thread_device(Q.device);
    Request req{};
    if (hand_off) pthread_cond_init(&req.cv, nullptr);
    // the decoded pixels are page-locked where they lie, so that the upload runs at the speed of the link (re-done
    // when an image of another size moves the buffer)
    void* locked = nullptr;
    auto lock_pixels = [&](void* p, size_t bytes) {
        if (p == locked) return;
        if (locked) mrgingham_amd_host_unregister(locked);
        locked = mrgingham_amd_host_register(p, bytes) == 0 ? p : nullptr;
    };
    // a larger image is about to move the pixel buffer: the registration goes first (registered memory must stay allocated
    // until it is unregistered, include/mrgingham_amd.h), then the buffer grows with nothing locked in it
    im.before_grow = [&](size_t n8, size_t n16) {
        if (locked) mrgingham_amd_host_unregister(locked);
        locked = nullptr;
        if (n8 > im.px8.capacity()) im.px8.reserve(n8 + n8 / 8);
        if (n16 > im.px16.capacity()) im.px16.reserve(n16 + n16 / 8);
    };
    // MRGINGHAM_AMD_CLI_TIMING=1: where a worker's time goes (stderr, at its end)
    static const bool timing = getenv("MRGINGHAM_AMD_CLI_TIMING") != nullptr;
    auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t_read = 0, t_proc = 0, t_out = 0;
    int nimg = 0;
    for (int i = ijob; i < (int)opt.globbed.gl_pathc; i += opt.jobs) {
        const char* filename = opt.globbed.gl_pathv[i];
        const double t0 = now();
        if (!read_image(filename, im)) {  // mrgingham-from-image.cc:58-68
            fprintf(stderr, "Couldn't open image '%s'\n", filename);
            flockfile(stdout);
            printf("## Couldn't open image '%s'\n", filename);
            printf("%s - - -\n", filename);
            funlockfile(stdout);
            break;
        }
        mrgingham_amd_cli_options o{};
        o.do_clahe = opt.doclahe;
        o.blur_radius = opt.blur_radius;
        o.gridn = opt.gridn;
        o.image_pyramid_level = opt.level;
        o.do_refine = opt.do_refine;
        o.do_blobs = opt.doblobs;
        o.debug = opt.debug;
        o.debug_sequence_x = opt.debug_sequence_x;
        o.debug_sequence_y = opt.debug_sequence_y;
        o.filename = filename;
        if (im.depth == 16) lock_pixels(im.px16.data(), im.px16.capacity() * 2);
        else lock_pixels(im.px8.data(), im.px8.capacity());
        const double t1 = now();
        // 8- and 16-bit images go to the device as they are (mrgingham-from-image.cc:71-92)
        const void* px = im.depth == 16 ? (const void*)im.px16.data() : (const void*)im.px8.data();
        int level;
        if (!hand_off) {
            level = process(px, im.depth == 16 ? 16 : 8, im.w, im.h, &o, xy.data(), lv.data());
        } else {
            req.px = px; req.depth = im.depth == 16 ? 16 : 8; req.w = im.w; req.h = im.h;
            req.o = &o; req.xy = xy.data(); req.lv = lv.data(); req.done = false;
            pthread_mutex_lock(&Q.mu);
            Q.q.push_back(&req);
            pthread_cond_signal(&Q.work);
            while (!req.done) pthread_cond_wait(&req.cv, &Q.mu);
            pthread_mutex_unlock(&Q.mu);
            level = req.level;
        }
        const double t2 = now();
        flockfile(stdout);
        if (level >= 0)  // mrgingham-from-image.cc:174-183
            for (int k = 0; k < N; ++k)
                printf("%s %f %f %d\n", filename, xy[2 * k], xy[2 * k + 1], (opt.do_refine && !opt.doblobs) ? (int)lv[k] : level);
        else
            printf("%s - - -\n", filename);
        funlockfile(stdout);
        t_read += t1 - t0; t_proc += t2 - t1; t_out += now() - t2; ++nimg;
    }
    if (timing && nimg)
        fprintf(stderr, "worker %d: %d images; per image: read + decode %.3f ms, upload + device + grid finder %.3f ms, output %.3f ms\n",
                ijob, nimg, t_read / nimg, t_proc / nimg, t_out / nimg);
    if (locked) mrgingham_amd_host_unregister(locked);
    if (hand_off) pthread_cond_destroy(&req.cv);
    return nullptr;
}

}  // namespace

int main(int argc, char* argv[]) {
    static const struct option longopts[] = {
        {"blobs", no_argument, nullptr, 'B'},          {"blur", required_argument, nullptr, 'b'},
        {"noclahe", no_argument, nullptr, 'C'},        {"level", required_argument, nullptr, 'l'},
        {"no-refine", no_argument, nullptr, 'R'},      {"jobs", required_argument, nullptr, 'j'},
        {"gridn", required_argument, nullptr, 'N'},    {"debug", no_argument, nullptr, 'd'},
        {"debug-sequence", required_argument, nullptr, 'D'}, {"help", no_argument, nullptr, 'h'},
        {"gpus", required_argument, nullptr, 'G'},
        {nullptr, 0, nullptr, 0}};
    bool doblobs = false;
    int c;
    while ((c = getopt_long(argc, argv, "hj:b:l:", longopts, nullptr)) != -1) {
        switch (c) {
            case 'h': printf(kUsage, argv[0]); return 0;
            case 'B': doblobs = true; break;
            case 'C': opt.doclahe = false; break;
            case 'R': opt.do_refine = false; break;
            case 'd': opt.debug = true; break;
            case 'D': {
                int x, y;
                if (sscanf(optarg, "%d,%d", &x, &y) != 2) {
                    fprintf(stderr, "I could not parse 'x,y' from --debug-sequence '%s'. Giving up\n", optarg);
                    fprintf(stderr, kUsage, argv[0]);
                    return -1;
                }
                opt.debug_sequence_x = x;
                opt.debug_sequence_y = y;
                break;
            }
            case 'N': opt.gridn = atoi(optarg); break;
            case 'b': opt.blur_radius = atoi(optarg); break;
            case 'l': opt.level = atoi(optarg); break;
            case 'j': opt.jobs = atoi(optarg); break;
            case 'G': opt.gpus = !strcmp(optarg, "all") ? -1 : atoi(optarg); if (opt.gpus == 0) opt.gpus = -2; break;
            default:
                fprintf(stderr, "Unknown option\n");
                fprintf(stderr, kUsage, argv[0]);
                return 1;
        }
    }
    if (optind > argc - 1) {
        fprintf(stderr, "Not enough arguments: need image globs\n");
        fprintf(stderr, kUsage, argv[0]);
        return 1;
    }
    if (opt.jobs <= 0) {
        fprintf(stderr, "The job count must be a positive integer\n");
        fprintf(stderr, kUsage, argv[0]);
        return 1;
    }
    if (doblobs && opt.level >= 0) {  // mrgingham-from-image.cc:305-309
        fprintf(stderr, "ERROR: 'image_pyramid_level' only implemented for chessboards.\n");
        return 1;
    }
    opt.doblobs = doblobs;
    if (doblobs) opt.level = 0;
    if (opt.gridn < 2) {
        fprintf(stderr, "--gridn value must be >= 2\n");
        return 1;
    }
    if (opt.blur_radius < 0) {
        fprintf(stderr, "--blur value must be >= 0\n");
        return 1;
    }
    int append = 0;
    for (int i = optind; i < argc; ++i) {
        const int r = glob(argv[i], append | GLOB_ERR | GLOB_MARK | GLOB_NOSORT | GLOB_TILDE_CHECK, nullptr, &opt.globbed);
        if (r == GLOB_NOMATCH) { fprintf(stderr, "'%s' matched no files!\n", argv[i]); return 1; }
        if (r != 0) { fprintf(stderr, "globbing '%s' failed!\n", argv[i]); return 1; }
        append = GLOB_APPEND;
    }
    if (opt.debug && opt.globbed.gl_pathc != 1) {
        fprintf(stderr, "When debugging, pass one image at a time. Got %d instead\n", (int)opt.globbed.gl_pathc);
        return 1;
    }
    // every worker's context overlaps three HIP streams; HIP maps streams onto GPU_MAX_HW_QUEUES
    // hardware queues (default 4) and streams that share one serialise (must be set before HIP starts)
    setenv("GPU_MAX_HW_QUEUES", "8", 0);
    const int ndev = mrgingham_amd_device_count();
    if (ndev <= 0) {
        fprintf(stderr, "mrgingham-amd-from-image: no HIP device: this tool has no CPU path\n");
        return 2;
    }
    if (opt.gpus == -2 || opt.gpus < -2) {
        fprintf(stderr, "--gpus takes a positive count or 'all'\n");
        return 1;
    }
    if (opt.gpus == -1) opt.gpus = ndev;
    if (opt.gpus > ndev) {
        fprintf(stderr, "mrgingham-amd-from-image: --gpus %d, but this node has %d: using %d\n", opt.gpus, ndev, ndev);
        opt.gpus = ndev;
    }
    printf("## generated with");
    for (int i = 0; i < argc; ++i) printf(" %s", argv[i]);
    printf("\n# filename x y level\n");
    fflush(stdout);
    const bool pinned_by_env = getenv("MRGINGHAM_AMD_DEVICE") != nullptr && opt.gpus <= 0;
    const int ng = opt.gpus > 0 ? opt.gpus : (pinned_by_env ? 1 : ndev);
    device_queues = std::vector<DeviceQueue>((size_t)ng);
    for (int d = 0; d < ng; ++d) device_queues[(size_t)d].device = pinned_by_env ? -1 : d;
    hand_off = opt.jobs > kDeviceSlots * ng;
    std::vector<pthread_t> dth;
    if (hand_off) {
        dth.resize((size_t)kDeviceSlots * ng);
        for (size_t i = 0; i < dth.size(); ++i) pthread_create(&dth[i], nullptr, device_thread, &device_queues[i % (size_t)ng]);
    }
    std::vector<pthread_t> th((size_t)opt.jobs);
    for (int i = 0; i < opt.jobs; ++i) pthread_create(&th[i], nullptr, worker, (void*)(intptr_t)i);
    for (int i = 0; i < opt.jobs; ++i) pthread_join(th[i], nullptr);
    for (DeviceQueue& Q : device_queues) {
        pthread_mutex_lock(&Q.mu);
        Q.stop = true;
        pthread_cond_broadcast(&Q.work);
        pthread_mutex_unlock(&Q.mu);
    }
    for (pthread_t& t : dth) pthread_join(t, nullptr);
    globfree(&opt.globbed);
    return 0;
}
