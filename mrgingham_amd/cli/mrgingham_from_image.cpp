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
// mrgingham_amd_process_image; 16-bit input is reduced to 8 bit on the host the way the reference
// does (convertTo(CV_8U, 255/65535), :91) and is only accepted together with --noclahe.
#include <getopt.h>
#include <glob.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

#include <cmath>
#include <string>
#include <vector>

#include "mrgingham_amd.h"

namespace {

struct Image {
    int w = 0, h = 0, depth = 0;  // depth 8 or 16
    std::vector<uint8_t> px8;
    std::vector<uint16_t> px16;
};

bool read_file(const char* path, std::vector<uint8_t>& buf) {
    FILE* f = fopen(path, "rb");
    if (!f) return false;
    fseek(f, 0, SEEK_END);
    const long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    if (n <= 0) { fclose(f); return false; }
    buf.resize((size_t)n);
    const bool ok = fread(buf.data(), 1, (size_t)n, f) == (size_t)n;
    fclose(f);
    return ok;
}

bool decode_pgm(const std::vector<uint8_t>& b, Image& im) {
    size_t p = 2;
    auto next_int = [&](int& v) {
        for (;;) {
            while (p < b.size() && (b[p] == ' ' || b[p] == '\t' || b[p] == '\n' || b[p] == '\r')) ++p;
            if (p < b.size() && b[p] == '#') { while (p < b.size() && b[p] != '\n') ++p; continue; }
            break;
        }
        if (p >= b.size() || b[p] < '0' || b[p] > '9') return false;
        long x = 0;
        while (p < b.size() && b[p] >= '0' && b[p] <= '9') { x = x * 10 + (b[p] - '0'); if (x > 1 << 30) return false; ++p; }
        v = (int)x;
        return true;
    };
    int w, h, maxval;
    if (!next_int(w) || !next_int(h) || !next_int(maxval)) return false;
    if (p >= b.size()) return false;
    ++p;  // the single whitespace after maxval
    if (w <= 0 || h <= 0 || maxval <= 0 || maxval > 65535) return false;
    const size_t n = (size_t)w * h;
    im.w = w; im.h = h;
    if (maxval < 256) {
        if (b.size() - p < n) return false;
        im.depth = 8;
        im.px8.assign(b.begin() + p, b.begin() + p + n);
    } else {
        if (b.size() - p < 2 * n) return false;
        im.depth = 16;
        im.px16.resize(n);
        for (size_t i = 0; i < n; ++i) im.px16[i] = (uint16_t)((b[p + 2 * i] << 8) | b[p + 2 * i + 1]);
    }
    return true;
}

inline uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | (p[1] << 16) | (p[2] << 8) | p[3]; }

bool decode_png(const std::vector<uint8_t>& b, Image& im) {
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    if (b.size() < 8 + 25 || memcmp(b.data(), sig, 8)) return false;
    size_t p = 8;
    int w = 0, h = 0, bits = 0, ctype = -1, interlace = 0;
    std::vector<uint8_t> idat, plte;
    while (p + 12 <= b.size()) {
        const uint32_t len = be32(&b[p]);
        const char* type = (const char*)&b[p + 4];
        if (p + 12 + (size_t)len > b.size()) return false;
        const uint8_t* d = &b[p + 8];
        if (!memcmp(type, "IHDR", 4) && len >= 13) {
            w = (int)be32(d); h = (int)be32(d + 4); bits = d[8]; ctype = d[9]; interlace = d[12];
        } else if (!memcmp(type, "PLTE", 4)) plte.assign(d, d + len);
        else if (!memcmp(type, "IDAT", 4)) idat.insert(idat.end(), d, d + len);
        else if (!memcmp(type, "IEND", 4)) break;
        p += 12 + (size_t)len;
    }
    if (w <= 0 || h <= 0 || interlace != 0 || (bits != 8 && bits != 16)) return false;
    int ch;
    switch (ctype) {
        case 0: ch = 1; break;
        case 2: ch = 3; break;
        case 3: ch = 1; if (bits != 8) return false; break;
        case 4: ch = 2; break;
        case 6: ch = 4; break;
        default: return false;
    }
    const size_t bpp = (size_t)ch * bits / 8, rowb = (size_t)w * bpp;
    std::vector<uint8_t> raw((rowb + 1) * (size_t)h);
    uLongf rawlen = (uLongf)raw.size();
    if (uncompress(raw.data(), &rawlen, idat.data(), (uLong)idat.size()) != Z_OK || rawlen != raw.size()) return false;
    std::vector<uint8_t> img(rowb * (size_t)h);
    for (int y = 0; y < h; ++y) {
        const uint8_t ft = raw[(rowb + 1) * y];
        const uint8_t* s = &raw[(rowb + 1) * y + 1];
        uint8_t* o = &img[rowb * y];
        const uint8_t* up = y ? o - rowb : nullptr;
        for (size_t i = 0; i < rowb; ++i) {
            const int a = i >= bpp ? o[i - bpp] : 0, bb = up ? up[i] : 0, c = (up && i >= bpp) ? up[i - bpp] : 0;
            int pred = 0;
            switch (ft) {
                case 0: pred = 0; break;
                case 1: pred = a; break;
                case 2: pred = bb; break;
                case 3: pred = (a + bb) >> 1; break;
                case 4: {
                    const int pa = abs(bb - c), pb = abs(a - c), pc = abs(a + bb - 2 * c);
                    pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? bb : c);
                    break;
                }
                default: return false;
            }
            o[i] = (uint8_t)(s[i] + pred);
        }
    }
    const size_t n = (size_t)w * h;
    im.w = w; im.h = h; im.depth = bits;
    auto grey = [](uint32_t r, uint32_t g, uint32_t bl) { return (r * 4899u + g * 9617u + bl * 1868u + 8192u) >> 14; };
    if (bits == 8) {
        im.px8.resize(n);
        for (size_t i = 0; i < n; ++i) {
            const uint8_t* q = &img[i * bpp];
            if (ctype == 0 || ctype == 4) im.px8[i] = q[0];
            else if (ctype == 3) {
                if ((size_t)q[0] * 3 + 2 >= plte.size()) return false;
                im.px8[i] = (uint8_t)grey(plte[q[0] * 3], plte[q[0] * 3 + 1], plte[q[0] * 3 + 2]);
            } else im.px8[i] = (uint8_t)grey(q[0], q[1], q[2]);
        }
    } else {
        im.px16.resize(n);
        for (size_t i = 0; i < n; ++i) {
            const uint8_t* q = &img[i * bpp];
            auto s16 = [&](int k) { return (uint32_t)((q[2 * k] << 8) | q[2 * k + 1]); };
            im.px16[i] = (uint16_t)((ctype == 0 || ctype == 4) ? s16(0) : grey(s16(0), s16(1), s16(2)));
        }
    }
    return true;
}

bool read_image(const char* path, Image& im) {
    std::vector<uint8_t> b;
    if (!read_file(path, b) || b.size() < 8) return false;
    if (b[0] == 'P' && b[1] == '5') return decode_pgm(b, im);
    if (b[0] == 0x89 && b[1] == 'P') return decode_png(b, im);
    return false;
}

struct Options {
    glob_t globbed;
    int jobs = 1, blur_radius = 1, gridn = 10, level = -1;
    bool doclahe = true, do_refine = true, debug = false;
} opt;

const char* kUsage =
    "Usage: %s [--gridn N] [--noclahe] [--blur radius] [--level l] [--no-refine] [--jobs N]\n"
    "          [--debug] [--debug-sequence x,y] imageglobs...\n"
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
    "  --blobs         circle grids: not supported by this build\n"
    "  --debug, --debug-sequence x,y   accepted; no intermediate dumps are written\n";

void* worker(void* arg) {
    const int ijob = (int)(intptr_t)arg;
    const int N = opt.gridn * opt.gridn;
    std::vector<double> xy((size_t)N * 2);
    std::vector<signed char> lv((size_t)N);
    std::vector<uint8_t> tmp8;
    for (int i = ijob; i < (int)opt.globbed.gl_pathc; i += opt.jobs) {
        const char* filename = opt.globbed.gl_pathv[i];
        Image im;
        if (!read_image(filename, im)) {  // mrgingham-from-image.cc:58-68
            fprintf(stderr, "Couldn't open image '%s'\n", filename);
            flockfile(stdout);
            printf("## Couldn't open image '%s'\n", filename);
            printf("%s - - -\n", filename);
            funlockfile(stdout);
            break;
        }
        const uint8_t* px = im.px8.data();
        if (im.depth == 16) {
            if (opt.doclahe) {
                fprintf(stderr, "Couldn't process image '%s': 16-bit images are only handled with --noclahe by this build\n", filename);
                flockfile(stdout);
                printf("## Couldn't process image '%s': 16-bit images are only handled with --noclahe by this build\n", filename);
                printf("%s - - -\n", filename);
                funlockfile(stdout);
                break;
            }
            // image0.convertTo(image1, CV_8U, 255./65535.), mrgingham-from-image.cc:91
            tmp8.resize(im.px16.size());
            for (size_t k = 0; k < tmp8.size(); ++k) {
                const long r = lrint((double)im.px16[k] * (255. / 65535.));
                tmp8[k] = (uint8_t)(r < 0 ? 0 : r > 255 ? 255 : r);
            }
            px = tmp8.data();
        }
        const int level = mrgingham_amd_process_image(px, im.w, im.h, im.w, opt.doclahe, opt.blur_radius, opt.gridn,
                                                      opt.level, opt.do_refine, xy.data(), lv.data());
        flockfile(stdout);
        if (level >= 0)  // mrgingham-from-image.cc:174-183
            for (int k = 0; k < N; ++k)
                printf("%s %f %f %d\n", filename, xy[2 * k], xy[2 * k + 1], opt.do_refine ? (int)lv[k] : level);
        else
            printf("%s - - -\n", filename);
        funlockfile(stdout);
    }
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
                break;
            }
            case 'N': opt.gridn = atoi(optarg); break;
            case 'b': opt.blur_radius = atoi(optarg); break;
            case 'l': opt.level = atoi(optarg); break;
            case 'j': opt.jobs = atoi(optarg); break;
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
    if (doblobs) {
        fprintf(stderr, "ERROR: --blobs (circle grids, find_blobs.cc) is not part of this build\n");
        return 1;
    }
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
    if (mrgingham_amd_device_count() <= 0) {
        fprintf(stderr, "mrgingham-amd-from-image: no HIP device: this tool has no CPU path\n");
        return 2;
    }
    printf("## generated with");
    for (int i = 0; i < argc; ++i) printf(" %s", argv[i]);
    printf("\n# filename x y level\n");
    fflush(stdout);
    std::vector<pthread_t> th((size_t)opt.jobs);
    for (int i = 0; i < opt.jobs; ++i) pthread_create(&th[i], nullptr, worker, (void*)(intptr_t)i);
    for (int i = 0; i < opt.jobs; ++i) pthread_join(th[i], nullptr);
    globfree(&opt.globbed);
    return 0;
}
