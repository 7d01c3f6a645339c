// See image_io.h.
#include "image_io.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

#include <cmath>

namespace mrg {

// Largest side the library accepts anywhere (int16 coordinates, find_chessboard_corners.cc:91): checked
// BEFORE anything is allocated or indexed from a header field of an untrusted file.
constexpr int kMaxSide = 32767;

static bool read_file(const char* path, std::vector<uint8_t>& buf) {
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

static void about_to_hold(Image& im, size_t n8, size_t n16) {
    if (im.before_grow && (n8 > im.px8.capacity() || n16 > im.px16.capacity())) im.before_grow(n8, n16);
}

static bool decode_pgm(const std::vector<uint8_t>& b, Image& im) {
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
    if (w <= 0 || h <= 0 || w > kMaxSide || h > kMaxSide || maxval <= 0 || maxval > 65535) return false;
    const size_t n = (size_t)w * h;
    im.w = w; im.h = h;
    if (maxval < 256) {
        if (b.size() - p < n) return false;
        im.depth = 8;
        about_to_hold(im, n, 0);
        im.px8.assign(b.begin() + p, b.begin() + p + n);
    } else {
        if (b.size() - p < 2 * n) return false;
        im.depth = 16;
        about_to_hold(im, 0, n);
        im.px16.resize(n);
        for (size_t i = 0; i < n; ++i) im.px16[i] = (uint16_t)((b[p + 2 * i] << 8) | b[p + 2 * i + 1]);
    }
    return true;
}

static inline uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | (p[1] << 16) | (p[2] << 8) | p[3]; }

static bool decode_png(const std::vector<uint8_t>& b, Image& im) {
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    if (b.size() < 8 + 25 || memcmp(b.data(), sig, 8)) return false;
    size_t p = 8;
    int w = 0, h = 0, bits = 0, ctype = -1, interlace = 0;
    bool have_ihdr = false;
    std::vector<uint8_t> idat, plte;
    while (p + 12 <= b.size()) {
        const uint32_t len = be32(&b[p]);
        const char* type = (const char*)&b[p + 4];
        if (p + 12 + (size_t)len > b.size()) return false;
        const uint8_t* d = &b[p + 8];
        if (!memcmp(type, "IHDR", 4)) {
            // exactly one IHDR, first, 13 bytes; sides within the library's limit (a crafted header must
            // not size the buffers below)
            if (have_ihdr || p != 8 || len != 13) return false;
            const uint32_t uw = be32(d), uh = be32(d + 4);
            if (uw == 0 || uh == 0 || uw > (uint32_t)kMaxSide || uh > (uint32_t)kMaxSide) return false;
            w = (int)uw; h = (int)uh; bits = d[8]; ctype = d[9]; interlace = d[12];
            have_ihdr = true;
        } else if (!have_ihdr) return false;  // any other chunk before IHDR
        else if (!memcmp(type, "PLTE", 4)) plte.assign(d, d + len);
        else if (!memcmp(type, "IDAT", 4)) idat.insert(idat.end(), d, d + len);
        else if (!memcmp(type, "IEND", 4)) break;
        p += 12 + (size_t)len;
    }
    if (!have_ihdr || idat.empty() || interlace != 0 || (bits != 8 && bits != 16)) return false;
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
        about_to_hold(im, n, 0);
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
        about_to_hold(im, 0, n);
        im.px16.resize(n);
        for (size_t i = 0; i < n; ++i) {
            const uint8_t* q = &img[i * bpp];
            auto s16 = [&](int k) { return (uint32_t)((q[2 * k] << 8) | q[2 * k + 1]); };
            im.px16[i] = (uint16_t)((ctype == 0 || ctype == 4) ? s16(0) : grey(s16(0), s16(1), s16(2)));
        }
    }
    return true;
}

// 8-bit binary PGM, the format a calibration run usually feeds the tool: the pixels go from the file straight into
// px8 -- no copy of the whole file in between (a 12 MB image: one pass over memory less per image).  Returns 1 = done,
// 0 = not such a file (the general path decides), -1 = such a file, but broken.
static int read_pgm8_direct(const char* path, Image& im) {
    FILE* f = fopen(path, "rb");
    if (!f) return -1;
    uint8_t head[128];
    const size_t got = fread(head, 1, sizeof(head), f);
    int rc = 0;
    if (got >= 8 && head[0] == 'P' && head[1] == '5') {
        size_t p = 2;
        int v[3] = {0, 0, 0};
        bool ok = true;
        for (int k = 0; k < 3 && ok; ++k) {
            for (;;) {
                while (p < got && (head[p] == ' ' || head[p] == '\t' || head[p] == '\n' || head[p] == '\r')) ++p;
                if (p < got && head[p] == '#') { while (p < got && head[p] != '\n') ++p; continue; }
                break;
            }
            if (p >= got || head[p] < '0' || head[p] > '9') { ok = false; break; }
            long x = 0;
            while (p < got && head[p] >= '0' && head[p] <= '9') { x = x * 10 + (head[p] - '0'); if (x > 1 << 30) { ok = false; break; } ++p; }
            v[k] = (int)x;
        }
        // (a header with a long comment does not fit the 128 bytes: the general path takes it)
        if (ok && p < got && v[2] > 0 && v[2] < 256 && v[0] > 0 && v[1] > 0 && v[0] <= kMaxSide && v[1] <= kMaxSide) {
            ++p;  // the single whitespace after maxval
            const size_t n = (size_t)v[0] * v[1];
            im.w = v[0]; im.h = v[1]; im.depth = 8;
            about_to_hold(im, n, 0);
            im.px8.resize(n);
            const size_t have = got - p < n ? got - p : n;
            memcpy(im.px8.data(), head + p, have);
            rc = fread(im.px8.data() + have, 1, n - have, f) == n - have ? 1 : -1;
        }
    }
    fclose(f);
    return rc;
}

bool read_image(const char* path, Image& im) {
    // never throws: the callers are extern "C" entry points and detached worker threads
    try {
        const int direct = read_pgm8_direct(path, im);
        if (direct != 0) return direct > 0;
        std::vector<uint8_t>& b = im.file;
        if (!read_file(path, b) || b.size() < 8) return false;
        if (b[0] == 'P' && b[1] == '5') return decode_pgm(b, im);
        if (b[0] == 0x89 && b[1] == 'P') return decode_png(b, im);
        return false;
    } catch (...) {  // std::bad_alloc / length_error on a file that claims more than can be held
        return false;
    }
}

bool write_png_gray8(const char* path, const uint8_t* px, int w, int h) {
    if (!path || !px || w <= 0 || h <= 0) return false;
    try {
        std::vector<uint8_t> raw((size_t)(w + 1) * h);
        for (int y = 0; y < h; ++y) {
            raw[(size_t)(w + 1) * y] = 0;  // filter type "none"
            memcpy(&raw[(size_t)(w + 1) * y + 1], px + (size_t)w * y, (size_t)w);
        }
        uLongf clen = compressBound((uLong)raw.size());
        std::vector<uint8_t> comp(clen);
        if (compress2(comp.data(), &clen, raw.data(), (uLong)raw.size(), 3) != Z_OK) return false;
        FILE* f = fopen(path, "wb");
        if (!f) return false;
        auto chunk = [&](const char* type, const uint8_t* d, uint32_t n) {
            uint8_t hdr[8] = {(uint8_t)(n >> 24), (uint8_t)(n >> 16), (uint8_t)(n >> 8), (uint8_t)n,
                              (uint8_t)type[0], (uint8_t)type[1], (uint8_t)type[2], (uint8_t)type[3]};
            uLong crc = crc32(0L, hdr + 4, 4);
            if (n) crc = crc32(crc, d, n);
            const uint8_t tail[4] = {(uint8_t)(crc >> 24), (uint8_t)(crc >> 16), (uint8_t)(crc >> 8), (uint8_t)crc};
            fwrite(hdr, 1, 8, f);
            if (n) fwrite(d, 1, n, f);
            fwrite(tail, 1, 4, f);
        };
        static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
        fwrite(sig, 1, 8, f);
        const uint8_t ihdr[13] = {(uint8_t)(w >> 24), (uint8_t)(w >> 16), (uint8_t)(w >> 8), (uint8_t)w,
                                  (uint8_t)(h >> 24), (uint8_t)(h >> 16), (uint8_t)(h >> 8), (uint8_t)h, 8, 0, 0, 0, 0};
        chunk("IHDR", ihdr, 13);
        chunk("IDAT", comp.data(), (uint32_t)clen);
        chunk("IEND", nullptr, 0);
        const bool ok = !ferror(f);
        fclose(f);
        return ok;
    } catch (...) {
        return false;
    }
}

void to_8bit_imread(const Image& im, std::vector<uint8_t>& out) {
    out.resize(im.px16.size());
    for (size_t k = 0; k < out.size(); ++k) out[k] = (uint8_t)(im.px16[k] >> 8);
}

void to_8bit(const Image& im, std::vector<uint8_t>& out) {
    out.resize(im.px16.size());
    for (size_t k = 0; k < out.size(); ++k) {
        const long r = lrint((double)im.px16[k] * (255. / 65535.));
        out[k] = (uint8_t)(r < 0 ? 0 : r > 255 ? 255 : r);
    }
}

}  // namespace mrg
