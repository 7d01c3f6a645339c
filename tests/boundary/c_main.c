/* BOUNDARY test: a plain C99 program that knows nothing but include/mrgingham_amd.h -- what a C caller of the
 * reference's symbols (ChESS.h:31-34, mrgingham_pywrap_cplusplus_bridge.h:10-42) links against.  No arguments: only
 * takes the address of every declared function (the header is valid C and the library exports what it declares).
 * With a PGM file: runs the reference's three symbols and the preprocessing entry on it and prints what came back. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "mrgingham_amd.h"

struct ints { int* xy; int n; double scale; };
static bool take_ints(int* xy, int N, double scale, void* cookie) {
    struct ints* c = (struct ints*)cookie;
    c->xy = (int*)malloc(sizeof(int) * 2 * (size_t)(N > 0 ? N : 1));
    if (!c->xy) return false;
    memcpy(c->xy, xy, sizeof(int) * 2 * (size_t)N);
    c->n = N;
    c->scale = scale;
    return true;
}
struct doubles { double* xy; int n; };
static bool take_doubles(double* xy, int N, void* cookie) {
    struct doubles* c = (struct doubles*)cookie;
    c->xy = (double*)malloc(sizeof(double) * 2 * (size_t)(N > 0 ? N : 1));
    if (!c->xy) return false;
    memcpy(c->xy, xy, sizeof(double) * 2 * (size_t)N);
    c->n = N;
    return true;
}

int main(int argc, char** argv) {
    /* every function the header declares, by address (link check) */
    typedef void (*any_fn)(void);
    const any_fn syms[] = {
#define MRG_SYMBOL(name) (any_fn)name,
#include "all_symbols.inc" /* written by tests/test_c_client.py: one MRG_SYMBOL(...) per entry of mrgingham_amd/_lib.py EXPORTS */
#undef MRG_SYMBOL
    };
    size_t i, nsyms = sizeof(syms) / sizeof(syms[0]);
    for (i = 0; i < nsyms; ++i)
        if (!syms[i]) return 3;
    if (argc < 2) {
        printf("symbols %d abi %d\n", (int)nsyms, mrgingham_amd_abi_version());
        return 0;
    }
    {
        int w = 0, h = 0, depth = 0, k;
        uint8_t* img;
        int16_t* resp;
        struct ints cand = {0, 0, 0};
        struct doubles board = {0, 0};
        bool found;
        if (mrgingham_amd_read_image(argv[1], 0, NULL, 0, &w, &h, &depth) != 0 || w <= 0 || h <= 0 || depth != 8) return 4;
        img = (uint8_t*)mrgingham_amd_host_alloc((size_t)w * h);      /* page-locked: the wrappers copy straight out of it */
        resp = (int16_t*)calloc((size_t)w * h, sizeof(int16_t));
        if (!img || !resp || mrgingham_amd_read_image(argv[1], 0, img, (size_t)w * h, &w, &h, &depth) != 0) return 5;
        printf("image %d %d\n", w, h);
        mrgingham_ChESS_response_5(resp, img, w, h, w);
        {
            long long sum = 0;
            long long n = (long long)w * h, j;
            for (j = 0; j < n; ++j) sum += (long long)resp[j] * (1 + j % 7);
            printf("response_checksum %lld\n", sum);
        }
        found = find_chessboard_corners_from_image_array_C(h, w, w, (char*)img, 1, false, false, take_ints, &cand);
        printf("corners found %d n %d scale %.6f\n", (int)found, cand.n, cand.scale);
        for (k = 0; k < cand.n; ++k) printf("p %d %d\n", cand.xy[2 * k], cand.xy[2 * k + 1]);
        found = find_chessboard_from_image_array_C(h, w, w, (char*)img, 10, -1, false, false, -1, -1, take_doubles, &board);
        printf("board found %d n %d\n", (int)found, board.n);
        for (k = 0; k < board.n; ++k) printf("b %.17g %.17g\n", board.xy[2 * k], board.xy[2 * k + 1]);
        /* an unreasonable level: false, nothing delivered (find_chessboard_corners.cc:433-441) */
        {
            struct ints none = {0, -1, 0};
            const bool bad = find_chessboard_corners_from_image_array_C(h, w, w, (char*)img, 11, false, false, take_ints, &none);
            printf("bad_level %d %d\n", (int)bad, none.n);
        }
        mrgingham_amd_host_free(img);
        free(resp);
        free(cand.xy);
        free(board.xy);
    }
    return 0;
}
