// Sanitizer driver for the host side of libhsefr (csrc/build.sh with HSEFR_ASAN=1; tests/test_plan_blob_fuzz_cpu.py runs it).
// Reads valid plan blobs (written by the Python lowering), and pushes mutated copies through hsefr_plan_validate and
// hsefr_engine_create: truncations, bit flips, and field mutations of the header / buffer table / op table (offsets, sizes, kinds,
// buffer ids, flags) with boundary values.  Every mutant must come back as HSEFR_OK or as a negative status WITH a message; a crash,
// an AddressSanitizer report or an UndefinedBehaviorSanitizer report fails the run (both abort).  No GPU is needed or touched: a
// blob that passes validation stops at hipGetDevice on a machine without a device.
//
// The tables are mutated in an exact-size heap copy of [header | buffers | ops] followed by the (untouched, shared) weight blob only
// when the blob is small; for big plans the tables are copied to a heap block of exactly the truncated size so that any read past
// the end is an ASan error, and the untruncated mutants run in place (mutate, call, restore).
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "../../include/hsefr.h"

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint64_t rnd() {
    rng_state ^= rng_state << 13;
    rng_state ^= rng_state >> 7;
    rng_state ^= rng_state << 17;
    return rng_state;
}

static long long n_ok = 0, n_err = 0, n_silent = 0;

static void run_one(const void* blob, size_t bytes, const char* what) {
    int rc = hsefr_plan_validate(blob, bytes);
    if (rc == HSEFR_OK) {
        ++n_ok;
        // the launch wrappers' host side (shape checks, tile choices, grid arithmetic) on the mutant's shapes: every launcher runs with the
        // launch suppressed (hsefr_plan_describe) -- HSEFR_OK or a status with a message
        static char table[1 << 18];
        const int rd = hsefr_plan_describe(blob, bytes, 4, table, sizeof(table));
        if (rd != HSEFR_OK && !hsefr_last_error_string()[0]) { ++n_silent; fprintf(stderr, "no message for status %d (%s, plan_describe)\n", rd, what); }
        hsefr_engine* e = nullptr;
        rc = hsefr_engine_create(blob, bytes, 4, &e);       // valid: goes on to the device (none here: HSEFR_ERR_HIP / NOMEM) or succeeds
        if (rc == HSEFR_OK) hsefr_engine_destroy(e);
        else if (!hsefr_last_error_string()[0]) { ++n_silent; fprintf(stderr, "no message for status %d (%s, engine_create)\n", rc, what); }
        return;
    }
    ++n_err;
    if (rc > 0 || rc < HSEFR_ERR_SHAPE || !hsefr_last_error_string()[0]) {
        ++n_silent;
        fprintf(stderr, "status %d without a message / outside hsefr_status (%s)\n", rc, what);
    }
    hsefr_engine* e = nullptr;
    const int rc2 = hsefr_engine_create(blob, bytes, 4, &e);
    if (rc2 == HSEFR_OK) { fprintf(stderr, "engine_create accepted what plan_validate refused (%s)\n", what); abort(); }
}

int main(int argc, char** argv) {
    if (argc < 3) { fprintf(stderr, "usage: fuzz_plan <mutants per seed> <plan file>...\n"); return 2; }
    const long long per_seed = atoll(argv[1]);
    static const int32_t edge32[] = {0, 1, -1, -2, -3, 2, 3, 7, 63, 64, 65, 127, 128, 255, 256, 511, 512, 4095, 4096, 65535, 65536, 0x7fffffff,
                                     (int32_t)0x80000000, (int32_t)0xffff0000, 1 << 20, 1 << 24, 1 << 30};
    static const uint64_t edge64[] = {0ull, 1ull, 15ull, 16ull, 17ull, 0xffffffffull, 0x100000000ull, 0x7fffffffffffffffull, 0x8000000000000000ull,
                                      0xffffffffffffffffull, 0xfffffffffffffff0ull};
    for (int f = 2; f < argc; ++f) {
        FILE* fp = fopen(argv[f], "rb");
        if (!fp) { perror(argv[f]); return 2; }
        fseek(fp, 0, SEEK_END);
        const size_t bytes = (size_t)ftell(fp);
        fseek(fp, 0, SEEK_SET);
        std::vector<unsigned char> seed(bytes);
        if (fread(seed.data(), 1, bytes, fp) != bytes) { fprintf(stderr, "short read\n"); return 2; }
        fclose(fp);
        if (bytes < sizeof(hsefr_plan_header)) { fprintf(stderr, "%s: not a plan\n", argv[f]); return 2; }
        hsefr_plan_header h;
        memcpy(&h, seed.data(), sizeof(h));
        const size_t tables = sizeof(h) + (size_t)h.n_buffers * sizeof(hsefr_plan_buffer) + (size_t)h.n_ops * sizeof(hsefr_plan_op);
        if (hsefr_plan_validate(seed.data(), bytes) != HSEFR_OK) {
            fprintf(stderr, "%s: the seed itself is refused: %s\n", argv[f], hsefr_last_error_string());
            return 1;
        }
        // the working copy: exact size on the heap (reads past the end are ASan errors)
        unsigned char* w = (unsigned char*)malloc(bytes);
        memcpy(w, seed.data(), bytes);
        char what[160];
        for (long long it = 0; it < per_seed; ++it) {
            const unsigned mode = (unsigned)(rnd() % 10);
            if (mode == 0) {                                   // truncation: an exact-size copy of the first `cut` bytes
                size_t cut = (rnd() & 1) ? (size_t)(rnd() % (tables + 64 < bytes ? tables + 64 : bytes)) : (size_t)(rnd() % bytes);
                unsigned char* t = (unsigned char*)malloc(cut ? cut : 1);
                memcpy(t, seed.data(), cut);
                snprintf(what, sizeof(what), "%s truncated to %zu", argv[f], cut);
                run_one(t, cut, what);
                free(t);
                continue;
            }
            // 1-3 mutations inside the tables (where every offset, size and id lives), sometimes one in the blob
            const int nmut = 1 + (int)(rnd() % 3);
            size_t where[3];
            unsigned char saved[3][8];
            size_t len[3];
            for (int m = 0; m < nmut; ++m) {
                size_t off;
                if (mode == 1) {                               // a single bit anywhere in the tables
                    off = (size_t)(rnd() % tables);
                    len[m] = 1;
                    memcpy(saved[m], w + off, 1);
                    w[off] ^= (unsigned char)(1u << (rnd() % 8));
                } else if (mode == 2 && bytes > tables) {      // a byte in the weight blob (must not matter to validation)
                    off = tables + (size_t)(rnd() % (bytes - tables));
                    len[m] = 1;
                    memcpy(saved[m], w + off, 1);
                    w[off] = (unsigned char)rnd();
                } else if (mode <= 6) {                        // an aligned 32-bit field <- a boundary value
                    off = (size_t)(rnd() % (tables / 4)) * 4;
                    len[m] = 4;
                    memcpy(saved[m], w + off, 4);
                    const int32_t v = edge32[rnd() % (sizeof(edge32) / sizeof(edge32[0]))];
                    memcpy(w + off, &v, 4);
                } else if (mode <= 8) {                        // an aligned 64-bit field <- a boundary value / near the blob size
                    off = (size_t)(rnd() % (tables / 8)) * 8;
                    len[m] = 8;
                    memcpy(saved[m], w + off, 8);
                    uint64_t v = edge64[rnd() % (sizeof(edge64) / sizeof(edge64[0]))];
                    if (rnd() & 1) v = h.blob_bytes - 64 + (rnd() % 128);
                    if (rnd() % 4 == 0) v &= ~15ull;
                    memcpy(w + off, &v, 8);
                } else {                                       // a random 32-bit value
                    off = (size_t)(rnd() % (tables / 4)) * 4;
                    len[m] = 4;
                    memcpy(saved[m], w + off, 4);
                    const uint32_t v = (uint32_t)rnd();
                    memcpy(w + off, &v, 4);
                }
                where[m] = off;
            }
            snprintf(what, sizeof(what), "%s mode %u at %zu", argv[f], mode, where[0]);
            run_one(w, bytes, what);
            for (int m = nmut - 1; m >= 0; --m) memcpy(w + where[m], saved[m], len[m]);
        }
        if (memcmp(w, seed.data(), bytes) != 0) { fprintf(stderr, "internal: working copy not restored\n"); return 1; }
        free(w);
    }
    printf("fuzz_plan: %lld mutants still valid, %lld refused, %lld without a message\n", n_ok, n_err, n_silent);
    return n_silent ? 1 : 0;
}
