/* The C ABI used from plain C (C99), host-only entry points: the header must stand on its own
 * and the calls must behave as include/mfcc_hip.h says.  No GPU work. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "mfcc_hip.h"

#define CHECK(cond)                                                        \
    do {                                                                   \
        if (!(cond)) {                                                     \
            printf("FAILED line %d: %s\n", __LINE__, #cond);               \
            return 1;                                                      \
        }                                                                  \
    } while (0)

int main(int argc, char** argv) {
    const char* dir = argc > 1 ? argv[1] : "/tmp";
    mfcc_config c;
    mfcc_config_default(&c);
    CHECK(mfcc_abi_version() == MFCC_ABI_VERSION);
    CHECK(c.n_fft == 2048 && c.hop_length == 512 && c.n_mels == 128 && c.center == 1);
    c.n_mfcc = 13;                              /* /root/reference/src/preprocess.py:60-62 */
    CHECK(mfcc_num_frames(&c, 166960) == 327);  /* 1 + T / hop for a centred transform */

    int64_t so[4] = {0, 166960, 166960 + 32000, 166960 + 32000 + 511};
    int64_t fo[4];
    CHECK(mfcc_batch_layout(&c, so, 3, fo) == MFCC_OK);
    CHECK(fo[0] == 0 && fo[1] == 327 && fo[2] == 327 + 63 && fo[3] == 327 + 63 + 1);

    /* packing: three ragged utterances, a sub-range that straddles two of them, float -> int16 */
    float u0[5] = {0.5f, -0.25f, 0.0f, 0.125f, -1.0f}, u1[1] = {0.75f}, u2[3] = {1.0f / 32768.0f, -0.5f, 0.25f};
    const void* srcs[3] = {u0, u1, u2};
    int64_t off[4] = {0, 5, 6, 9};
    int16_t dst[9];
    memset(dst, 0x55, sizeof dst);
    CHECK(mfcc_host_pack_range(srcs, off, 3, dst, 0, 3, 8, 2) == 1);
    CHECK(dst[3] == 4096 && dst[4] == -32768 && dst[5] == 24576 && dst[6] == 1 && dst[7] == -16384);
    CHECK(dst[2] == 0x5555 && dst[8] == 0x5555);
    u2[1] = 0.3f;                               /* off the int16 grid */
    CHECK(mfcc_host_pack_range(srcs, off, 3, dst, 0, 0, 9, 2) == 0);

    /* two feature files from one buffer, on two threads */
    float feats[5 * 13];
    for (int i = 0; i < 5 * 13; ++i) feats[i] = (float)i * 0.5f;
    char p0[512], p1[512];
    snprintf(p0, sizeof p0, "%s/10_20_0.hdf5", dir);
    snprintf(p1, sizeof p1, "%s/10_20_1.hdf5", dir);
    const char* paths[2] = {p0, p1};
    const float* data[2] = {feats, feats + 2 * 13};
    int64_t rows[2] = {2, 3};
    const char* names[5] = {"speaker_id", "chapter_id", "utterance_id", "sample_rate", "transcript"};
    int32_t kinds[5] = {0, 0, 0, 0, 1};
    int64_t ints[10] = {10, 20, 0, 16000, 0, 10, 20, 1, 16000, 0};
    const char* strs[10] = {0, 0, 0, 0, "FIRST", 0, 0, 0, 0, "SECOND"};
    CHECK(mfcc_host_write_feature_files_mt(2, paths, "mfccs", data, rows, 13, 5, names, kinds, ints, strs, 2) == MFCC_OK);
    FILE* f = fopen(p1, "rb");
    CHECK(f != NULL);
    unsigned char magic[8];
    CHECK(fread(magic, 1, 8, f) == 8 && memcmp(magic, "\x89HDF\r\n\x1a\n", 8) == 0);
    fclose(f);
    const char* bad[1] = {"/nonexistent-directory/x/y.hdf5"};
    CHECK(mfcc_host_write_feature_files_mt(1, bad, "mfccs", data, rows, 13, 5, names, kinds, ints, strs, 1) == MFCC_E_INVALID);
    CHECK(strlen(mfcc_host_file_error()) > 0);
    printf("c_abi_ok\n");
    return 0;
}
