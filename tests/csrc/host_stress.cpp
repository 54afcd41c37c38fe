// Host-only entry points of the C ABI under randomised inputs, meant to be linked against a build
// of the library whose host code carries AddressSanitizer + UBSan (tools/host_sanitize.sh) or
// ThreadSanitizer.  Buffers are heap blocks of exactly the documented sizes, so a write or read
// one element past what include/mfcc_hip.h promises trips the sanitizer.  No GPU work.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <thread>
#include <vector>

#include <dirent.h>
#include <sys/stat.h>
#include <unistd.h>

#include "mfcc_hip.h"

#define CHECK(cond)                                                          \
    do {                                                                     \
        if (!(cond)) {                                                       \
            std::printf("FAILED line %d: %s\n", __LINE__, #cond);            \
            std::exit(1);                                                    \
        }                                                                    \
    } while (0)

namespace {

std::mt19937_64 rng(20240611);

int64_t uniform(int64_t lo, int64_t hi) { return std::uniform_int_distribution<int64_t>(lo, hi)(rng); }

mfcc_config random_config() {
    mfcc_config c;
    mfcc_config_default(&c);
    static const int ffts[] = {2048, 1024, 960, 800, 768, 640, 600, 576, 512, 480, 400, 384, 320, 256, 192, 160, 128, 64, 30};
    c.n_fft = ffts[uniform(0, (int64_t)(sizeof ffts / sizeof ffts[0]) - 1)];
    c.hop_length = (int32_t)uniform(1, c.n_fft);
    c.win_length = uniform(0, 2) ? 0 : (int32_t)uniform(1, c.n_fft);
    c.center = uniform(0, 3) != 0;
    c.pad_mode = (int32_t)uniform(0, 1);
    static const int mels[] = {1, 13, 20, 40, 64, 80, 128, 200, 256};
    c.n_mels = mels[uniform(0, 8)];
    c.n_mfcc = (int32_t)uniform(1, c.n_mels);
    static const double srs[] = {8000, 16000, 22050, 44100, 48000};
    c.sr = srs[uniform(0, 4)];
    c.fmin = uniform(0, 2) ? 0.0 : 50.0;
    c.fmax = uniform(0, 2) ? 0.0 : c.sr / 2.5;
    c.htk = (int32_t)uniform(0, 1);
    c.mel_norm = (int32_t)uniform(0, 1);
    c.dct_type = (int32_t)uniform(1, 3);
    c.dct_norm = c.dct_type == 1 ? MFCC_DCT_NORM_NONE : (int32_t)uniform(0, 1);
    if (c.dct_type == 1 && c.n_mels < 2) c.n_mels = 2;
    if (c.n_mfcc > c.n_mels) c.n_mfcc = c.n_mels;
    return c;
}

void tables() {
    for (int it = 0; it < 300; ++it) {
        const mfcc_config c = random_config();
        const int bins = 1 + c.n_fft / 2;
        // exact-size heap blocks: one element too many written is a heap overflow
        std::vector<double> win((size_t)c.n_fft);
        std::vector<float> mel((size_t)c.n_mels * bins), dct((size_t)c.n_mfcc * c.n_mels);
        const int rw = mfcc_host_window(&c, win.data());
        const int rm = mfcc_host_mel_basis(&c, mel.data());
        const int rd = mfcc_host_dct_basis(&c, dct.data());
        CHECK(rw == MFCC_OK || rw == MFCC_E_INVALID);
        CHECK(rm == MFCC_OK || rm == MFCC_E_INVALID);
        CHECK(rd == MFCC_OK || rd == MFCC_E_INVALID);
        if (rw == MFCC_OK)
            for (double v : win) CHECK(v >= -1e-12 && v <= 1.0 + 1e-12);
        if (rm == MFCC_OK)
            for (float v : mel) CHECK(v >= 0.0f && v == v);
        if (rd == MFCC_OK)
            for (float v : dct) CHECK(v == v && v > -4.0f && v < 4.0f);
        // user-supplied window / filterbank of exactly the documented sizes are only read
        mfcc_config c2 = c;
        std::vector<double> user_win((size_t)(c.win_length ? c.win_length : c.n_fft), 0.5);
        c2.window = user_win.data();
        c2.mel_basis = mel.data();
        (void)mfcc_host_window(&c2, win.data());
        (void)mfcc_host_mel_basis(&c2, mel.data());
    }
    // arguments that must be refused, not dereferenced
    mfcc_config c;
    mfcc_config_default(&c);
    CHECK(mfcc_host_window(nullptr, nullptr) != MFCC_OK);
    CHECK(mfcc_host_window(&c, nullptr) != MFCC_OK);
    CHECK(mfcc_host_mel_basis(&c, nullptr) != MFCC_OK);
    CHECK(mfcc_host_dct_basis(&c, nullptr) != MFCC_OK);
    c.n_fft = 0;
    std::vector<double> w(16);
    CHECK(mfcc_host_window(&c, w.data()) != MFCC_OK);
    mfcc_config_default(&c);
    c.struct_size = 8;
    mfcc_plan* plan = nullptr;
    CHECK(mfcc_plan_create(&c, &plan) != MFCC_OK && plan == nullptr);
    CHECK(std::strlen(mfcc_last_error()) > 0);
}

void layouts() {
    for (int it = 0; it < 500; ++it) {
        mfcc_config c = random_config();
        const int n = (int)uniform(0, 50);
        std::vector<int64_t> so((size_t)n + 1, 0), fo((size_t)n + 1, -7);
        for (int u = 0; u < n; ++u) so[(size_t)u + 1] = so[(size_t)u] + uniform(0, uniform(0, 1) ? 20 : 100000);
        const int rc = mfcc_batch_layout(&c, so.data(), n, fo.data());
        if (rc == MFCC_OK) {
            CHECK(fo[0] == 0);
            for (int u = 0; u < n; ++u) {
                const int64_t f = mfcc_num_frames(&c, so[(size_t)u + 1] - so[(size_t)u]);
                CHECK(f >= 0 && fo[(size_t)u + 1] - fo[(size_t)u] == f);
            }
        }
    }
    mfcc_config c;
    mfcc_config_default(&c);
    int64_t one[2] = {0, 10};
    CHECK(mfcc_batch_layout(&c, nullptr, 1, one) != MFCC_OK);
    CHECK(mfcc_batch_layout(&c, one, 1, nullptr) != MFCC_OK);
    CHECK(mfcc_batch_layout(&c, one, -1, one) != MFCC_OK);
    int64_t back[3] = {0, 10, 5};               // offsets that run backwards
    int64_t fo[3];
    CHECK(mfcc_batch_layout(&c, back, 2, fo) != MFCC_OK);
    CHECK(mfcc_num_frames(&c, -1) < 0);
    CHECK(mfcc_num_frames(&c, INT64_MAX / 2) > 0);
}

void packing() {
    for (int it = 0; it < 400; ++it) {
        const int n = (int)uniform(1, 30);
        const int mode = (int)uniform(0, 2);
        std::vector<std::vector<float>> f32((size_t)n);
        std::vector<std::vector<int16_t>> i16((size_t)n);
        std::vector<const void*> srcs((size_t)n);
        std::vector<int64_t> off((size_t)n + 1, 0);
        for (int u = 0; u < n; ++u) {
            const int64_t len = uniform(0, 3) ? uniform(1, 5000) : 0;      // empty utterances too
            off[(size_t)u + 1] = off[(size_t)u] + len;
            if (mode == 1) {
                i16[(size_t)u].resize((size_t)len);
                for (auto& v : i16[(size_t)u]) v = (int16_t)uniform(-32768, 32767);
                srcs[(size_t)u] = i16[(size_t)u].data();
            } else {
                f32[(size_t)u].resize((size_t)len);
                for (auto& v : f32[(size_t)u]) v = (float)uniform(-32768, 32767) / 32768.0f;
                srcs[(size_t)u] = f32[(size_t)u].data();
            }
        }
        const int64_t total = off[(size_t)n];
        if (total == 0) continue;
        // a staging chunk [c0, c1) cut into pieces as the packing threads cut it
        const int64_t c0 = uniform(0, total - 1), c1 = uniform(c0 + 1, total);
        const size_t esz = mode == 0 ? 4 : 2;
        std::vector<unsigned char> dst((size_t)(c1 - c0) * esz);            // exact size
        int64_t a = c0;
        while (a < c1) {
            const int64_t b = std::min<int64_t>(c1, a + uniform(1, 3000));
            CHECK(mfcc_host_pack_range(srcs.data(), off.data(), n, dst.data(), c0, a, b, mode) == 1);
            a = b;
        }
        // compare with the concatenation
        int64_t u = 0;
        for (int64_t s = c0; s < c1; ++s) {
            while (off[(size_t)u + 1] <= s) ++u;
            const int64_t at = s - off[(size_t)u];
            if (mode == 0) {
                float got;
                std::memcpy(&got, dst.data() + (size_t)(s - c0) * 4, 4);
                CHECK(got == f32[(size_t)u][(size_t)at]);
            } else {
                int16_t got;
                std::memcpy(&got, dst.data() + (size_t)(s - c0) * 2, 2);
                const int16_t want = mode == 1 ? i16[(size_t)u][(size_t)at] : (int16_t)(f32[(size_t)u][(size_t)at] * 32768.0f);
                CHECK(got == want);
            }
        }
        CHECK(mfcc_host_pack_range(srcs.data(), off.data(), n, dst.data(), c0, c0, c0, mode) == 1);   // empty range
        CHECK(mfcc_host_pack_range(srcs.data(), off.data(), n, dst.data(), c0, c0 - 1 < 0 ? -1 : c0 - 1, c0, mode) < 0);
    }
    // the float -> int16 conversion: special values are off the grid, never undefined behaviour
    const float odd[] = {0.3f, 1.0f, -1.0f, 2.0f, 1e30f, -1e30f, __builtin_inff(), -__builtin_inff(), __builtin_nanf(""), 1.0f / 65536.0f};
    for (float v : odd) {
        std::vector<float> src(37, 0.25f);
        src[(size_t)uniform(0, 36)] = v;
        std::vector<int16_t> out(37);
        const int rc = mfcc_host_pcm16_from_float(src.data(), 37, out.data());
        CHECK(rc == (v == -1.0f ? 1 : 0));
    }
    CHECK(mfcc_host_pcm16_from_float(nullptr, 0, nullptr) == 1);
    CHECK(mfcc_host_pcm16_from_float(nullptr, 4, nullptr) < 0);
}

int count_entries(const std::string& dir) {
    DIR* d = opendir(dir.c_str());
    if (!d) return -1;
    int n = 0;
    while (dirent* e = readdir(d))
        if (std::strcmp(e->d_name, ".") && std::strcmp(e->d_name, "..")) ++n;
    closedir(d);
    return n;
}

void files(const std::string& dir) {
    const char* names[5] = {"speaker_id", "chapter_id", "utterance_id", "sample_rate", "transcript"};
    const int32_t kinds[5] = {0, 0, 0, 0, 1};
    for (int it = 0; it < 12; ++it) {
        const int n = (int)uniform(1, 300);
        const int cols = (int)uniform(1, 40);
        std::vector<int64_t> rows((size_t)n), ro((size_t)n + 1, 0);
        for (int i = 0; i < n; ++i) {
            rows[(size_t)i] = uniform(0, 5) ? uniform(1, 400) : 0;          // empty datasets too
            ro[(size_t)i + 1] = ro[(size_t)i] + rows[(size_t)i];
        }
        std::vector<float> base((size_t)ro[(size_t)n] * cols);               // exact size
        for (size_t i = 0; i < base.size(); ++i) base[i] = (float)i;
        std::vector<std::string> path_s((size_t)n), text((size_t)n);
        std::vector<const char*> paths((size_t)n), strs((size_t)n * 5, nullptr);
        std::vector<const float*> data((size_t)n);
        std::vector<int64_t> ints((size_t)n * 5, 0);
        for (int i = 0; i < n; ++i) {
            path_s[(size_t)i] = dir + "/" + std::to_string(it) + "_" + std::to_string(i) + ".hdf5";
            paths[(size_t)i] = path_s[(size_t)i].c_str();
            data[(size_t)i] = base.data() + (size_t)ro[(size_t)i] * cols;
            text[(size_t)i].assign((size_t)uniform(0, 600), 'A' + (char)(i % 26));   // empty transcripts too
            if (uniform(0, 4) == 0) text[(size_t)i] += "\xc3\xa9\xe2\x82\xac";        // multi-byte UTF-8
            strs[(size_t)i * 5 + 4] = text[(size_t)i].c_str();
            ints[(size_t)i * 5 + 0] = uniform(0, 9999);
            ints[(size_t)i * 5 + 1] = uniform(0, 999999);
            ints[(size_t)i * 5 + 2] = i;
            ints[(size_t)i * 5 + 3] = 16000;
        }
        const int threads = (int)uniform(0, 9);
        CHECK(mfcc_host_write_feature_files_mt(n, paths.data(), "mfccs", data.data(), rows.data(), cols, 5, names, kinds,
                                               ints.data(), strs.data(), threads) == MFCC_OK);
        // every file is there, starts with the signature and holds its dataset's bytes verbatim
        for (int i = 0; i < n; ++i) {
            FILE* f = std::fopen(paths[(size_t)i], "rb");
            CHECK(f != nullptr);
            std::vector<unsigned char> all;
            unsigned char buf[65536];
            size_t got;
            while ((got = std::fread(buf, 1, sizeof buf, f)) > 0) all.insert(all.end(), buf, buf + got);
            std::fclose(f);
            CHECK(all.size() >= 8 && std::memcmp(all.data(), "\x89HDF\r\n\x1a\n", 8) == 0);
            const size_t nbytes = (size_t)rows[(size_t)i] * cols * 4;
            if (nbytes) {
                bool found = false;
                const unsigned char* want = reinterpret_cast<const unsigned char*>(data[(size_t)i]);
                for (size_t at = 0; at + nbytes <= all.size() && !found; at += 8)
                    found = std::memcmp(all.data() + at, want, nbytes) == 0;
                CHECK(found);
            }
            ::unlink(paths[(size_t)i]);
        }
        CHECK(count_entries(dir) == 0);          // no temporaries or side directories left behind
    }
    // single-file and in-order variants, overwrite of an existing file, refusal of bad arguments
    const std::string p = dir + "/one.hdf5";
    float x[6] = {1, 2, 3, 4, 5, 6};
    int64_t iv[5] = {1, 2, 3, 16000, 0};
    const char* sv[5] = {nullptr, nullptr, nullptr, nullptr, "HELLO"};
    for (int rep = 0; rep < 2; ++rep)
        CHECK(mfcc_host_write_feature_file(p.c_str(), "mfccs", x, 2, 3, 5, names, kinds, iv, sv) == MFCC_OK);
    const char* pp[1] = {p.c_str()};
    const float* dd[1] = {x};
    int64_t rr[1] = {2};
    CHECK(mfcc_host_write_feature_files(1, pp, "mfccs", dd, rr, 3, 5, names, kinds, iv, sv) == MFCC_OK);
    CHECK(mfcc_host_write_feature_files(1, pp, "mfccs", dd, rr, 3, 0, nullptr, nullptr, nullptr, nullptr) == MFCC_OK);
    CHECK(mfcc_host_write_feature_file(nullptr, "mfccs", x, 2, 3, 5, names, kinds, iv, sv) != MFCC_OK);
    CHECK(mfcc_host_write_feature_file(p.c_str(), nullptr, x, 2, 3, 5, names, kinds, iv, sv) != MFCC_OK);
    CHECK(mfcc_host_write_feature_file(p.c_str(), "mfccs", nullptr, 2, 3, 5, names, kinds, iv, sv) != MFCC_OK);
    CHECK(mfcc_host_write_feature_file(p.c_str(), "mfccs", x, -1, 3, 5, names, kinds, iv, sv) != MFCC_OK);
    CHECK(mfcc_host_write_feature_file(p.c_str(), "mfccs", x, 2, 3, 5, nullptr, kinds, iv, sv) != MFCC_OK);
    const char* no_text[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    (void)mfcc_host_write_feature_file(p.c_str(), "mfccs", x, 2, 3, 5, names, kinds, iv, no_text);   // refused or written as ""
    CHECK(mfcc_host_write_feature_files_mt(0, nullptr, "mfccs", nullptr, nullptr, 3, 0, nullptr, nullptr, nullptr, nullptr, 4) == MFCC_OK);
    ::unlink(p.c_str());
    // several callers at once, each with its own batch (the writer keeps no shared state but the error text)
    std::vector<std::thread> callers;
    for (int t = 0; t < 4; ++t)
        callers.emplace_back([&, t]() {
            std::vector<std::string> ps;
            std::vector<const char*> pc;
            std::vector<const float*> dc;
            std::vector<int64_t> rc, ic;
            std::vector<const char*> sc;
            for (int i = 0; i < 40; ++i) ps.push_back(dir + "/t" + std::to_string(t) + "_" + std::to_string(i) + ".hdf5");
            for (int i = 0; i < 40; ++i) {
                pc.push_back(ps[(size_t)i].c_str());
                dc.push_back(x);
                rc.push_back(2);
                for (int k = 0; k < 5; ++k) { ic.push_back(k == 3 ? 16000 : i); sc.push_back(k == 4 ? "TEXT" : nullptr); }
            }
            CHECK(mfcc_host_write_feature_files_mt(40, pc.data(), "mfccs", dc.data(), rc.data(), 3, 5, names, kinds, ic.data(),
                                                   sc.data(), 3) == MFCC_OK);
            const char* bad[1] = {"/nonexistent-directory/x/y.hdf5"};
            CHECK(mfcc_host_write_feature_files_mt(1, bad, "mfccs", dc.data(), rc.data(), 3, 5, names, kinds, ic.data(), sc.data(), 2) != MFCC_OK);
            CHECK(std::strlen(mfcc_host_file_error()) > 0);
            for (const std::string& s : ps) ::unlink(s.c_str());
        });
    for (std::thread& th : callers) th.join();
    CHECK(count_entries(dir) == 0);
}

}  // namespace

int main(int argc, char** argv) {
    const std::string dir = argc > 1 ? argv[1] : "/tmp";
    CHECK(mfcc_abi_version() == MFCC_ABI_VERSION);
    tables();
    layouts();
    packing();
    files(dir);
    std::printf("host_stress_ok\n");
    return 0;
}
