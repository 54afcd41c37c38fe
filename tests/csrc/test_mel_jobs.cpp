// Host check of the job packing of the mel filterbank (host_tables.cpp: build_mel_jobs):
// evaluating the jobs the way the 2048-point kernel does must reproduce basis @ power.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../speech_to_text_amd/csrc/host_tables.h"

static int check(int n_mels, double sr, double fmax, int htk, int max_rows, bool dense) {
    mfcc_config c;
    c.struct_size = (int32_t)sizeof(mfcc_config);
    c.n_fft = 2048; c.hop_length = 512; c.win_length = 0; c.center = 1; c.pad_mode = MFCC_PAD_CONSTANT;
    c.power = 2.0f; c.sr = sr; c.n_mels = n_mels; c.fmin = 0.0; c.fmax = fmax; c.htk = htk;
    c.mel_norm = MFCC_MEL_NORM_SLANEY; c.ref = 1.0f; c.amin = 1e-10f; c.top_db = 80.0f;
    c.n_mfcc = n_mels < 13 ? n_mels : 13; c.dct_type = 2; c.dct_norm = MFCC_DCT_NORM_ORTHO; c.lifter = 0.0f;
    c.window = nullptr; c.mel_basis = nullptr;
    mfcc_host::Resolved r;
    const std::string e = mfcc_host::resolve(&c, &r);
    if (!e.empty()) { std::printf("resolve: %s\n", e.c_str()); return 1; }
    std::vector<float> basis;
    mfcc_host::build_mel_basis(r, basis);
    const int nb = r.n_bins;
    if (dense) {
        std::srand(7);
        for (auto& v : basis) v = (std::rand() % 5 == 0) ? 0.0f : (float)std::rand() / RAND_MAX;
    }
    const int J = 3, ROW = 64, STRIDE = 80;
    mfcc_host::MelJobs mj;
    mfcc_host::build_mel_jobs(basis, n_mels, nb, J, ROW, STRIDE, max_rows, 4, mj);
    if (mj.n_jobs < 4) { std::printf("fewer than four jobs\n"); return 1; }
    // power spectrum in the padded layout; the padding holds finite junk
    std::vector<double> P(nb);
    std::vector<float> X(STRIDE * 17 + 64, 3.25f);
    std::srand(11);
    for (int k = 0; k < nb; ++k) {
        P[k] = std::exp(10.0 * std::rand() / RAND_MAX);
        X[k + (STRIDE - ROW) * (k / ROW)] = (float)P[k];
    }
    if (mj.max_read >= (int)X.size()) { std::printf("max_read %d too large\n", mj.max_read); return 1; }
    const int n_groups = (n_mels + 63) / 64;
    // cells start as junk except the rows the kernel clears
    std::vector<double> cells((size_t)(mj.n_rows + 1) * 64, 777.0);
    for (int r = 0; r < mj.n_rows; ++r)
        if (mj.zero_rows >> r & 1) for (int x = 0; x < 64; ++x) cells[(size_t)r * 64 + x] = 0.0;
    std::vector<int> writes(cells.size(), 0);
    for (int l = 0; l < 64; ++l) {
        double run = 0.0;
        for (int j = 0; j < mj.n_jobs; ++j) {
            double acc = 0.0;
            for (int i = 0; i < J; ++i)
                for (int q = 0; q < 4; ++q)
                    acc += (double)mj.w[((size_t)(j * J + i) * 64 + l) * 4 + q] * X[mj.at[j * 64 + l] + 4 * i + q];
            if (mj.cont[j * 64 + l] && j == 0) { std::printf("chain continues into slot 0\n"); return 1; }
            run = mj.cont[j * 64 + l] ? run + acc : acc;
            const int cell = mj.cell[j * 64 + l];
            if (cell < 0 || cell >= (int)cells.size()) { std::printf("cell out of range\n"); return 1; }
            if (cell >= mj.n_rows * 64) continue;       // mid-chain or idle
            if (mj.shared) cells[cell] += run;
            else { cells[cell] = run; if (++writes[cell] > 1) { std::printf("cell written twice\n"); return 1; } }
        }
    }
    double worst = 0.0;
    int row0 = 0;
    for (int p = 0; p < n_groups; ++p) {
        for (int x = 0; x < 64 && 64 * p + x < n_mels; ++x) {
            double got = 0.0, want = 0.0;
            for (int rr = 0; rr < mj.rows[p]; ++rr) got += cells[(size_t)(row0 + rr) * 64 + x];
            const int f = 64 * p + x;
            for (int k = 0; k < nb; ++k) want += (double)basis[(size_t)f * nb + k] * (float)P[k];
            worst = std::fmax(worst, std::fabs(got - want) / (std::fabs(want) + 1e-30));
        }
        row0 += mj.rows[p];
    }
    std::printf("n_mels %3d sr %6.0f htk %d dense %d: n_jobs %d rows %d+%d+%d+%d zero %x shared %d conflicts %d max_rel_err %.2e\n",
                n_mels, sr, htk, (int)dense, mj.n_jobs, mj.rows[0], mj.rows[1], mj.rows[2], mj.rows[3],
                mj.zero_rows, (int)mj.shared, mj.bank_conflicts, worst);
    return worst < 1e-12 ? 0 : 1;
}

int main() {
    int bad = 0;
    bad += check(128, 22050.0, 0.0, 0, 13, false);   // the reference's filterbank
    bad += check(128, 22050.0, 0.0, 0, 2, false);    // forces the shared-cell mode
    bad += check(40, 16000.0, 8000.0, 0, 13, false);
    bad += check(80, 16000.0, 7600.0, 1, 13, false);
    bad += check(256, 44100.0, 0.0, 0, 13, false);
    bad += check(1, 22050.0, 0.0, 0, 13, false);
    bad += check(10, 22050.0, 0.0, 0, 13, false);
    bad += check(20, 22050.0, 0.0, 0, 13, true);     // arbitrary (dense) user filterbank
    std::printf(bad ? "FAILED\n" : "mel_jobs_ok\n");
    return bad;
}
