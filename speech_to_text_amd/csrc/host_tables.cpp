// Host-side table construction (see host_tables.h).  Double precision throughout, rounded to
// float32 exactly where librosa / scipy round.
#include "host_tables.h"

#include <algorithm>
#include <cmath>
#include <numeric>

namespace mfcc_host {

static const double kPi = 3.14159265358979323846;

std::string resolve(const mfcc_config* in, Resolved* out) {
    if (!in) return "config is NULL";
    if (in->struct_size != (int32_t)sizeof(mfcc_config))
        return "mfcc_config.struct_size does not match this library (ABI mismatch)";
    mfcc_config c = *in;
    if (c.n_fft < 2) return "n_fft must be at least 2";
    if (c.n_fft % 2) return "odd n_fft is not supported";
    if (c.hop_length <= 0) return "hop_length must be a positive integer";
    if (c.win_length == 0) c.win_length = c.n_fft;
    if (c.win_length < 1 || c.win_length > c.n_fft)
        return "win_length must be in [1, n_fft]";  // pad_center: target size >= input size
    if (c.pad_mode != MFCC_PAD_CONSTANT && c.pad_mode != MFCC_PAD_REFLECT)
        return "pad_mode must be 'constant' or 'reflect'";
    if (!(c.power > 0.0f)) return "power must be positive";
    if (!(c.sr > 0.0)) return "sr must be positive";
    if (c.n_mels < 1) return "n_mels must be positive";
    if (c.fmax <= 0.0) c.fmax = c.sr / 2.0;
    if (!(c.amin > 0.0f)) return "amin must be strictly positive";
    if (c.n_mfcc < 1 || c.n_mfcc > c.n_mels) return "n_mfcc must be in [1, n_mels]";
    if (c.dct_type < 1 || c.dct_type > 3) return "dct_type must be 1, 2 or 3";
    if (c.dct_type == 1 && c.n_mels < 2) return "DCT-I needs n_mels >= 2";
    if (c.lifter < 0.0f) return "MFCC lifter must be a non-negative number";
    if (c.mel_norm != MFCC_MEL_NORM_NONE && c.mel_norm != MFCC_MEL_NORM_SLANEY)
        return "unsupported mel norm";
    if (c.dct_norm != MFCC_DCT_NORM_NONE && c.dct_norm != MFCC_DCT_NORM_ORTHO)
        return "unsupported dct norm";
    out->c = c;
    out->n_bins = 1 + c.n_fft / 2;
    return "";
}

int64_t num_frames(const Resolved& r, int64_t n_samples) {
    if (n_samples < 0) return -1;
    int64_t padded = n_samples + (r.c.center ? 2 * (int64_t)(r.c.n_fft / 2) : 0);
    if (padded < r.c.n_fft) return -1;
    return 1 + (padded - r.c.n_fft) / r.c.hop_length;
}

// scipy.signal.get_window('hann', M, fftbins=True): the symmetric window of M+1 points,
// 0.5 + 0.5 cos(linspace(-pi, pi, M+1)), without its last sample; then librosa pad_center.
void build_window(const Resolved& r, std::vector<double>& w) {
    const int n_fft = r.c.n_fft, wl = r.c.win_length;
    w.assign(n_fft, 0.0);
    const int lpad = (n_fft - wl) / 2;
    for (int n = 0; n < wl; ++n) {
        double v;
        if (r.c.window) {
            v = r.c.window[n];
        } else if (wl == 1) {
            v = 1.0;
        } else {
            const double step = (2.0 * kPi) / wl;
            v = 0.5 + 0.5 * std::cos(-kPi + n * step);
        }
        w[lpad + n] = v;
    }
}

static double hz_to_mel(double f, bool htk) {
    if (htk) return 2595.0 * std::log10(1.0 + f / 700.0);
    const double f_sp = 200.0 / 3;
    const double min_log_hz = 1000.0, min_log_mel = min_log_hz / f_sp;
    const double logstep = std::log(6.4) / 27.0;
    if (f >= min_log_hz) return min_log_mel + std::log(f / min_log_hz) / logstep;
    return f / f_sp;
}

static double mel_to_hz(double m, bool htk) {
    if (htk) return 700.0 * (std::pow(10.0, m / 2595.0) - 1.0);
    const double f_sp = 200.0 / 3;
    const double min_log_hz = 1000.0, min_log_mel = min_log_hz / f_sp;
    const double logstep = std::log(6.4) / 27.0;
    if (m >= min_log_mel) return min_log_hz * std::exp(logstep * (m - min_log_mel));
    return f_sp * m;
}

void build_mel_basis(const Resolved& r, std::vector<float>& out) {
    const int n_mels = r.c.n_mels, n_bins = r.n_bins, n_fft = r.c.n_fft;
    out.assign((size_t)n_mels * n_bins, 0.0f);
    if (r.c.mel_basis) {
        std::copy(r.c.mel_basis, r.c.mel_basis + (size_t)n_mels * n_bins, out.begin());
        return;
    }
    const bool htk = r.c.htk != 0;
    // np.fft.rfftfreq(n, d): arange(n//2+1) * (1.0 / (n * d)), d = 1.0 / sr
    const double val = 1.0 / (n_fft * (1.0 / r.c.sr));
    std::vector<double> fftfreqs(n_bins);
    for (int k = 0; k < n_bins; ++k) fftfreqs[k] = k * val;
    // np.linspace(min_mel, max_mel, n_mels + 2)
    const int npts = n_mels + 2;
    const double min_mel = hz_to_mel(r.c.fmin, htk), max_mel = hz_to_mel(r.c.fmax, htk);
    const double step = (max_mel - min_mel) / (npts - 1);
    std::vector<double> mel_f(npts);
    for (int i = 0; i < npts; ++i) {
        const double m = (i == npts - 1) ? max_mel : i * step + min_mel;
        mel_f[i] = mel_to_hz(m, htk);
    }
    for (int i = 0; i < n_mels; ++i) {
        const double fd0 = mel_f[i + 1] - mel_f[i], fd1 = mel_f[i + 2] - mel_f[i + 1];
        const double enorm = 2.0 / (mel_f[i + 2] - mel_f[i]);
        for (int k = 0; k < n_bins; ++k) {
            const double lower = -(mel_f[i] - fftfreqs[k]) / fd0;
            const double upper = (mel_f[i + 2] - fftfreqs[k]) / fd1;
            // weights[i] = np.maximum(0, np.minimum(lower, upper)) stored to a float32 array
            float wv = (float)std::max(0.0, std::min(lower, upper));
            // weights *= enorm[:, None]: float32 (x) float64 evaluated in double, stored float32
            if (r.c.mel_norm == MFCC_MEL_NORM_SLANEY) wv = (float)((double)wv * enorm);
            out[(size_t)i * n_bins + k] = wv;
        }
    }
}

void build_dct_basis(const Resolved& r, std::vector<float>& d) {
    const int N = r.c.n_mels, K = r.c.n_mfcc;
    const bool ortho = r.c.dct_norm == MFCC_DCT_NORM_ORTHO;
    d.assign((size_t)K * N, 0.0f);
    for (int k = 0; k < K; ++k) {
        for (int n = 0; n < N; ++n) {
            double v = 0.0;
            if (r.c.dct_type == 2) {
                v = 2.0 * std::cos(kPi * k * (2.0 * n + 1.0) / (2.0 * N));
                if (ortho) v *= (k == 0) ? std::sqrt(1.0 / (4.0 * N)) : std::sqrt(1.0 / (2.0 * N));
            } else if (r.c.dct_type == 3) {
                if (ortho)
                    v = (n == 0) ? 1.0 / std::sqrt((double)N)
                                 : std::sqrt(2.0 / N) * std::cos(kPi * (2.0 * k + 1.0) * n / (2.0 * N));
                else
                    v = (n == 0) ? 1.0 : 2.0 * std::cos(kPi * (2.0 * k + 1.0) * n / (2.0 * N));
            } else {  // type 1
                const double c = std::cos(kPi * k * n / (N - 1.0));
                if (n == 0) v = 1.0;
                else if (n == N - 1) v = (k % 2) ? -1.0 : 1.0;
                else v = 2.0 * c;
                if (ortho) {
                    if (n == 0 || n == N - 1) v *= std::sqrt(2.0);
                    v *= std::sqrt(1.0 / (2.0 * (N - 1.0)));
                    if (k == 0 || k == N - 1) v *= 1.0 / std::sqrt(2.0);
                }
            }
            d[(size_t)k * N + n] = (float)v;
        }
    }
}

// librosa.feature.mfcc: M *= 1 + (lifter / 2) * sin(pi * arange(1, 1 + n_mfcc, float32) / lifter)
void build_lifter(const Resolved& r, std::vector<float>& l) {
    l.assign(r.c.n_mfcc, 1.0f);
    if (r.c.lifter > 0.0f) {
        for (int k = 0; k < r.c.n_mfcc; ++k) {
            const float a = (float)(kPi) * (float)(k + 1) / r.c.lifter;
            l[k] = 1.0f + (r.c.lifter / 2.0f) * std::sin(a);
        }
    }
}

// ds_read_b128 services a wavefront in four groups of 16 lanes; only lanes of one group can
// bank-conflict, and they do not if their 16-byte slots differ modulo 16.
static const int kB128Groups[4][16] = {
    {0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27},
    {4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31},
    {32, 33, 34, 35, 44, 45, 46, 47, 52, 53, 54, 55, 56, 57, 58, 59},
    {36, 37, 38, 39, 40, 41, 42, 43, 48, 49, 50, 51, 60, 61, 62, 63}};

namespace {
// Kuhn's bipartite matching: filters of one pass -> (lane group, slot residue) pairs.
struct Matcher {
    int n;                                   // filters in the pass
    std::vector<std::vector<int>> feas;      // feasible residues per filter
    int owner[4][16];
    bool try_assign(int i, bool (&seen)[4][16]) {
        for (int g = 0; g < 4; ++g)
            for (int r : feas[i]) {
                if (seen[g][r]) continue;
                seen[g][r] = true;
                if (owner[g][r] < 0 || try_assign(owner[g][r], seen)) { owner[g][r] = i; return true; }
            }
        return false;
    }
};
}  // namespace

void build_sparse_mel(const std::vector<float>& basis, int n_mels, int n_bins, int chunk_quads, SparseMel& s) {
    std::vector<int> first(n_mels, 0), last(n_mels, -1);
    for (int m = 0; m < n_mels; ++m) {
        int lo = -1, hi = -1;
        for (int k = 0; k < n_bins; ++k)
            if (basis[(size_t)m * n_bins + k] != 0.0f) { if (lo < 0) lo = k; hi = k; }
        first[m] = lo < 0 ? 0 : lo;
        last[m] = lo < 0 ? -1 : hi;
    }
    auto quads = [&](int m) { return last[m] < 0 ? 0 : last[m] / 4 - first[m] / 4 + 1; };
    std::vector<int> order(n_mels);
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return quads(a) > quads(b); });
    // Bins are read four at a time (ds_read_b128): a lane starts at a multiple of 4 at or below
    // its filter's first non-zero bin, possibly earlier (zero weights) so that the lanes of each
    // 16-lane group hit distinct LDS bank groups.
    s.n_pass = (n_mels + 63) / 64;
    s.rounds.assign(s.n_pass, 0);
    s.round_off.assign(s.n_pass, 0);
    s.start.assign((size_t)s.n_pass * 64, 0);
    s.filt.assign((size_t)s.n_pass * 64, -1);
    s.total_rounds = 0;
    s.max_read = 0;
    std::vector<std::vector<int>> pass_lane_filter(s.n_pass, std::vector<int>(64, -1));
    std::vector<std::vector<int>> pass_lane_start(s.n_pass, std::vector<int>(64, 0));
    for (int p = 0; p < s.n_pass; ++p) {
        const int nf = std::min(64, n_mels - p * 64);
        int rq = 0;
        for (int l = 0; l < nf; ++l) rq = std::max(rq, quads(order[p * 64 + l]));
        rq = std::max(chunk_quads, (rq + chunk_quads - 1) / chunk_quads * chunk_quads);
        Matcher mt;
        mt.n = nf;
        bool matched = false;
        for (int attempt = 0; attempt < 2 && !matched; ++attempt) {
            mt.feas.assign(nf, {});
            for (int i = 0; i < nf; ++i) {
                const int m = order[p * 64 + i];
                const int s4 = first[m] / 4, slack = rq - quads(m);
                for (int d = 0; d <= slack && d < 16 && s4 - d >= 0; ++d)
                    mt.feas[i].push_back(((s4 - d) % 16 + 16) % 16);
                if (attempt == 1) mt.feas[i].resize(1);  // fallback: natural starts, conflicts allowed
            }
            for (auto& g : mt.owner) for (int& o : g) o = -1;
            if (attempt == 0) {
                std::vector<int> idx(nf);
                std::iota(idx.begin(), idx.end(), 0);
                std::stable_sort(idx.begin(), idx.end(),
                                 [&](int a, int b) { return mt.feas[a].size() < mt.feas[b].size(); });
                matched = true;
                for (int i : idx) {
                    bool seen[4][16] = {};
                    if (!mt.try_assign(i, seen)) { matched = false; break; }
                }
            } else {
                matched = true;  // handled below: plain lane order
            }
        }
        // lanes
        bool used_match = false;
        {
            int cnt = 0;
            for (auto& g : mt.owner) for (int o : g) cnt += o >= 0;
            used_match = cnt == nf;
        }
        if (used_match) {
            for (int g = 0; g < 4; ++g) {
                int j = 0;
                for (int r = 0; r < 16; ++r) {
                    const int i = mt.owner[g][r];
                    if (i < 0) continue;
                    const int m = order[p * 64 + i];
                    const int s4 = first[m] / 4;
                    const int d = ((s4 - r) % 16 + 16) % 16;
                    const int lane = kB128Groups[g][j++];
                    pass_lane_filter[p][lane] = m;
                    pass_lane_start[p][lane] = 4 * (s4 - d);
                }
            }
        } else {
            for (int l = 0; l < nf; ++l) {
                const int m = order[p * 64 + l];
                pass_lane_filter[p][l] = m;
                pass_lane_start[p][l] = 4 * (first[m] / 4);
            }
        }
        s.rounds[p] = 4 * rq;
        s.round_off[p] = s.total_rounds;
        s.total_rounds += s.rounds[p];
    }
    s.w.assign((size_t)s.total_rounds * 64, 0.0f);
    for (int p = 0; p < s.n_pass; ++p) {
        for (int l = 0; l < 64; ++l) {
            const int m = pass_lane_filter[p][l];
            const int st = pass_lane_start[p][l];
            s.filt[p * 64 + l] = m;
            s.start[p * 64 + l] = st;
            if (m < 0) continue;
            // quad j of the pass holds bins st + 4j .. st + 4j + 3:
            //   w[((round_off/4 + j) * 64 + l) * 4 + c]
            for (int k = first[m]; k <= last[m]; ++k) {
                const int rel = k - st;
                s.w[((size_t)(s.round_off[p] / 4 + rel / 4) * 64 + l) * 4 + (rel & 3)] =
                    basis[(size_t)m * n_bins + k];
            }
            s.max_read = std::max(s.max_read, st + s.rounds[p] - 1);
        }
    }
}

namespace {
struct MelJob {
    int filt;    // filter
    int quad;    // first quad, in units of 4 floats of the padded layout
    int slack;   // the job may start up to this many quads earlier
};
struct MelChain {
    int filt, index;       // filter and which of its chains this is
    int first, len;        // jobs[first .. first + len)
};
struct Placement {
    std::vector<int> lane, slot0;   // per chain
    std::vector<int> shift;         // per job
    int conflicts = 0;
    bool ok = false;
};

// Lanes of the four 16-lane groups a ds_read_b128 is serviced in, inverted.
void lane_groups(int (&group_of)[64]) {
    for (int g = 0; g < 4; ++g)
        for (int j = 0; j < 16; ++j) group_of[kB128Groups[g][j]] = g;
}

// One greedy pass: chains in the given order, each to the lane (with room) where its jobs collide
// with the fewest bank-group residues already taken in that lane's group, fullest lane first
// among equals.
Placement place_chains(const std::vector<MelJob>& jobs, const std::vector<MelChain>& chains,
                       const std::vector<int>& order, int n_slots, uint32_t seed) {
    int group_of[64];
    lane_groups(group_of);
    Placement pl;
    pl.lane.assign(chains.size(), -1);
    pl.slot0.assign(chains.size(), 0);
    pl.shift.assign(jobs.size(), 0);
    std::vector<int> fill(64, 0);
    std::vector<char> used((size_t)4 * n_slots * 16, 0);
    std::vector<int> best_shift, cur_shift;
    for (int ci : order) {
        const MelChain& c = chains[ci];
        int best_lane = -1, best_cost = 1 << 30;
        for (int l = 0; l < 64; ++l) {
            if (fill[l] + c.len > n_slots) continue;
            const int g = group_of[l];
            int conf = 0;
            cur_shift.assign(c.len, 0);
            for (int i = 0; i < c.len; ++i) {
                const MelJob& jb = jobs[c.first + i];
                const int s = fill[l] + i;
                int d_ok = -1;
                for (int d = 0; d <= jb.slack && d <= jb.quad; ++d)
                    if (!used[((size_t)g * n_slots + s) * 16 + (jb.quad - d) % 16]) { d_ok = d; break; }
                if (d_ok < 0) { ++conf; d_ok = 0; }
                cur_shift[i] = d_ok;
            }
            // seed != 0: equal candidates in a pseudo-random order
            seed = seed * 1664525u + 1013904223u;
            const int noise = seed ? (int)((seed >> 10) & 7) : g;
            const int cost = conf * 1024 + (n_slots - fill[l]) * 8 + noise;
            if (cost < best_cost) { best_cost = cost; best_lane = l; best_shift = cur_shift; }
        }
        if (best_lane < 0) return pl;   // no room: not ok
        const int g = group_of[best_lane];
        pl.lane[ci] = best_lane;
        pl.slot0[ci] = fill[best_lane];
        for (int i = 0; i < c.len; ++i) {
            const MelJob& jb = jobs[c.first + i];
            pl.shift[c.first + i] = best_shift[i];
            char& u = used[((size_t)g * n_slots + fill[best_lane] + i) * 16 + (jb.quad - best_shift[i]) % 16];
            if (u) ++pl.conflicts;
            u = 1;
        }
        fill[best_lane] += c.len;
    }
    pl.ok = true;
    return pl;
}

// Bank-group collisions of a complete assignment (chains per lane, in slot order); fills in
// slot0 and the per-job shifts it chose.
int count_conflicts(const std::vector<MelJob>& jobs, const std::vector<MelChain>& chains,
                    const std::vector<std::vector<int>>& lane_chains, int n_slots, Placement* pl) {
    int group_of[64];
    lane_groups(group_of);
    std::vector<char> used((size_t)4 * n_slots * 16, 0);
    int conflicts = 0;
    for (int l = 0; l < 64; ++l) {
        const int g = group_of[l];
        int s = 0;
        for (int ci : lane_chains[l]) {
            const MelChain& c = chains[ci];
            if (pl) { pl->lane[ci] = l; pl->slot0[ci] = s; }
            for (int i = 0; i < c.len; ++i, ++s) {
                const MelJob& jb = jobs[c.first + i];
                int d_ok = -1;
                for (int d = 0; d <= jb.slack && d <= jb.quad; ++d)
                    if (!used[((size_t)g * n_slots + s) * 16 + (jb.quad - d) % 16]) { d_ok = d; break; }
                if (d_ok < 0) { ++conflicts; d_ok = 0; }
                used[((size_t)g * n_slots + s) * 16 + (jb.quad - d_ok) % 16] = 1;
                if (pl) pl->shift[c.first + i] = d_ok;
            }
        }
    }
    return conflicts;
}

// Local search from a greedy placement: swap equal-length chains between lanes, swap whole
// lanes, rotate the chains of a lane; keep whatever does not add collisions.
void improve_placement(const std::vector<MelJob>& jobs, const std::vector<MelChain>& chains, int n_slots,
                       Placement& pl) {
    std::vector<std::vector<int>> lanes(64);
    std::vector<int> by_slot(chains.size());
    std::iota(by_slot.begin(), by_slot.end(), 0);
    std::stable_sort(by_slot.begin(), by_slot.end(), [&](int a, int b) { return pl.slot0[a] < pl.slot0[b]; });
    for (int ci : by_slot) lanes[pl.lane[ci]].push_back(ci);
    int cur = count_conflicts(jobs, chains, lanes, n_slots, nullptr);
    uint32_t rng = 2463534242u;
    auto next = [&](uint32_t n) { rng ^= rng << 13; rng ^= rng >> 17; rng ^= rng << 5; return rng % n; };
    for (int it = 0; it < 30000 && cur > 0; ++it) {
        const int la = (int)next(64), lb = (int)next(64);
        const int kind = (int)next(3);
        auto trial = lanes;
        if (kind == 0) {
            if (la == lb || trial[la].empty() || trial[lb].empty()) continue;
            const size_t ia = next((uint32_t)trial[la].size()), ib = next((uint32_t)trial[lb].size());
            if (chains[trial[la][ia]].len != chains[trial[lb][ib]].len) continue;
            std::swap(trial[la][ia], trial[lb][ib]);
        } else if (kind == 1) {
            if (la == lb) continue;
            std::swap(trial[la], trial[lb]);
        } else {
            if (trial[la].size() < 2) continue;
            std::rotate(trial[la].begin(), trial[la].begin() + 1, trial[la].end());
        }
        const int c = count_conflicts(jobs, chains, trial, n_slots, nullptr);
        if (c <= cur) { cur = c; lanes.swap(trial); }
    }
    pl.conflicts = count_conflicts(jobs, chains, lanes, n_slots, &pl);
}
}  // namespace

void build_mel_jobs(const std::vector<float>& basis, int n_mels, int n_bins, int quads, int row_bins,
                    int row_stride, int max_rows, int max_rows_per_group, MelJobs& out) {
    const int J = quads;
    const int pad_quads = (row_stride - row_bins) / 4;
    auto addr = [&](int bin) { return bin + (row_stride - row_bins) * (bin / row_bins); };
    std::vector<int> first(n_mels, 0), last(n_mels, -1), job0(n_mels + 1, 0);
    std::vector<MelJob> jobs;
    for (int m = 0; m < n_mels; ++m) {
        job0[m] = (int)jobs.size();
        int lo = -1, hi = -1;
        for (int k = 0; k < n_bins; ++k)
            if (basis[(size_t)m * n_bins + k] != 0.0f) { if (lo < 0) lo = k; hi = k; }
        if (lo < 0) continue;
        first[m] = lo; last[m] = hi;
        for (int r = lo / row_bins; r <= hi / row_bins; ++r) {
            const int a = std::max(lo, r * row_bins), b = std::min(hi, (r + 1) * row_bins - 1);
            const int t0 = addr(a) / 4, nq = addr(b) / 4 - t0 + 1, nj = (nq + J - 1) / J;
            for (int i = 0; i < nj; ++i) {
                const int slack = (i == nj - 1) ? std::min(J * nj - nq, pad_quads) : 0;
                jobs.push_back(MelJob{m, t0 + J * i, slack});
            }
        }
    }
    job0[n_mels] = (int)jobs.size();
    const int n_groups = (n_mels + 63) / 64;

    // The kernel runs up to four jobs per lane as one batch: never fewer than four slots.
    int n_slots = std::max(4, ((int)jobs.size() + 63) / 64);
    std::vector<MelChain> chains;
    Placement best;
    for (;; ++n_slots) {
        // a filter's jobs form chains of at most n_slots consecutive jobs, one lane each
        chains.clear();
        for (int m = 0; m < n_mels; ++m) {
            const int n = job0[m + 1] - job0[m];
            if (n == 0) continue;
            const int nc = (n + n_slots - 1) / n_slots;
            int at = job0[m];
            for (int c = 0; c < nc; ++c) {
                const int len = n / nc + (c < n % nc ? 1 : 0);
                chains.push_back(MelChain{m, c, at, len});
                at += len;
            }
        }
        // longest chains first; a few deterministic reshuffles of the equal-length runs
        std::vector<int> order(chains.size());
        std::iota(order.begin(), order.end(), 0);
        std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return chains[a].len > chains[b].len; });
        best = Placement();
        uint32_t rng = 12345u;
        for (int attempt = 0; attempt < 16; ++attempt) {
            if (attempt > 0) {
                for (size_t i = 0; i + 1 < order.size(); ++i) {
                    rng = rng * 1664525u + 1013904223u;
                    size_t hi = i;
                    while (hi + 1 < order.size() && chains[order[hi + 1]].len == chains[order[i]].len) ++hi;
                    const size_t j = i + (rng >> 8) % (hi - i + 1);
                    std::swap(order[i], order[j]);
                }
            }
            Placement pl = place_chains(jobs, chains, order, n_slots, attempt == 0 ? 0u : rng | 1u);
            if (pl.ok && (!best.ok || pl.conflicts < best.conflicts)) best = pl;
            if (best.ok && best.conflicts == 0) break;
        }
        if (best.ok) break;
    }
    if (best.conflicts > 0) improve_placement(jobs, chains, n_slots, best);

    out.n_jobs = n_slots;
    out.quads = J;
    out.at.assign((size_t)n_slots * 64, 0);
    out.cell.assign((size_t)n_slots * 64, 0);
    out.cont.assign((size_t)n_slots * 64, 0);
    out.w.assign((size_t)n_slots * J * 64 * 4, 0.0f);
    out.max_read = 0;
    out.bank_conflicts = best.conflicts;
    // cell rows: one per chain of the filter with the most chains in each group of 64 filters
    std::vector<int> n_chains(n_mels, 0);
    for (const MelChain& c : chains) n_chains[c.filt] = std::max(n_chains[c.filt], c.index + 1);
    out.n_rows = 0;
    for (int p = 0; p < 4; ++p) {
        out.rows[p] = 0;
        if (p >= n_groups) continue;
        for (int f = 64 * p; f < std::min(n_mels, 64 * p + 64); ++f) out.rows[p] = std::max(out.rows[p], n_chains[f]);
        out.rows[p] = std::max(out.rows[p], 1);
        out.n_rows += out.rows[p];
    }
    out.shared = out.n_rows > max_rows;
    for (int p = 0; p < 4; ++p) out.shared = out.shared || out.rows[p] > max_rows_per_group;
    if (out.shared) {
        out.n_rows = n_groups;
        for (int p = 0; p < n_groups; ++p) out.rows[p] = 1;
    }
    std::vector<int> row0(5, 0);
    for (int p = 0; p < 4; ++p) row0[p + 1] = row0[p] + out.rows[p];
    // rows some filter of the group leaves unwritten must be cleared before every frame
    out.zero_rows = 0;
    for (int p = 0; p < n_groups; ++p)
        for (int c = 0; c < out.rows[p]; ++c) {
            bool hole = out.shared;
            for (int f = 64 * p; f < std::min(n_mels, 64 * p + 64); ++f) hole = hole || n_chains[f] <= c;
            if (hole) out.zero_rows |= 1u << (row0[p] + c);
        }
    for (int s = 0; s < n_slots; ++s)
        for (int l = 0; l < 64; ++l) out.cell[s * 64 + l] = out.n_rows * 64 + l;   // idle / mid-chain

    std::vector<char> done((size_t)n_mels * n_bins, 0);
    for (size_t ci = 0; ci < chains.size(); ++ci) {      // in filter order, a filter's jobs in sequence
        const MelChain& c = chains[ci];
        const int l = best.lane[ci];
        for (int i = 0; i < c.len; ++i) {
            const MelJob& jb = jobs[c.first + i];
            const int s = best.slot0[ci] + i;
            const int at = 4 * (jb.quad - best.shift[c.first + i]);
            out.at[s * 64 + l] = at;
            out.cont[s * 64 + l] = i > 0;
            if (i == c.len - 1)
                out.cell[s * 64 + l] = (row0[c.filt / 64] + (out.shared ? 0 : c.index)) * 64 + (c.filt & 63);
            out.max_read = std::max(out.max_read, at + 4 * J - 1);
            for (int k = first[c.filt]; k <= last[c.filt]; ++k) {
                const int rel = addr(k) - at;
                if (rel < 0 || rel >= 4 * J || done[(size_t)c.filt * n_bins + k]) continue;
                done[(size_t)c.filt * n_bins + k] = 1;
                out.w[((size_t)(s * J + rel / 4) * 64 + l) * 4 + (rel & 3)] = basis[(size_t)c.filt * n_bins + k];
            }
        }
    }
}

}  // namespace mfcc_host
