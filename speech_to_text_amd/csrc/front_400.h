// Front end for n_fft = 16 A with any A from 8 to 63 that is not a power of two (every multiple
// of 16 from 128 to 1008 but 256 and 512: the 8 .. 60 ms
// windows of 16 kHz speech front ends, or shorter ones at 32 / 48 kHz, taken as the transform
// length, e.g. n_fft = 400, hop 160, 80 mels) on gfx950: one wavefront per four consecutive
// frames, no barriers in the frame loop.
//
//   pad / frame / window -> N-point real DFT (A x 16) -> |X|^power -> sparse mel filterbank
//
// Same role as front_2048.h (the stft + abs**power + mel einsum part of librosa.feature.mfcc,
// /root/reference/src/preprocess.py:134) for these n_fft.
//
// Frame x[n], n = 16 a + b (a < A, b < 16); bin k = c + A d (c < A, d < 16):
//     W_N^(n k) = W_A^(a c) W_N^(b c) W_16^(b d)
//  step 1  lane (t = lane>>4, b = lane&15) holds the A samples a = 0..A-1 of frame t and computes
//          the real A-point DFT over a for c = 0..A/2 in registers (fc::rdft25: 5 x 5;
//          fc::rdft_even / fc::rdft_odd: folded direct sums)
//  twiddle Z[b][c] = Y[b][c] W_N^(b c)
//  exchange through LDS: column (t, c) = 16 complex, columns 36 floats apart (conflict-free
//          16-byte reads)
//  step 2  lane t NC + c (NC = A/2 + 1 columns per frame; beyond 64 columns, at A = 40, a lane
//          takes a second column, lane + 64) reads its column and does the 16-point
//          DFT over b in registers -> X[c + A d]; d >= 8 are the mirror images N - k of the
//          columns c > A/2 (for even A the middle column c = A/2 mirrors onto itself)
//  mel     as in front_pow2.h: lane = (frame, filter) pair, 16 filters per round
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "front_pow2.h"

#ifndef MFCC_16A_WIN_LDS
#define MFCC_16A_WIN_LDS (~0ULL << 30)
#endif

// The lengths that are built, as A = n_fft / 16.  Wavefronts per workgroup follow the region size
// and the registers the A-point transform needs: 16 up to A = 25 and at 30, 12 up to A = 40 (12 KB
// regions), 8 beyond, 6 from A = 61 on (18 KB regions).  A second, smaller count takes over when a large
// filterbank leaves less LDS: 8 for the lengths that were built first, 4 for the longest three.
#define MFCC_16A_LENGTHS(X) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15) X(17) X(18) X(19) X(20) X(21) X(22) X(23) X(24) \
    X(25) X(26) X(27) X(28) X(29) X(30) X(31) X(33) X(34) X(35) X(36) X(37) X(38) X(39) X(40) X(41) X(42) X(43) X(44) X(45) \
    X(46) X(47) X(48) X(49) X(50) X(51) X(52) X(53) X(54) X(55) X(56) X(57) X(58) X(59) X(60) X(61) X(62) X(63)
constexpr int k16a_waves(int A) { return (A <= 25 || A == 30) ? 16 : (A <= 40 ? 12 : (A <= 60 ? 8 : 6)); }
constexpr int k16a_fallback_waves(int A) { return (A == 20 || A == 25 || A == 30 || A == 40) ? 8 : (A >= 57 ? 4 : 0); }
// The kernels are instantiated in front_400_inst.hip, which is compiled MFCC_16A_PARTS times (the
// lengths dealt round-robin over the parts) so that the build runs in parallel; part k exports
// k16a_lookup_part<k>(n_fft, waves, power == 2, 16-bit PCM) -> kernel, or nullptr for a length or
// wavefront count that is not its own.
#define MFCC_16A_PARTS 6
constexpr int k16a_part(int A) {
    int i = 0;
#define MFCC_16A_INDEX(B) if (B == A) return i % MFCC_16A_PARTS; ++i;
    MFCC_16A_LENGTHS(MFCC_16A_INDEX)
#undef MFCC_16A_INDEX
    return -1;
}
typedef void (*Pow2Fn)(const Pow2Args);
#define MFCC_16A_FOR_EACH_PART(X) X(0) X(1) X(2) X(3) X(4) X(5)
#define MFCC_16A_DECLARE_PART(K) Pow2Fn k16a_lookup_part##K(int n_fft, int waves, bool pw2, bool pcm16);
MFCC_16A_FOR_EACH_PART(MFCC_16A_DECLARE_PART)
#undef MFCC_16A_DECLARE_PART

namespace k16a {
constexpr int T = 4;                    // frames per wavefront pass
constexpr int COL = 36;                 // floats per exchange column: 16 complex + 4 floats of padding
template <int A>
struct Geo {
    static_assert(A >= 8 && A <= 63, "supported: n_fft = 16 A for A in 8 .. 63");
    static constexpr int N = 16 * A, NB = N / 2 + 1, NA = A, NC = A / 2 + 1;   // NC columns per frame
    static constexpr int ROUNDS = (T * NC + 63) / 64;                          // step-2 columns per lane
    static_assert(ROUNDS <= 2, "at most two columns per lane");
    // frame t's power spectrum starts at float P_STRIDE t: the bases are 16 floats apart modulo
    // 64, so the T NC lanes of a store (NC <= 16 consecutive bins per frame) hit different banks
    static constexpr int P_STRIDE = (NB + 8 + 47) / 64 * 64 + 16;
    static constexpr int TW_FLOATS = (NC - 1) * 16 * 2;                          // W_N^(b c), c = 1..NC-1
    static constexpr int REGION_MIN = T * NC * COL > T * P_STRIDE ? T * NC * COL : T * P_STRIDE;
    static constexpr int REGION = REGION_MIN > k2048::XCH_FLOATS ? (REGION_MIN + 63) / 64 * 64 : k2048::XCH_FLOATS;
    static constexpr int p_base(int t) { return P_STRIDE * t; }
    // The window: A registers per lane, or (bit A of MFCC_16A_WIN_LDS) a table in LDS as [b][a],
    // rows WIN_ROW floats apart -- a multiple of 4 whose quarter is odd, so that the 16-byte reads
    // of the 16 residues b fall into 16 different bank groups.  Default: LDS from A = 30 on, where
    // the window's registers are what the kernel would spill.
    static constexpr bool WIN_LDS = ((unsigned long long)(MFCC_16A_WIN_LDS) >> A) & 1;
    static constexpr int WIN_ROW = ((A + 3) / 4) % 2 ? (A + 3) / 4 * 4 : (A + 3) / 4 * 4 + 4;
    static constexpr int WIN_FLOATS = WIN_LDS ? 16 * WIN_ROW : 0;
    static constexpr size_t lds_bytes(int waves, int mel_quads, int mel_rounds) {
        return sizeof(float) * ((size_t)waves * REGION + TW_FLOATS + WIN_FLOATS + 4 * (size_t)mel_quads * (64 / T) +
                                2 * 64 * (size_t)mel_rounds + 4 /* unit counter */);
    }
};
static_assert(Geo<25>::P_STRIDE == 272 && Geo<25>::REGION == k2048::XCH_FLOATS, "n_fft = 400 layout");

template <int A>
__device__ __forceinline__ void rdft_a(const float (&x)[A], float (&yr)[A / 2 + 1], float (&yi)[A / 2 + 1]) {
    if constexpr (A == 25) fc::rdft25(x, yr, yi);
    else if constexpr (A % 2 == 0) fc::rdft_even<A>(x, yr, yi);
    else fc::rdft_odd<A>(x, yr, yi);
}
}  // namespace k16a

// Pow2Args with tw1 = W_N^(b c) as [c - 1][b] ((NC - 1) x 16); w64 and tw128 are unused.
template <int A, int WAVES, int PW, bool PCM16>
__global__ __launch_bounds__(WAVES * 64) void k_melspec_16a(const Pow2Args a) {
    using namespace k2048;
    using namespace k16a;
    using G = Geo<A>;
    constexpr int N = G::N, NB = G::NB, NA = G::NA, NC = G::NC, REGION = G::REGION, TW_FLOATS = G::TW_FLOATS;
    constexpr int THREADS = WAVES * 64;
    constexpr int FPR = 64 / T;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float* X = lds + wave * REGION;
    float2* X2 = reinterpret_cast<float2*>(X);
    float* tab = lds + WAVES * REGION;
    float2* tw_l = reinterpret_cast<float2*>(tab);
    tab += TW_FLOATS;
    float* win_l = tab;
    tab += G::WIN_FLOATS;
    float4* melw_l = reinterpret_cast<float4*>(tab);
    uint2* pairs_l = reinterpret_cast<uint2*>(melw_l + a.mel_quads * FPR);
    int* run_ctr = reinterpret_cast<int*>(pairs_l + a.mel_rounds * 64);

    for (int i = tid; i < WAVES * REGION; i += THREADS) lds[i] = 0.0f;
    for (int i = tid; i < TW_FLOATS / 2; i += THREADS) tw_l[i] = a.tw1[i];
    for (int i = tid; i < G::WIN_FLOATS; i += THREADS) {
        const int wb = i / G::WIN_ROW, wa = i - wb * G::WIN_ROW;
        win_l[i] = wa < NA ? a.window[16 * wa + wb] : 0.0f;
    }
    for (int i = tid; i < a.mel_quads * FPR; i += THREADS) melw_l[i] = a.melw[i];
    for (int i = tid; i < a.mel_rounds * 64; i += THREADS) pairs_l[i] = a.mel_pairs[i];
    if (tid == 0) *run_ctr = 0;
    __syncthreads();

    const int b = lane & 15, tf = lane >> 4;        // step 1: residue and frame of this lane
    // step 2: round r gives this lane column lane + 64 r = t2 NC + c2 (lanes past T NC idle)
    constexpr int ROUNDS = G::ROUNDS;
    int c2[ROUNDS], pb[ROUNDS];
    bool active2[ROUNDS];
    const float* xr[ROUNDS];
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        const int col = min(lane + 64 * r, T * NC - 1);
        const int t2 = col / NC;
        c2[r] = col - t2 * NC;
        pb[r] = G::p_base(t2);
        active2[r] = lane + 64 * r < T * NC;
        xr[r] = X + col * COL;
    }
    float win[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) win[i] = G::WIN_LDS ? 0.0f : a.window[16 * i + b];
    const float4* win_q = reinterpret_cast<const float4*>(win_l + b * G::WIN_ROW);
    const int xw = (tf * NC * COL) / 2 + b;         // complex index of Z[b][0] of frame tf

    float runmax = 0.0f;
    int cur_grp = -1;
    const int n_tiles = a.hdr->n_tiles;
    // Work unit = 4 consecutive frames of a tile; the workgroup's wavefronts take the units of its
    // contiguous run one at a time from a counter in LDS (see front_pow2.h).
    const int64_t n_units = (int64_t)n_tiles * (TILE_FRAMES / T);
    const int64_t per = (n_units + gridDim.x - 1) / gridDim.x;
    const int64_t u0 = (int64_t)blockIdx.x * per;
    const int count = u0 < n_units ? (int)min(per, n_units - u0) : 0;
    auto request_unit = [&]() -> int {
        int v = 0;
        if (lane == 0) v = atomicAdd(run_ctr, 1);
        return v;
    };
    int tix = -2;
    Tile ti = {}, ti_next = {};

    int next_v = 0;
    for (int cur = __builtin_amdgcn_readfirstlane(request_unit()); cur < count;
         cur = __builtin_amdgcn_readfirstlane(next_v)) {
        next_v = request_unit();
        const int64_t u = u0 + cur;
        const int tidx = (int)(u >> 2), sub = (int)(u & 3);
        if (tidx != tix) {
            ti = (tidx == tix + 1) ? ti_next : load_tile(a.tiles, tidx);
            tix = tidx;
            ti_next = load_tile(a.tiles, min(tix + 1, n_tiles - 1));
        }
        const Tile tc = ti;
        const int fb = tc.f0 + sub * T;
        const int nvalid = min(T, tc.F - fb);
        if (nvalid <= 0) continue;
        if (tc.grp != cur_grp) {
            flush_max(runmax, cur_grp, a.gmax, lane);
            cur_grp = tc.grp;
            runmax = 0.0f;
        }
        const int64_t row0 = tc.row0 + (fb - tc.f0);

        // ---- samples, window ----------------------------------------------------------------------
        float x[NA];
        const int first = fb * a.hop - a.pad;                 // start of the pass's first frame
        const bool all_in = nvalid == T && first >= 0 && first + (T - 1) * a.hop + N <= tc.T;
        if (all_in) {
            const int64_t s0 = tc.soff + first + tf * a.hop + b;
            if (PCM16) {
                const int16_t* pp = static_cast<const int16_t*>(a.pcm) + s0;
#pragma unroll
                for (int i = 0; i < NA; ++i) x[i] = PCM16_SCALE * (float)pp[16 * i];
            } else {
                const float* pp = static_cast<const float*>(a.pcm) + s0;
#pragma unroll
                for (int i = 0; i < NA; ++i) x[i] = pp[16 * i];
            }
        } else if (!(a.reflect && a.pad) && tc.T > 0) {
            // a pass with edge frames or fewer than four frames, under zero padding: the same loads
            // from addresses clamped into the utterance; lanes that fell outside it, and the lanes
            // of frames past the utterance's last, take 0
            const int f0 = first + tf * a.hop + b, last = tc.T - 1;
            if (PCM16) {
                const int16_t* pp = static_cast<const int16_t*>(a.pcm) + tc.soff;
#pragma unroll
                for (int i = 0; i < NA; ++i) x[i] = PCM16_SCALE * (float)pp[min(max(f0 + 16 * i, 0), last)];
            } else {
                const float* pp = static_cast<const float*>(a.pcm) + tc.soff;
#pragma unroll
                for (int i = 0; i < NA; ++i) x[i] = pp[min(max(f0 + 16 * i, 0), last)];
            }
            const bool live = tf < nvalid;
#pragma unroll
            for (int i = 0; i < NA; ++i) x[i] = (live && (unsigned)(f0 + 16 * i) <= (unsigned)last) ? x[i] : 0.0f;
        } else {
            // such a pass under reflect padding: frame by frame through LDS
#pragma unroll
            for (int i = 0; i < NA; ++i) x[i] = 0.0f;
#pragma unroll 1
            for (int t = 0; t < nvalid; ++t) {
                kp2::stage_edge_rows(X, a.pcm, tc.soff, PCM16, first + t * a.hop, tc.T, a.reflect && a.pad, lane, (N + 63) / 64, a.hop, N);
                if (tf == t) {
#pragma unroll
                    for (int i = 0; i < NA; ++i) x[i] = X[16 * i + b];
                }
            }
        }
        if (G::WIN_LDS) {
#pragma unroll
            for (int j = 0; j < (NA + 3) / 4; ++j) {
                const float4 w = win_q[j];
                x[4 * j] *= w.x;
                if (4 * j + 1 < NA) x[4 * j + 1] *= w.y;
                if (4 * j + 2 < NA) x[4 * j + 2] *= w.z;
                if (4 * j + 3 < NA) x[4 * j + 3] *= w.w;
            }
        } else {
#pragma unroll
            for (int i = 0; i < NA; ++i) x[i] *= win[i];
        }

        // ---- step 1: real DFT-25 over a, twiddle, exchange -----------------------------------------
        {
            float yr[NC], yi[NC];
            rdft_a<A>(x, yr, yi);
            float2 tw[NC];
#pragma unroll
            for (int c = 1; c < NC; ++c) tw[c] = tw_l[(c - 1) * 16 + b];
            X2[xw] = make_float2(yr[0], 0.0f);
#pragma unroll
            for (int c = 1; c < NC; ++c) {
                fc::cmul(yr[c], yi[c], tw[c].x, tw[c].y);
                X2[xw + c * (COL / 2)] = make_float2(yr[c], yi[c]);
            }
        }

        // ---- step 2: DFT-16 over b ---------------------------------------------------------------
        // (every column is read before the first power spectrum is stored: the two share the region)
        float dr[ROUNDS][16], di[ROUNDS][16];
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float4 v = *reinterpret_cast<const float4*>(xr[r] + 4 * j);
                dr[r][2 * j] = v.x; di[r][2 * j] = v.y; dr[r][2 * j + 1] = v.z; di[r][2 * j + 1] = v.w;
            }
        }

        // ---- power spectra -> LDS (natural bin order per frame) --------------------------------------
        // bin c2 + A d for d < 8; d >= 8 is the mirror image A (16 - d) - c2 (for c2 = 0 only
        // d = 8, bin N/2: the others repeat bins A d; for even A the column c2 = A/2 mirrors onto
        // its own bins, so its d >= 8 are skipped)
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r) {
            fc::cfft16(dr[r], di[r]);
            if (active2[r]) {
                const int c = c2[r];
                const bool mirrors = c != 0 && (A % 2 != 0 || c != A / 2);
#pragma unroll
                for (int d = 0; d < 16; ++d) {
                    const float re = dr[r][fc::cfft16_pos(d)], im = di[r][fc::cfft16_pos(d)];
                    const float pw = apply_power<PW>(re * re + im * im, a.pw_mode, a.half_power);
                    if (d < 8) X[pb[r] + c + A * d] = pw;
                    else if ((d == 8 && c == 0) || mirrors) X[pb[r] + A * (16 - d) - c] = pw;
                }
            }
        }

        if (a.spec_out) {
#pragma unroll
            for (int t = 0; t < T; ++t) {
                if (t < nvalid) {
                    float* so = a.spec_out + (row0 + t) * NB;
                    for (int k = lane; k < NB; k += 64) so[k] = X[G::p_base(t) + k];
                }
            }
        }

        // ---- sparse mel filterbank: lane = (frame, filter) pair (see front_pow2.h) --------------------
        if (a.mel_out) {
            float* mo = a.mel_out + row0 * a.n_mels;
            const int n_out = nvalid * a.n_mels;
            for (int rd = 0; rd < a.mel_rounds; ++rd) {
                const uint2 e = pairs_l[rd * 64 + lane];
                const int nq = __builtin_amdgcn_readfirstlane(e.y >> 16);
                const float4* pp = reinterpret_cast<const float4*>(X + (e.x & 0xffffu));
                const float4* wp = melw_l + (e.x >> 16);
                float acc0 = 0.0f, acc1 = 0.0f;
#pragma unroll 2
                for (int i = 0; i < nq; ++i) {
                    const float4 w = wp[i * FPR];
                    const float4 v = pp[i];
                    acc0 = fmaf(w.x, v.x, acc0);
                    acc1 = fmaf(w.y, v.y, acc1);
                    acc0 = fmaf(w.z, v.z, acc0);
                    acc1 = fmaf(w.w, v.w, acc1);
                }
                const float acc = acc0 + acc1;
                const int oi = (int)(e.y & 0xffffu);
                if (oi < n_out) {
                    mo[oi] = acc;
                    runmax = fmaxf(runmax, acc);
                }
            }
        }
    }
    flush_max(runmax, cur_grp, a.gmax, lane);
}
