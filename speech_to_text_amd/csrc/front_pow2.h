// Power-of-two front end for gfx950 (n_fft = 256, 512 or 1024): one wavefront per group of
// 2048 / n_fft consecutive frames, no barriers in the frame loop.
//
//   pad / frame / window -> n_fft-point real FFT -> |X|^power -> sparse mel filterbank
//
// Same role as front_2048.h (the stft + abs**power + mel einsum part of librosa.feature.mfcc,
// /root/reference/src/preprocess.py:134) for the short transforms of 16 kHz speech front ends.
//
// With R = n_fft / 128 and T = 16 / R, a pass handles T frames.  Frame a[n], n = n1 + 64 n2: lane
// n1 holds the 2R samples n2 = 0 .. 2R-1 of each of the T frames (32 registers in all).
//  step 1  real 2R-point DFT over n2 in registers -> A[n1][k2], k2 = 0 .. R
//  twiddle B[n1][k2] = A[n1][k2] W_N^(n1 k2), k2 = 1 .. R-1
//  exchange through LDS: slot t R + k2 holds column k2 of frame t (64 complex); slot t R carries
//          the two real columns k2 = 0 and R as the even/odd-folded real sequence of a 128-point
//          real DFT.  16 slots, the layout of front_2048.h.
//  step 2  lane (slot = lane>>2, q = lane&3): 4-point DFT across the quad with DPP, twiddle,
//          16-point DFT in registers -> X[k2 + 2R (4 ka + kb)]; ka >= 8 are mirror images
//  the special slots are post-processed as 128-point real DFTs -> bins R j, j = 0 .. 64
//  mel     lane = (frame, filter) pair: a dot product of the filter's span of the frame's power
//          spectrum (in LDS) with its weights; 64 / T filters per round
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "front_2048.h"

// 1: at eight columns per frame (n_fft = 1024) the window sits in LDS as well (measured: 1.74 ms
// per 10 h with window and step-1 twiddles in registers and 11 spilled registers, 1.69 with the
// twiddles in LDS, 1.655 with both there and no spills)
#ifndef MFCC_P2_WIN_LDS
#define MFCC_P2_WIN_LDS 1
#endif

namespace kp2 {

constexpr int SPEC_STRIDE = 136;   // floats between the frames' special-column scratch areas

// After step 2 the wavefront's LDS region holds the T power spectra (frame t from float
// p_base(t); the bases differ modulo 64 so that the 64 lanes of a power-spectrum store hit
// different banks where the geometry allows it), then the special-column scratch.
template <int LOG2N>
__host__ __device__ constexpr int p_base(int t) {
    return LOG2N == 9    ? (t == 0 ? 0 : t == 1 ? 388 : t == 2 ? 800 : 1188)
           : LOG2N == 10 ? (t == 0 ? 0 : 776)
                         : 160 * t + 16 * ((t >> 1) & 1);
}
template <int LOG2N>
struct Geo {
    static constexpr int N = 1 << LOG2N;
    static constexpr int R = N / 128;          // complex columns per frame (column 0: the two real ones)
    static constexpr int T = 16 / R;           // frames per pass
    static constexpr int ROWS = 2 * R;         // samples per lane and frame
    static constexpr int NB = N / 2 + 1;
    // step-1 twiddles: registers up to four columns, LDS beyond (at eight columns the 14
    // registers they would take are what the kernel spills)
    static constexpr int TW_LDS_FLOATS = R >= 8 ? (R - 1) * 64 * 2 + MFCC_P2_WIN_LDS * N : 0;
    static constexpr bool WIN_LDS = R >= 8 && MFCC_P2_WIN_LDS;
    static constexpr int SCR_OFF = LOG2N == 9 ? 1504 : LOG2N == 10 ? 1776 : 1296;
    static constexpr int REGION = (SCR_OFF + T * SPEC_STRIDE + 63) / 64 * 64 < k2048::XCH_FLOATS
                                      ? k2048::XCH_FLOATS
                                      : (SCR_OFF + T * SPEC_STRIDE + 63) / 64 * 64;
    // floats a mel read may cover from the start of a frame's power spectrum
    static constexpr int row_cap() {
        int cap = SCR_OFF - p_base<LOG2N>(T - 1);
        for (int t = 0; t + 1 < T; ++t) {
            const int c = p_base<LOG2N>(t + 1) - p_base<LOG2N>(t);
            if (c < cap) cap = c;
        }
        return cap;
    }
};
static_assert(Geo<9>::REGION == 2048 && Geo<10>::REGION == 2048, "exchange-sized regions expected");

// mel_quads: weight quads of all rounds together; fpr: filters per round
constexpr size_t lds_bytes(int region, int waves, int fpr, int mel_quads, int mel_rounds, int tw_floats = 0) {
    return sizeof(float) * ((size_t)waves * region + k2048::W64_FLOATS + tw_floats + 4 * (size_t)mel_quads * fpr +
                            2 * 64 * (size_t)mel_rounds + 4 /* unit counter */);
}

// hop, n_fft: of the transform (for the slice librosa reflects a tail frame in; `start` is the
// frame's first sample, t hop - n_fft / 2)
__device__ __noinline__ void stage_edge_rows(float* X, const void* pcm, int64_t soff, bool pcm16, int start, int T,
                                             int reflect, int lane, int rows, int hop, int n_fft) {
    const int lo = reflect ? k2048::reflect_slice_start(start, T, hop, n_fft) : 0;
#pragma unroll 1
    for (int n2 = 0; n2 < rows; ++n2) {
        const int idx = start + lane + 64 * n2;
        float v = 0.0f;
        if (reflect) {
            if (T > 0) v = k2048::load_sample(pcm, soff + k2048::reflect_index(idx, T, lo), pcm16);
        } else if (idx >= 0 && idx < T) {
            v = k2048::load_sample(pcm, soff + idx, pcm16);
        }
        X[lane + 64 * n2] = v;
    }
}

}  // namespace kp2

struct Pow2Args {
    const void* pcm;           // float32 samples, or int16 PCM (value / 32768) when pcm16 is set
    const Tile* tiles;
    const WsHeader* hdr;
    const float* window;       // n_fft
    const float2* tw1;         // (R - 1) x 64: W_N^(n1 k2), scaled for rdft_cols
    const float2* w64;         // 4 x 16
    const float2* tw128;       // 64
    const float4* melw;        // mel_quads x (64 / T) quads of weights: round by round, quad by quad,
                               // one column per filter of the round
    const uint2* mel_pairs;    // mel_rounds x 64, per lane: .x = float offset of the pair's first
                               // quad in the wavefront's region | index of its first weight quad
                               // << 16; .y = output index frame * n_mels + filter (0xffff: idle) |
                               // quads of the round << 16
    float* mel_out;
    float* spec_out;
    uint32_t* gmax;
    int32_t n_mels;
    int32_t mel_quads;         // all rounds together
    int32_t mel_rounds;
    int32_t hop;
    int32_t pad;
    int32_t reflect;
    int32_t pw_mode;
    float half_power;
};

template <int LOG2N, int WAVES, int PW, bool PCM16>
__global__ __launch_bounds__(WAVES * 64) void k_melspec_pow2(const Pow2Args a) {
    using namespace k2048;
    using G = kp2::Geo<LOG2N>;
    constexpr int N = G::N, R = G::R, T = G::T, ROWS = G::ROWS, NB = G::NB, REGION = G::REGION;
    constexpr int THREADS = WAVES * 64;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float* X = lds + wave * REGION;
    float2* X2 = reinterpret_cast<float2*>(X);
    float* tab = lds + WAVES * REGION;
    float2* w64_l = reinterpret_cast<float2*>(tab);
    tab += W64_FLOATS;
    constexpr bool TW_LDS = G::TW_LDS_FLOATS > 0;
    float2* tw_l = reinterpret_cast<float2*>(tab);
    tab += G::TW_LDS_FLOATS;
    float4* melw_l = reinterpret_cast<float4*>(tab);
    constexpr int FPR = 64 / T;                     // filters per mel round
    uint2* pairs_l = reinterpret_cast<uint2*>(melw_l + a.mel_quads * FPR);
    int* run_ctr = reinterpret_cast<int*>(pairs_l + a.mel_rounds * 64);

    for (int i = tid; i < WAVES * REGION; i += THREADS) lds[i] = 0.0f;
    // W_64^(m kb), m = 1..15, at slot m - 1 of row kb (m = 0 is 1 and never read): the eight
    // 16-byte reads of a row then carry only used values and stay ds_read_b128
    if (tid < 64) {
        const int m = tid & 15;
        w64_l[(tid >> 4) * W64_STRIDE + (m ? m - 1 : 15)] = m ? a.w64[tid] : make_float2(0.0f, 0.0f);
    }
    float4* win_l = reinterpret_cast<float4*>(tw_l + (R - 1) * 64);   // quads (j, j+1, j+2, j+3) of row group j / 4
    if (TW_LDS) {
        for (int i = tid; i < (R - 1) * 64; i += THREADS) tw_l[i] = a.tw1[i];
        if (G::WIN_LDS) {
            for (int i = tid; i < N / 4; i += THREADS) {
                const int g = i >> 6, l = i & 63;
                win_l[i] = make_float4(a.window[l + 64 * (4 * g)], a.window[l + 64 * (4 * g + 1)],
                                       a.window[l + 64 * (4 * g + 2)], a.window[l + 64 * (4 * g + 3)]);
            }
        }
    }
    for (int i = tid; i < a.mel_quads * FPR; i += THREADS) melw_l[i] = a.melw[i];
    for (int i = tid; i < a.mel_rounds * 64; i += THREADS) pairs_l[i] = a.mel_pairs[i];
    if (tid == 0) *run_ctr = 0;
    __syncthreads();

    const int q = lane & 3, slot = lane >> 2;
    const int kb = ((q & 1) << 1) | (q >> 1);       // the quad DFT leaves output kb in lane q
    const int k2 = slot & (R - 1), tf = slot / R;   // step 2: column and frame of this lane
    float win[ROWS];
#pragma unroll
    for (int j = 0; j < ROWS; ++j) win[j] = G::WIN_LDS ? 0.0f : a.window[lane + 64 * j];
    float2 tw[R];
#pragma unroll
    for (int k = 1; k < R; ++k) tw[k] = TW_LDS ? make_float2(0.0f, 0.0f) : a.tw1[(k - 1) * 64 + lane];
    const float2 tw128 = a.tw128[lane];

    // Quad DFT-4 signs (see front_2048.h): lanes 1..3 of a quad end up with negated outputs, which
    // |X|^2 does not see; the special slots pair outputs of equally negated lanes.
    const float sgn1 = (q & 2) ? -1.0f : 1.0f;
    const float sgn2 = (q == 1 || q == 2) ? -1.0f : 1.0f;
    const bool lane3 = (q == 3);
    // step-1 write of element `lane` of slot s: complex index s * COL_STRIDE + xw[s & 3]
    const int xw[4] = {xch_slot(0, lane), xch_slot(1, lane) - COL_STRIDE, xch_slot(2, lane) - 2 * COL_STRIDE,
                       xch_slot(3, lane) - 3 * COL_STRIDE};
    // special slot s as 128 reals y[n] = component n & 1 of complex n >> 1: y[lane], y[64 + lane]
    int y0[4], y1[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        y0[i] = 2 * (xch_slot(i, lane >> 1) - i * COL_STRIDE) + (lane & 1);
        y1[i] = 2 * (xch_slot(i, 32 + (lane >> 1)) - i * COL_STRIDE) + (lane & 1);
    }
    // step-2 read of pairs 8 q + j, j = 0..7, of the lane's slot: float index xr_base ^ (4 j)
    const int xr_base = 2 * (slot * COL_STRIDE + 16 * q) + 4 * (2 * (slot & 3) + (q >> 1));
    // power spectrum: ka < 8 -> bin k2 + 2R kb + 8R ka; ka >= 8 -> the mirror image
    // 2R (4 - kb) - k2 + 8R (15 - ka)
    int pbt = 0;
#pragma unroll
    for (int t = 0; t < T; ++t) pbt = (tf == t) ? kp2::p_base<LOG2N>(t) : pbt;
    const int pb0 = pbt + k2 + 2 * R * kb;
    const int pb1 = pbt + 2 * R * (4 - kb) - k2;
    const bool special = (k2 == 0);
    float2* s_wr = X2 + G::SCR_OFF / 2 + tf * (kp2::SPEC_STRIDE / 2) + kb;
    // R == 4 (measured a gain there only): the row of 16 lanes whose first quad is a special slot writes that slot's outputs,
    // lane 4 g + q of the row the outputs j = 4 (4 i + g) + kb, i = 0..3
    const bool row_special = ((lane >> 4) * 4) % R == 0;
    float2* s_wr4 = X2 + G::SCR_OFF / 2 + (((lane >> 4) * 4) / R) * (kp2::SPEC_STRIDE / 2) + 4 * ((lane >> 2) & 3) + kb;
    const int s_rd0 = G::SCR_OFF / 2 + lane, s_rd1 = G::SCR_OFF / 2 + ((64 - lane) & 63);

    float runmax = 0.0f;
    int cur_grp = -1;
    const int n_tiles = a.hdr->n_tiles;
    // Work unit = T consecutive frames of a tile (R units per 16-frame tile).  Each workgroup owns
    // a contiguous run of units and its wavefronts take them one at a time from a counter in LDS
    // (the next one is requested a pass ahead): at any moment the CU works on neighbouring frames,
    // which share most of their samples, and its wavefronts finish together.
    const int64_t n_units = (int64_t)n_tiles * R;
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
        const int tidx = (int)(u / R), sub = (int)(u % R);
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
        float r[T * ROWS];
#pragma unroll
        for (int t = 0; t < T; ++t) {
            float* rt = r + t * ROWS;
            const int start = (fb + t) * a.hop - a.pad;
            if (t >= nvalid) {
#pragma unroll
                for (int j = 0; j < ROWS; ++j) rt[j] = 0.0f;
            } else if (start >= 0 && start + N <= tc.T) {
                const int64_t s0 = tc.soff + start + lane;
                if (PCM16) {
                    const int16_t* pp = static_cast<const int16_t*>(a.pcm) + s0;
#pragma unroll
                    for (int j = 0; j < ROWS; ++j) rt[j] = PCM16_SCALE * (float)pp[64 * j];
                } else {
                    const float* pp = static_cast<const float*>(a.pcm) + s0;
#pragma unroll
                    for (int j = 0; j < ROWS; ++j) rt[j] = pp[64 * j];
                }
            } else {
                // edge frames (overlapping the padding or the utterance's end) come through LDS
                kp2::stage_edge_rows(X, a.pcm, tc.soff, PCM16, start, tc.T, a.reflect && a.pad, lane, ROWS, a.hop, N);
#pragma unroll
                for (int j = 0; j < ROWS; ++j) rt[j] = X[lane + 64 * j];
            }
        }
        if (G::WIN_LDS) {
#pragma unroll
            for (int g = 0; g < ROWS / 4; ++g) {
                const float4 w4 = win_l[64 * g + lane];
#pragma unroll
                for (int t = 0; t < T; ++t) {
                    r[t * ROWS + 4 * g] *= w4.x; r[t * ROWS + 4 * g + 1] *= w4.y;
                    r[t * ROWS + 4 * g + 2] *= w4.z; r[t * ROWS + 4 * g + 3] *= w4.w;
                }
            }
        } else {
#pragma unroll
            for (int t = 0; t < T; ++t) {
#pragma unroll
                for (int j = 0; j < ROWS; ++j) r[t * ROWS + j] *= win[j];
            }
        }

        // ---- step 1: real DFT over n2, twiddle, exchange -------------------------------------------
#pragma unroll
        for (int t = 0; t < T; ++t) {
            float ar[R], ai[R];
            fc::rdft_cols<R>(r + t * ROWS, ar, ai);
            const int s0 = t * R;
            X[2 * COL_STRIDE * s0 + y0[s0 & 3]] = ar[0];   // y[n1]      = sum over even n2
            X[2 * COL_STRIDE * s0 + y1[s0 & 3]] = ai[0];   // y[64 + n1] = sum over odd n2
#pragma unroll
            for (int k = 1; k < R; ++k) {
                const float2 twk = TW_LDS ? tw_l[(k - 1) * 64 + lane] : tw[k];
                fc::cmul(ar[k], ai[k], twk.x, twk.y);
                X2[(s0 + k) * COL_STRIDE + xw[(s0 + k) & 3]] = make_float2(ar[k], ai[k]);
            }
        }

        // ---- step 2: quad DFT-4 (DPP), twiddle, DFT-16 over m ---------------------------------------
        float dr[16], di[16];
        {
            float2 w6[16];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float4 v = *reinterpret_cast<const float4*>(X + (xr_base ^ (4 * j)));
                dr[2 * j] = v.x; di[2 * j] = v.y; dr[2 * j + 1] = v.z; di[2 * j + 1] = v.w;
            }
            const float4* wsrc = reinterpret_cast<const float4*>(
                __builtin_assume_aligned(reinterpret_cast<const char*>(w64_l) + kb * (W64_STRIDE * 8), 16));
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float4 v = wsrc[j];
                w6[2 * j + 1] = make_float2(v.x, v.y);
                if (j < 7) w6[2 * j + 2] = make_float2(v.z, v.w);
            }
            __builtin_amdgcn_sched_barrier(0);   // all exchange + twiddle reads in flight
            quad_fmac8<2>(dr, sgn1); quad_fmac8<2>(dr + 8, sgn1);
            quad_fmac8<2>(di, sgn1); quad_fmac8<2>(di + 8, sgn1);
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float xr = dr[i], xi = di[i];
                dr[i] = lane3 ? -xi : xr;               // lane 3 holds -(x1 - x3): multiply by +i
                di[i] = lane3 ? xr : xi;
            }
            quad_fmac8<1>(dr, sgn2); quad_fmac8<1>(dr + 8, sgn2);
            quad_fmac8<1>(di, sgn2); quad_fmac8<1>(di + 8, sgn2);
#pragma unroll
            for (int m = 1; m < 16; ++m) fc::cmul(dr[m], di[m], w6[m].x, w6[m].y);
            fc::cfft16(dr, di);
        }

        // ---- power spectra -> LDS (natural bin order per frame) --------------------------------------
        // All exchange reads of this wavefront are complete (data dependence above), so the region
        // can be overwritten.  The special slots go to their scratch first so that their round trip
        // through LDS overlaps the regular power computation.
        if (R == 4) {
            // A special slot's 64 outputs sit in the first four lanes of a row of 16.  An LDS write
            // costs the same with few active lanes as with 64, so they are first spread over the
            // row with DPP row shifts -- lane 4 g + q of the row takes outputs ka = 4 i + g of lane
            // q -- and leave in four writes instead of sixteen.
            float cr[4], ci[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                cr[i] = dr[fc::cfft16_pos(4 * i)];
                ci[i] = di[fc::cfft16_pos(4 * i)];
                cr[i] = dpp_row_shr<4, 0x2>(cr[i], dr[fc::cfft16_pos(4 * i + 1)]);
                ci[i] = dpp_row_shr<4, 0x2>(ci[i], di[fc::cfft16_pos(4 * i + 1)]);
                cr[i] = dpp_row_shr<8, 0x4>(cr[i], dr[fc::cfft16_pos(4 * i + 2)]);
                ci[i] = dpp_row_shr<8, 0x4>(ci[i], di[fc::cfft16_pos(4 * i + 2)]);
                cr[i] = dpp_row_shr<12, 0x8>(cr[i], dr[fc::cfft16_pos(4 * i + 3)]);
                ci[i] = dpp_row_shr<12, 0x8>(ci[i], di[fc::cfft16_pos(4 * i + 3)]);
            }
            if (row_special) {
#pragma unroll
                for (int i = 0; i < 4; ++i) s_wr4[16 * i] = make_float2(cr[i], ci[i]);
            }
        } else if (special) {
#pragma unroll
            for (int ka = 0; ka < 16; ++ka)
                s_wr[4 * ka] = make_float2(dr[fc::cfft16_pos(ka)], di[fc::cfft16_pos(ka)]);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (!special) {
#pragma unroll
            for (int ka = 0; ka < 16; ++ka) {
                const float re = dr[fc::cfft16_pos(ka)], im = di[fc::cfft16_pos(ka)];
                const float pw = apply_power<PW>(re * re + im * im, a.pw_mode, a.half_power);
                if (ka < 8) X[pb0 + 8 * R * ka] = pw; else X[pb1 + 8 * R * (15 - ka)] = pw;
            }
        }
#pragma unroll
        for (int t = 0; t < T; ++t) {
            // C = DFT-64 of c[r] = y[2r] + i y[2r+1]; Y[j] = E[j] + W_128^j O[j] = X[R j]
            const float2 cj = X2[s_rd0 + t * (kp2::SPEC_STRIDE / 2)];
            const float2 cm = X2[s_rd1 + t * (kp2::SPEC_STRIDE / 2)];
            const float er = cj.x + cm.x, ei = cj.y - cm.y;   // 2 E[j]
            float tr = cj.y + cm.y, ti_ = cm.x - cj.x;        // 2 O[j] = -i (C[j] - conj C[64-j])
            fc::cmul(tr, ti_, tw128.x, tw128.y);
            const float yr = er + tr, yi = ei + ti_;          // 2 Y[j]
            X[kp2::p_base<LOG2N>(t) + R * lane] = apply_power<PW>(0.25f * (yr * yr + yi * yi), a.pw_mode, a.half_power);
            if (lane == 0) {
                const float zr = er - tr, zi = ei - ti_;      // 2 Y[64] (real)
                X[kp2::p_base<LOG2N>(t) + N / 2] = apply_power<PW>(0.25f * (zr * zr + zi * zi), a.pw_mode, a.half_power);
            }
        }

        if (a.spec_out) {
#pragma unroll
            for (int t = 0; t < T; ++t) {
                if (t < nvalid) {
                    float* so = a.spec_out + (row0 + t) * NB;
                    for (int k = lane; k < NB; k += 64) so[k] = X[kp2::p_base<LOG2N>(t) + k];
                }
            }
        }

        // ---- sparse mel filterbank: lane = (frame, filter) pair --------------------------------------
        // Round rd takes the filters rd FPR .. rd FPR + FPR - 1 of all T frames; filters of one round
        // have similar widths, so each round reads only as many quads as its widest filter.
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
