// 2048-point front end for gfx950: one wavefront per frame, no barriers in the frame loop.
//
//   pad / frame / window -> 2048-point real FFT -> |X|^power -> sparse mel filterbank
//
// replaces the stft + abs**power + mel einsum part of librosa.feature.mfcc as called at
// /root/reference/src/preprocess.py:134.
//
// Frame a[n], n = n1 + 64 n2: lane n1 holds the 32 samples n2 = 0..31.
//  step 1  real 32-point DFT over n2 in registers -> A[n1][k2], k2 = 0..16
//  twiddle B[n1][k2] = A[n1][k2] W_2048^(n1 k2), k2 = 1..15
//  exchange through LDS (16 columns of 64 complex; column 0 carries the two real columns
//          k2 = 0 and 16 as the even/odd-folded real sequence of a 128-point real DFT)
//  step 2  lane (k2c = lane>>2, q = lane&3) reads column[k2c][16 q + m], m = 0..15 (128
//          contiguous bytes): 4-point DFT across the quad with DPP (output kb), twiddle
//          W_64^(m kb), 16-point DFT over m (output ka) -> X[128 ka + 32 kb + k2c];
//          ka >= 8 are the mirror images of bins 17..31 mod 32
//  column 0 is post-processed as a 128-point real DFT -> bins 16 j, j = 0..64
//  mel     the power spectrum is written to the wavefront's LDS region in natural bin order;
//          lane = filter, each lane reads its filter's contiguous bins four at a time
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "fft_core.h"

namespace k2048 {
constexpr int NFFT = 2048;
constexpr int NBIN = 1025;
constexpr int TILE_FRAMES = 16;
// Exchange column: 64 complex, 2 complex of padding after every 16, stride 72: the
// ds_write_b64 of step 1 and the ds_read_b128 of step 2 are both bank-conflict free.
constexpr int COL_STRIDE = 72;
constexpr int XCH_FLOATS = (15 * COL_STRIDE + 70) * 2;   // 2300 floats per wavefront
constexpr int SPECIAL_OFF = 1152;                 // float offset of the column-0 scratch
constexpr int SPECIAL_ROW = 24;                   // complex per kb row of that scratch
constexpr int TW1_FLOATS = 15 * 64 * 2;
constexpr int W64_STRIDE = 18;                    // complex per kb row (16 + 2 pad)
constexpr int W64_FLOATS = 4 * W64_STRIDE * 2;
constexpr int WIN_FLOATS = 2048;
constexpr int LDS_LIMIT = 160 * 1024;
constexpr int MAX_MEL_PASS = 4;                   // n_mels <= 256
constexpr int MEL_CHUNK = 4;                      // ds_read_b128 pairs in flight per batch (mel)

enum TableHome { IN_LDS = 0, IN_REGS = 1, IN_GLOBAL = 2 };

// LDS bytes of a variant: per-wavefront exchange regions, then the tables that live in LDS.
constexpr size_t lds_bytes(int waves, int win, int tw1, bool w64_regs, int melw_floats) {
    return sizeof(float) * ((size_t)waves * XCH_FLOATS + (tw1 == IN_LDS ? TW1_FLOATS : 0) +
                            (w64_regs ? 0 : W64_FLOATS) + (win == IN_LDS ? WIN_FLOATS : 0) +
                            (size_t)melw_floats);
}
}  // namespace k2048

struct __attribute__((aligned(32))) Tile {
    int64_t soff;  // first sample of the utterance in d_pcm
    int64_t row0;  // output row of the tile's first frame
    int32_t T;     // samples in the utterance
    int32_t f0;    // first frame of the tile within the utterance
    int32_t F;     // frames in the utterance
    int32_t grp;   // top_db group (slot of the running maximum)
};
static_assert(sizeof(Tile) == 32, "Tile must be 32 bytes");

struct WsHeader {
    int32_t n_tiles;
    int32_t pad[63];
};

struct MelPass {
    int32_t rounds;      // bins each lane reads in this pass (multiple of 4 * MEL_CHUNK)
    int32_t round_off;   // first round of this pass in the weight table
};

struct FrontArgs {
    const float* pcm;
    const Tile* tiles;
    const WsHeader* hdr;
    const float* window;     // 2048
    const float2* tw1;       // 15 x 64, scaled
    const float2* w64;       // 4 x 16
    const float2* tw128;     // 64
    const float4* melw;      // (total_rounds / 4) x 64 quads of weights
    const int32_t* mel_start;  // bin index (multiple of 4) each lane starts reading at
    const int32_t* mel_filt;
    const MelPass* mel_pass;
    float* mel_out;          // (rows, n_mels) or nullptr
    float* spec_out;         // (rows, 1025) or nullptr
    uint32_t* gmax;
    int32_t n_pass;
    int32_t n_mels;
    int32_t total_rounds;
    int32_t hop;
    int32_t pad;             // n_fft/2 if center else 0
    int32_t reflect;
    int32_t pw_mode;         // 0: power 2, 1: power 1, 2: general
    int32_t ablate;          // development only: bit 0 skips the mel stage
    float half_power;        // power / 2
};

namespace k2048 {

__device__ __forceinline__ float quad_swap2(float v) {  // lane i <- lane i^2 within its quad
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x4E,
                                                              0xF, 0xF, true));
}
__device__ __forceinline__ float quad_swap1(float v) {  // lane i <- lane i^1
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1,
                                                              0xF, 0xF, true));
}

// |X|^power from |X|^2.  PW = 0: power 2 (the reference's); PW = 1: any other power, decided at
// run time (kept out of the power-2 instantiations: powf is ~200 instructions per call site).
template <int PW>
__device__ __forceinline__ float apply_power(float mag2, int pw_mode, float half_power) {
    if (PW == 0) return mag2;
    return pw_mode == 1 ? sqrtf(mag2) : powf(mag2, half_power);
}

__device__ __forceinline__ int64_t reflect_index(int64_t i, int64_t T) {
    if (T <= 1) return 0;
    const int64_t period = 2 * (T - 1);
    int64_t m = i % period;
    if (m < 0) m += period;
    return m < T ? m : period - m;
}

__device__ __forceinline__ void flush_max(float runmax, int grp, uint32_t* gmax, int lane) {
    if (grp < 0) return;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) runmax = fmaxf(runmax, __shfl_xor(runmax, o));
    if (lane == 0 && runmax > 0.0f) atomicMax(&gmax[grp], __float_as_uint(runmax));
}

// Edge frames (those that overlap the zero / reflect padding) are rare: stage them through the
// wavefront's LDS region with a rolled loop so the hot path stays small.
__device__ __noinline__ void stage_edge_frame(float* X, const float* p, int start, int T, int reflect,
                                              int lane) {
#pragma unroll 1
    for (int n2 = 0; n2 < 32; ++n2) {
        const int idx = start + lane + 64 * n2;
        float v = 0.0f;
        if (reflect) {
            if (T > 0) v = p[reflect_index(idx, T)];
        } else if (idx >= 0 && idx < T) {
            v = p[idx];
        }
        X[lane + 64 * n2] = v;
    }
}

}  // namespace k2048

// WAVES     wavefronts per workgroup (one workgroup per CU: LDS-limited)
// WIN/TW1   where the window / step-1 twiddle tables live (k2048::TableHome)
// W64_REGS  step-2 twiddles in registers instead of LDS
// MELW_LDS  mel weights in LDS instead of streamed from global memory
// HOPR      0, or hop_length / 64 when that is an integer: samples shared with the previous frame
//           stay in registers and only the new rows are loaded, one frame ahead
// PW        0: |X|^2; 1: run-time power
template <int WAVES, int WIN, int TW1, bool W64_REGS, bool MELW_LDS, int HOPR, int PW>
__global__ __launch_bounds__(WAVES * 64) void k_melspec_2048(const FrontArgs a) {
    using namespace k2048;
    constexpr int THREADS = WAVES * 64;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float* X = lds + wave * XCH_FLOATS;
    float2* X2 = reinterpret_cast<float2*>(X);
    float* tab = lds + WAVES * XCH_FLOATS;
    float2* tw1_l = reinterpret_cast<float2*>(tab);
    if (TW1 == IN_LDS) tab += TW1_FLOATS;
    float2* w64_l = reinterpret_cast<float2*>(tab);
    if (!W64_REGS) tab += W64_FLOATS;
    float* win_l = tab;
    if (WIN == IN_LDS) tab += WIN_FLOATS;
    float4* melw_l = reinterpret_cast<float4*>(tab);

    for (int i = tid; i < WAVES * XCH_FLOATS; i += THREADS) lds[i] = 0.0f;
    if (TW1 == IN_LDS) { for (int i = tid; i < 15 * 64; i += THREADS) tw1_l[i] = a.tw1[i]; }
    if (!W64_REGS) { if (tid < 64) w64_l[(tid >> 4) * W64_STRIDE + (tid & 15)] = a.w64[tid]; }
    if (WIN == IN_LDS) { for (int i = tid; i < 2048; i += THREADS) win_l[i] = a.window[i]; }
    if (MELW_LDS) { for (int i = tid; i < a.total_rounds * 16; i += THREADS) melw_l[i] = a.melw[i]; }
    __syncthreads();

    const int q = lane & 3, k2c = lane >> 2;
    const int kb = ((q & 1) << 1) | (q >> 1);       // the quad DFT leaves output kb in lane q
    float win[WIN == IN_REGS ? 32 : 1];
    if (WIN == IN_REGS) {
#pragma unroll
        for (int n2 = 0; n2 < 32; ++n2) win[n2] = a.window[lane + 64 * n2];
    }
    float2 w64r[W64_REGS ? 16 : 1];
    if (W64_REGS) {
#pragma unroll
        for (int m = 0; m < 16; ++m) w64r[m] = a.w64[kb * 16 + m];
    }
    const float2 tw128 = a.tw128[lane];

    const float sgn1 = (q & 2) ? -1.0f : 1.0f;
    const float sgn2 = (q & 1) ? -1.0f : 1.0f;
    const bool lane3 = (q == 3);
    // exchange addressing (complex index c of column k2 lives at k2 * 72 + c + 2 * (c >> 4))
    const int xw_off = lane + 2 * (lane >> 4);                  // step-1 write, + k2 * COL_STRIDE
    const int y_off = lane + 4 * (lane >> 5);                   // column 0 as 128 reals
    const int xr_off = k2c * COL_STRIDE + 18 * q;               // step-2 read (complex index)
    // power spectrum addressing: ka < 8 -> bin 128 ka + pb0; ka >= 8 -> bin 128 (15 - ka) + pb1
    const int pb0 = 32 * kb + k2c;
    const int pb1 = 128 - 32 * kb - k2c;
    // column-0 scratch: C[j] at (j & 3) * SPECIAL_ROW + (j >> 2)
    const int sj = (lane & 3) * SPECIAL_ROW + (lane >> 2);
    const int sm = (((64 - lane) & 63) & 3) * SPECIAL_ROW + (((64 - lane) & 63) >> 2);

    // per-lane mel work, loop-invariant: first bin, filter index; per pass: rounds, weight offset
    int mel_st[MAX_MEL_PASS], mel_fi[MAX_MEL_PASS], mel_rounds[MAX_MEL_PASS], mel_roff[MAX_MEL_PASS];
#pragma unroll
    for (int p = 0; p < MAX_MEL_PASS; ++p) {
        mel_st[p] = 0; mel_fi[p] = -1; mel_rounds[p] = 0; mel_roff[p] = 0;
        if (p < a.n_pass) {
            mel_st[p] = a.mel_start[p * 64 + lane];
            mel_fi[p] = a.mel_filt[p * 64 + lane];
            mel_rounds[p] = __builtin_amdgcn_readfirstlane(a.mel_pass[p].rounds);      // scalar
            mel_roff[p] = __builtin_amdgcn_readfirstlane(a.mel_pass[p].round_off);
        }
    }

    float runmax = 0.0f;
    int cur_grp = -1;
    const int n_tiles = a.hdr->n_tiles;

    // Work unit = one frame slot (tile, frame-in-tile).  Every wavefront of the grid owns one
    // contiguous run of slots and walks it in order: consecutive frames of a run share 75 % of
    // their samples, which the wavefront re-reads from L2 a few microseconds later (runs of
    // different wavefronts do not overlap, so each sample leaves HBM once), and a wavefront
    // stays within one utterance for many frames (one atomicMax per utterance change).
    const int64_t n_slots = (int64_t)n_tiles * TILE_FRAMES;
    const int64_t slot_stride = 1;
    const int64_t n_runs = (int64_t)gridDim.x * WAVES;
    const int64_t per_run = (n_slots + n_runs - 1) / n_runs;
    const int64_t run = (int64_t)blockIdx.x * WAVES + wave;
    const int64_t slot_end = min(n_slots, (run + 1) * per_run);

    Tile ti;           // current work
    int t = 0;
    int64_t slot = run * per_run;

    // find the first valid slot at or after `s`; wave-uniform
    auto next_valid = [&](int64_t s, Tile& o_ti, int& o_t, int64_t& o_slot) -> bool {
        for (; s < slot_end; s += slot_stride) {
            const Tile c = a.tiles[s >> 4];
            const int tt = c.f0 + (int)(s & 15);
            if (tt < c.F) { o_ti = c; o_t = tt; o_slot = s; return true; }
        }
        return false;
    };
    auto is_interior = [&](const Tile& c, int tt) -> bool {
        const int start = tt * a.hop - a.pad;
        return start >= 0 && start + NFFT <= c.T;
    };

    bool have = next_valid(slot, ti, t, slot);
    // HOPR > 0 (hop = 64 * HOPR): consecutive frames of a run share their rows n2 >= HOPR, so the
    // unwindowed samples stay in registers, shift by HOPR rows per frame, and only the HOPR new
    // rows are loaded -- requested one frame ahead.
    float raw[HOPR > 0 ? 32 : 1];
    float rn[HOPR > 0 ? HOPR : 1];
    bool raw_ok = false, rn_ok = false;

    while (have) {
        if (ti.grp != cur_grp) {
            flush_max(runmax, cur_grp, a.gmax, lane);
            cur_grp = ti.grp;
            runmax = 0.0f;
        }
        const int64_t row = ti.row0 + (t - ti.f0);
        const bool interior = is_interior(ti, t);
        Tile ti_n;
        int t_n = 0;
        int64_t slot_n = 0;
        const bool have_n = next_valid(slot + slot_stride, ti_n, t_n, slot_n);

        // ---- samples --------------------------------------------------------------------------------
        float r[32];
        if (interior) {
            const float* pp = a.pcm + ti.soff + (t * a.hop - a.pad) + lane;
            if (HOPR > 0) {
                if (raw_ok && rn_ok) {
#pragma unroll
                    for (int n2 = 0; n2 < 32 - HOPR; ++n2) raw[n2] = raw[n2 + HOPR];
#pragma unroll
                    for (int j = 0; j < HOPR; ++j) raw[32 - HOPR + j] = rn[j];
                } else {
#pragma unroll
                    for (int n2 = 0; n2 < 32; ++n2) raw[n2] = pp[64 * n2];
                }
                raw_ok = true;
                rn_ok = have_n && ti_n.soff == ti.soff && t_n == t + 1 && is_interior(ti_n, t_n);
                if (rn_ok) {
                    const float* pn = pp + a.hop + 64 * (32 - HOPR);
#pragma unroll
                    for (int j = 0; j < HOPR; ++j) rn[j] = pn[64 * j];
                }
                __builtin_amdgcn_sched_barrier(0);   // keep the prefetch ahead of this frame's work
#pragma unroll
                for (int n2 = 0; n2 < 32; ++n2) r[n2] = raw[n2];
            } else {
#pragma unroll
                for (int n2 = 0; n2 < 32; ++n2) r[n2] = pp[64 * n2];
            }
        } else {
            // edge frames (overlapping the padding) come through LDS
            stage_edge_frame(X, a.pcm + ti.soff, t * a.hop - a.pad, ti.T, a.reflect, lane);
#pragma unroll
            for (int n2 = 0; n2 < 32; ++n2) r[n2] = X[lane + 64 * n2];
            raw_ok = false;
            rn_ok = false;
        }
        if (WIN == IN_REGS) {
#pragma unroll
            for (int n2 = 0; n2 < 32; ++n2) r[n2] *= win[n2];
        } else {
            float wv[32];
#pragma unroll
            for (int n2 = 0; n2 < 32; ++n2)
                wv[n2] = (WIN == IN_LDS) ? win_l[lane + 64 * n2] : a.window[lane + 64 * n2];
            __builtin_amdgcn_sched_barrier(0);   // all table reads in flight before the first use
#pragma unroll
            for (int n2 = 0; n2 < 32; ++n2) r[n2] *= wv[n2];
        }

        // ---- step 1: real DFT-32 over n2, twiddle, exchange ------------------------------------
        {
            float zr[16], zi[16];
#pragma unroll
            for (int m = 0; m < 16; ++m) { zr[m] = r[2 * m]; zi[m] = r[2 * m + 1]; }
            fc::cfft16(zr, zi);
            __builtin_amdgcn_sched_barrier(0);
            float2 tw[16];
#pragma unroll
            for (int k2 = 1; k2 < 16; ++k2)
                tw[k2] = (TW1 == IN_LDS) ? tw1_l[(k2 - 1) * 64 + lane] : a.tw1[(k2 - 1) * 64 + lane];
            __builtin_amdgcn_sched_barrier(0);   // twiddle reads overlap the conjugate split
            float ar[16], ai[16];
            fc::rdft32_split(zr, zi, ar, ai);
            X[y_off] = ar[0];                    // y[n1]      = sum over even n2
            X[72 + y_off] = ai[0];               // y[64 + n1] = sum over odd n2
#pragma unroll
            for (int k2 = 1; k2 < 16; ++k2) {
                fc::cmul(ar[k2], ai[k2], tw[k2].x, tw[k2].y);
                X2[k2 * COL_STRIDE + xw_off] = make_float2(ar[k2], ai[k2]);
            }
        }

        // ---- step 2: quad DFT-4 (DPP), twiddle, DFT-16 over m -----------------------------------
        float dr[16], di[16];
        {
            float2 w6[16];
            const float4* src = reinterpret_cast<const float4*>(X2 + xr_off);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float4 v = src[j];
                dr[2 * j] = v.x; di[2 * j] = v.y; dr[2 * j + 1] = v.z; di[2 * j + 1] = v.w;
            }
            if (W64_REGS) {
#pragma unroll
                for (int m = 1; m < 16; ++m) w6[m] = w64r[m];
            } else {
                const float4* wsrc = reinterpret_cast<const float4*>(w64_l + kb * W64_STRIDE);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float4 v = wsrc[j];
                    w6[2 * j] = make_float2(v.x, v.y); w6[2 * j + 1] = make_float2(v.z, v.w);
                }
            }
            __builtin_amdgcn_sched_barrier(0);   // all exchange + twiddle reads in flight
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                float xr = fmaf(sgn1, dr[i], quad_swap2(dr[i]));
                float xi = fmaf(sgn1, di[i], quad_swap2(di[i]));
                const float yr = lane3 ? xi : xr;       // lane 3: multiply by -i
                const float yi = lane3 ? -xr : xi;
                dr[i] = fmaf(sgn2, yr, quad_swap1(yr));
                di[i] = fmaf(sgn2, yi, quad_swap1(yi));
            }
#pragma unroll
            for (int m = 1; m < 16; ++m) fc::cmul(dr[m], di[m], w6[m].x, w6[m].y);
            fc::cfft16(dr, di);
        }

        // ---- power spectrum -> LDS (natural bin order) -------------------------------------------
        // All exchange reads of this wavefront are complete (data dependence above), so the
        // exchange region can be overwritten.  Column 0 goes to its scratch first so that its
        // round trip through LDS overlaps the regular power computation.
        if (lane < 4) {
            float2* S = reinterpret_cast<float2*>(X + SPECIAL_OFF) + kb * SPECIAL_ROW;
#pragma unroll
            for (int ka = 0; ka < 16; ++ka)
                S[ka] = make_float2(dr[fc::cfft16_pos(ka)], di[fc::cfft16_pos(ka)]);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (lane >= 4) {
#pragma unroll
            for (int ka = 0; ka < 16; ++ka) {
                const float re = dr[fc::cfft16_pos(ka)], im = di[fc::cfft16_pos(ka)];
                const float pw = apply_power<PW>(re * re + im * im, a.pw_mode, a.half_power);
                if (ka < 8) X[pb0 + 128 * ka] = pw; else X[pb1 + 128 * (15 - ka)] = pw;
            }
        }
        {
            // column 0: C = DFT-64 of c[r] = y[2r] + i y[2r+1]; Y[j] = E[j] + W_128^j O[j]
            const float2* S = reinterpret_cast<const float2*>(X + SPECIAL_OFF);
            const float2 cj = S[sj];
            const float2 cm = S[sm];
            const float er = cj.x + cm.x, ei = cj.y - cm.y;   // 2 E[j]
            float tr = cj.y + cm.y, ti_ = cm.x - cj.x;        // 2 O[j] = -i (C[j] - conj C[64-j])
            fc::cmul(tr, ti_, tw128.x, tw128.y);
            const float yr = er + tr, yi = ei + ti_;          // 2 Y[j]
            X[16 * lane] = apply_power<PW>(0.25f * (yr * yr + yi * yi), a.pw_mode, a.half_power);
            if (lane == 0) {
                const float zr = er - tr, zi = ei - ti_;      // 2 Y[64] (real)
                X[1024] = apply_power<PW>(0.25f * (zr * zr + zi * zi), a.pw_mode, a.half_power);
            }
        }

        if (a.spec_out) {
            float* so = a.spec_out + row * NBIN;
#pragma unroll
            for (int i = 0; i < 16; ++i) so[i * 64 + lane] = X[i * 64 + lane];
            if (lane == 0) so[1024] = X[1024];
        }

        // ---- sparse mel filterbank: lane = filter, four bins per read -----------------------------
        if (a.mel_out && !(a.ablate & 1)) {
            float* mo = a.mel_out + row * a.n_mels;
#pragma unroll
            for (int p = 0; p < MAX_MEL_PASS; ++p) {
                if (p < a.n_pass) {
                    const float4* wp = (MELW_LDS ? melw_l : a.melw) + (mel_roff[p] >> 2) * 64 + lane;
                    const float4* pp = reinterpret_cast<const float4*>(X + mel_st[p]);
                    float acc0 = 0.0f, acc1 = 0.0f;
                    const int nq = mel_rounds[p] >> 2;
                    for (int i = 0; i < nq; i += MEL_CHUNK) {
                        float4 w[MEL_CHUNK], v[MEL_CHUNK];
#pragma unroll
                        for (int j = 0; j < MEL_CHUNK; ++j) { w[j] = wp[(i + j) * 64]; v[j] = pp[i + j]; }
                        __builtin_amdgcn_sched_barrier(0);   // 2 * MEL_CHUNK reads in flight
#pragma unroll
                        for (int j = 0; j < MEL_CHUNK; ++j) {
                            acc0 = fmaf(w[j].x, v[j].x, acc0);
                            acc1 = fmaf(w[j].y, v[j].y, acc1);
                            acc0 = fmaf(w[j].z, v[j].z, acc0);
                            acc1 = fmaf(w[j].w, v[j].w, acc1);
                        }
                    }
                    const float acc = acc0 + acc1;
                    if (mel_fi[p] >= 0) {
                        mo[mel_fi[p]] = acc;
                        runmax = fmaxf(runmax, acc);
                    }
                }
            }
        }

        have = have_n; ti = ti_n; t = t_n; slot = slot_n;
    }
    flush_max(runmax, cur_grp, a.gmax, lane);
}
