// libmfcc_hip: hand-written gfx950 kernels for the librosa MFCC pipeline and their C ABI
// (include/mfcc_hip.h).  Replaces the per-utterance CPU call at
// /root/reference/src/preprocess.py:134 for a ragged batch of utterances.
//
// Pipeline (all float32 on device):
//   k_prepare_scan / k_prepare_tiles  ragged batch -> table of 16-frame tiles
//   k_melspec_2048                    pad + frame + window + 2048-point real FFT + |.|^power
//                                     + sparse mel filterbank, one wavefront per frame;
//                                     per-utterance maximum by atomicMax
//   k_db_dct                          10 log10, top_db clamp against the utterance maximum,
//                                     DCT on v_mfma_f32_16x16x4_f32, lifter, (F, n_mfcc) store
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/mfcc_hip.h"
#include "fft_core.h"
#include "host_tables.h"

namespace {

thread_local std::string g_err;

int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}

#define HIP_TRY(expr)                                                                     \
    do {                                                                                  \
        hipError_t e_ = (expr);                                                           \
        if (e_ != hipSuccess)                                                             \
            return fail(MFCC_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));   \
    } while (0)

// ------------------------------------------------------------------------------------------
// Ragged batch -> tile table
// ------------------------------------------------------------------------------------------
constexpr int TILE_FRAMES = 16;

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

// One block.  tile_off[u] = number of tiles before utterance u; also zeroes the group maxima.
__global__ __launch_bounds__(1024) void k_prepare_scan(const int64_t* __restrict__ frame_off,
                                                       int n_utts, int32_t* __restrict__ tile_off,
                                                       uint32_t* __restrict__ gmax,
                                                       WsHeader* __restrict__ hdr) {
    __shared__ int32_t part[1024];
    const int tid = threadIdx.x;
    const int seg = (n_utts + 1023) / 1024;
    const int lo = min(tid * seg, n_utts), hi = min(lo + seg, n_utts);
    int32_t s = 0;
    for (int u = lo; u < hi; ++u) {
        const int64_t F = frame_off[u + 1] - frame_off[u];
        s += (int32_t)((F + TILE_FRAMES - 1) / TILE_FRAMES);
        gmax[u] = 0u;
    }
    part[tid] = s;
    __syncthreads();
    if (tid == 0) {
        int32_t run = 0;
        for (int i = 0; i < 1024; ++i) { const int32_t v = part[i]; part[i] = run; run += v; }
        tile_off[n_utts] = run;
        hdr->n_tiles = run;
    }
    __syncthreads();
    int32_t run = part[tid];
    for (int u = lo; u < hi; ++u) {
        tile_off[u] = run;
        const int64_t F = frame_off[u + 1] - frame_off[u];
        run += (int32_t)((F + TILE_FRAMES - 1) / TILE_FRAMES);
    }
}

// One thread per tile: find its utterance by binary search in tile_off.
__global__ __launch_bounds__(256) void k_prepare_tiles(const int64_t* __restrict__ sample_off,
                                                       const int64_t* __restrict__ frame_off,
                                                       const int32_t* __restrict__ group,
                                                       const int32_t* __restrict__ tile_off,
                                                       int n_utts, Tile* __restrict__ tiles) {
    const int tile = blockIdx.x * blockDim.x + threadIdx.x;
    const int n_tiles = tile_off[n_utts];
    if (tile >= n_tiles) return;
    int lo = 0, hi = n_utts;  // last u with tile_off[u] <= tile
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (tile_off[mid] <= tile) lo = mid; else hi = mid;
    }
    const int u = lo;
    Tile t;
    t.soff = sample_off[u];
    t.T = (int32_t)(sample_off[u + 1] - sample_off[u]);
    t.F = (int32_t)(frame_off[u + 1] - frame_off[u]);
    t.f0 = (tile - tile_off[u]) * TILE_FRAMES;
    t.row0 = frame_off[u] + t.f0;
    t.grp = group ? group[u] : u;
    tiles[tile] = t;
}

// ------------------------------------------------------------------------------------------
// 2048-point front end: one wavefront per frame
// ------------------------------------------------------------------------------------------
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
namespace k2048 {
constexpr int NFFT = 2048;
constexpr int NBIN = 1025;
// Exchange column: 64 complex, 2 complex of padding after every 16, stride 72: the
// ds_write_b64 of step 1 and the ds_read_b128 of step 2 are both bank-conflict free.
constexpr int COL_STRIDE = 72;
constexpr int XCH_FLOATS = 16 * COL_STRIDE * 2;   // 2304 floats per wavefront
constexpr int SPECIAL_OFF = 1152;                 // float offset of the column-0 scratch
constexpr int SPECIAL_ROW = 24;                   // complex per kb row of that scratch
constexpr int TW1_FLOATS = 15 * 64 * 2;
constexpr int W64_STRIDE = 18;                    // complex per kb row (16 + 2 pad)
constexpr int W64_FLOATS = 4 * W64_STRIDE * 2;
constexpr int WIN_FLOATS = 2048;
constexpr int LDS_LIMIT = 160 * 1024;
constexpr int MAX_MEL_PASS = 4;                   // n_mels <= 256
constexpr int MEL_CHUNK = 2;                      // ds_read_b128 pairs in flight per batch (mel)
// LDS bytes of a variant: per-wavefront exchange regions, then tw1, then (optional) w64,
// window and mel-weight tables.
constexpr size_t lds_bytes(int waves, bool win_regs, bool w64_regs, int melw_floats) {
    return sizeof(float) * ((size_t)waves * XCH_FLOATS + TW1_FLOATS + (w64_regs ? 0 : W64_FLOATS) +
                            (win_regs ? 0 : WIN_FLOATS) + (size_t)melw_floats);
}
}  // namespace k2048

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

template <int WAVES, bool WIN_REGS, bool W64_REGS, bool MELW_LDS, int PW>
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
    tab += TW1_FLOATS;
    float2* w64_l = reinterpret_cast<float2*>(tab);
    if (!W64_REGS) tab += W64_FLOATS;
    float* win_l = tab;
    if (!WIN_REGS) tab += WIN_FLOATS;
    float4* melw_l = reinterpret_cast<float4*>(tab);

    for (int i = tid; i < WAVES * XCH_FLOATS; i += THREADS) lds[i] = 0.0f;
    for (int i = tid; i < 15 * 64; i += THREADS) tw1_l[i] = a.tw1[i];
    if (!W64_REGS) { if (tid < 64) w64_l[(tid >> 4) * W64_STRIDE + (tid & 15)] = a.w64[tid]; }
    if (!WIN_REGS) { for (int i = tid; i < 2048; i += THREADS) win_l[i] = a.window[i]; }
    if (MELW_LDS) { for (int i = tid; i < a.total_rounds * 16; i += THREADS) melw_l[i] = a.melw[i]; }
    __syncthreads();

    const int q = lane & 3, k2c = lane >> 2;
    const int kb = ((q & 1) << 1) | (q >> 1);       // the quad DFT leaves output kb in lane q
    float win[WIN_REGS ? 32 : 1];
    if (WIN_REGS) {
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
            mel_rounds[p] = a.mel_pass[p].rounds;
            mel_roff[p] = a.mel_pass[p].round_off;
        }
    }

    float runmax = 0.0f;
    int cur_grp = -1;
    const int n_tiles = a.hdr->n_tiles;

    // Work unit = one frame slot (tile, frame-in-tile); consecutive wavefronts take consecutive
    // frames so that the 75 % sample overlap between neighbouring frames is served by L1/L2.
    // The raw samples of the NEXT frame are requested before the current one is transformed
    // (software pipelining: the wavefront never waits on HBM latency with nothing to do).
    // Each workgroup owns one contiguous block of slots and its wavefronts interleave inside it,
    // so a wavefront stays within one utterance for many frames (one atomicMax per utterance
    // change, not per frame).
    const int64_t n_slots = (int64_t)n_tiles * TILE_FRAMES;
    const int64_t slot_stride = WAVES;
    const int64_t per_wg = ((n_slots + gridDim.x - 1) / gridDim.x + WAVES - 1) / WAVES * WAVES;
    const int64_t slot_end = min(n_slots, (int64_t)(blockIdx.x + 1) * per_wg);

    Tile ti;           // current work
    int t = 0;
    bool have = false, interior = false;
    int64_t slot = (int64_t)blockIdx.x * per_wg + wave;
    float r[32];

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

    have = next_valid(slot, ti, t, slot);
    if (have) {
        interior = is_interior(ti, t);
        if (interior) {
            const float* pp = a.pcm + ti.soff + (t * a.hop - a.pad) + lane;
#pragma unroll
            for (int n2 = 0; n2 < 32; ++n2) r[n2] = pp[64 * n2];
        }
    }

    while (have) {
        if (ti.grp != cur_grp) {
            flush_max(runmax, cur_grp, a.gmax, lane);
            cur_grp = ti.grp;
            runmax = 0.0f;
        }
        const int64_t row = ti.row0 + (t - ti.f0);

        // ---- request the next frame's samples ------------------------------------------------
        Tile ti_n;
        int t_n = 0;
        int64_t slot_n = 0;
        bool interior_n = false;
        float rn[32];
        const bool have_n = next_valid(slot + slot_stride, ti_n, t_n, slot_n);
        if (have_n) {
            interior_n = is_interior(ti_n, t_n);
            if (interior_n) {
                const float* pp = a.pcm + ti_n.soff + (t_n * a.hop - a.pad) + lane;
#pragma unroll
                for (int n2 = 0; n2 < 32; ++n2) rn[n2] = pp[64 * n2];
            }
        }
        __builtin_amdgcn_sched_barrier(0);       // keep the prefetch ahead of this frame's work

        // ---- edge frames come through LDS; then window -------------------------------------------
        if (!interior) {
            stage_edge_frame(X, a.pcm + ti.soff, t * a.hop - a.pad, ti.T, a.reflect, lane);
#pragma unroll
            for (int n2 = 0; n2 < 32; ++n2) r[n2] = X[lane + 64 * n2];
        }
        if (WIN_REGS) {
#pragma unroll
            for (int n2 = 0; n2 < 32; ++n2) r[n2] *= win[n2];
        } else {
            float wv[32];
#pragma unroll
            for (int n2 = 0; n2 < 32; ++n2) wv[n2] = win_l[lane + 64 * n2];
            __builtin_amdgcn_sched_barrier(0);   // all LDS reads in flight before the first use
#pragma unroll
            for (int n2 = 0; n2 < 32; ++n2) r[n2] *= wv[n2];
        }

        // ---- step 1: real DFT-32 over n2, twiddle, exchange ------------------------------
        {
            float2 tw[16];
#pragma unroll
            for (int k2 = 1; k2 < 16; ++k2) tw[k2] = tw1_l[(k2 - 1) * 64 + lane];
            __builtin_amdgcn_sched_barrier(0);   // twiddle reads overlap the in-register DFT
            float ar[16], ai[16];
            fc::rdft32(r, ar, ai);
            X[y_off] = ar[0];                    // y[n1]      = sum over even n2
            X[72 + y_off] = ai[0];               // y[64 + n1] = sum over odd n2 (complex index 32 + ...)
#pragma unroll
            for (int k2 = 1; k2 < 16; ++k2) {
                fc::cmul(ar[k2], ai[k2], tw[k2].x, tw[k2].y);
                X2[k2 * COL_STRIDE + xw_off] = make_float2(ar[k2], ai[k2]);
            }
        }

        // ---- step 2: quad DFT-4 (DPP), twiddle, DFT-16 over m -------------------------------
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

        // ---- power spectrum -> LDS (natural bin order) -------------------------------------
        // All exchange reads of this wavefront are complete (data dependence above), so the
        // exchange region can be overwritten.
        if (lane < 4) {
            float2* S = reinterpret_cast<float2*>(X + SPECIAL_OFF) + kb * SPECIAL_ROW;
#pragma unroll
            for (int ka = 0; ka < 16; ++ka)
                S[ka] = make_float2(dr[fc::cfft16_pos(ka)], di[fc::cfft16_pos(ka)]);
        } else {
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

        // ---- sparse mel filterbank: lane = filter, two bins per round ----------------------
        if (a.mel_out && !(a.ablate & 1)) {
            float* mo = a.mel_out + row * a.n_mels;
#pragma unroll
            for (int p = 0; p < MAX_MEL_PASS; ++p) {
                if (p < a.n_pass) {
                    const float4* wp = (MELW_LDS ? melw_l : a.melw) + (mel_roff[p] >> 2) * 64 + lane;
                    const float4* pp = reinterpret_cast<const float4*>(X + mel_st[p]);
                    float acc0 = 0.0f, acc1 = 0.0f;
                    for (int i = 0; i < (mel_rounds[p] >> 2); i += MEL_CHUNK) {
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

        have = have_n; ti = ti_n; t = t_n; slot = slot_n; interior = interior_n;
#pragma unroll
        for (int n2 = 0; n2 < 32; ++n2) r[n2] = rn[n2];
    }
    flush_max(runmax, cur_grp, a.gmax, lane);
}

// ------------------------------------------------------------------------------------------
// power_to_db + DCT on MFMA: one wavefront per 16-frame tile
// ------------------------------------------------------------------------------------------
struct DbDctArgs {
    const Tile* tiles;
    const WsHeader* hdr;
    const float* mel;        // (rows, n_mels)
    const uint32_t* gmax;
    const float* dct_frag;   // [rt][step][lane]
    const float* lifter;     // n_mfcc
    float* out;              // (rows, n_mfcc)
    int32_t n_mels;
    int32_t n_mels_pad;      // multiple of 64
    int32_t n_mfcc;
    int32_t n_rt;            // row tiles of 16 coefficients
    int32_t vec_ok;          // mel rows are 16-byte aligned
    float amin;
    float ref_db;            // 10 log10(max(amin, |ref|))
    float top_db;            // < 0: no clamp
};

typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void k_db_dct(const DbDctArgs a) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int tile = blockIdx.x * 4 + wave;
    if (tile >= a.hdr->n_tiles) return;
    const Tile ti = a.tiles[tile];
    const int nvalid = min(TILE_FRAMES, ti.F - ti.f0);
    const int f = lane & 15, kq = lane >> 4;
    const int64_t row = ti.row0 + min(f, nvalid - 1);
    const float* mrow = a.mel + row * a.n_mels;

    float thr = -INFINITY;
    if (a.top_db >= 0.0f) {
        const float gm = __uint_as_float(a.gmax[ti.grp]);
        const float maxdb = 10.0f * log10f(fmaxf(a.amin, gm)) - a.ref_db;
        thr = maxdb - a.top_db;
    }

    const int steps_total = a.n_mels_pad / 4;
    for (int rt0 = 0; rt0 < a.n_rt; rt0 += 4) {
        f32x4 acc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
        for (int c = 0; c < a.n_mels_pad / 64; ++c) {
            float L[16];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int k = 64 * c + 16 * j + 4 * kq;
                float v[4];
                if (a.vec_ok && k + 3 < a.n_mels) {
                    const float4 t4 = *reinterpret_cast<const float4*>(mrow + k);
                    v[0] = t4.x; v[1] = t4.y; v[2] = t4.z; v[3] = t4.w;
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] = (k + i < a.n_mels) ? mrow[k + i] : 0.0f;
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float db = 10.0f * log10f(fmaxf(a.amin, v[i])) - a.ref_db;
                    L[4 * j + i] = fmaxf(db, thr);
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (rt0 + r < a.n_rt) {
                    const float* fr = a.dct_frag + ((size_t)(rt0 + r) * steps_total + 16 * c) * 64 + lane;
#pragma unroll
                    for (int s = 0; s < 16; ++s)
                        acc[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(fr[s * 64], L[s], acc[r], 0, 0, 0);
                }
            }
        }
        if (f < nvalid) {
            float* orow = a.out + (ti.row0 + f) * a.n_mfcc;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int cidx = (rt0 + r) * 16 + 4 * kq + i;
                    if (cidx < a.n_mfcc) orow[cidx] = acc[r][i] * a.lifter[cidx];
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// CMVN statistics (extension)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_cmvn_accumulate(const float* __restrict__ x, int64_t n_rows,
                                                         int n_mfcc, double* __restrict__ stats) {
    // thread (c, r): column c = tid % n_mfcc_pad handled by rows r, r + stride, ...
    extern __shared__ double sh[];  // 2 * blockDim.x
    const int tid = threadIdx.x;
    const int cols = n_mfcc;
    const int rows_per_block = blockDim.x / cols;
    const int c = tid % cols, rr = tid / cols;
    double s = 0.0, s2 = 0.0;
    if (rr < rows_per_block) {
        for (int64_t r = (int64_t)blockIdx.x * rows_per_block + rr; r < n_rows;
             r += (int64_t)gridDim.x * rows_per_block) {
            const double v = (double)x[r * cols + c];
            s += v; s2 += v * v;
        }
    }
    sh[tid] = s; sh[blockDim.x + tid] = s2;
    __syncthreads();
    if (tid < cols) {
        double a = 0.0, b = 0.0;
        for (int i = 0; i < rows_per_block; ++i) { a += sh[i * cols + tid]; b += sh[blockDim.x + i * cols + tid]; }
        atomicAdd(&stats[tid], a);
        atomicAdd(&stats[cols + tid], b);
    }
    if (blockIdx.x == 0 && tid == 0) atomicAdd(&stats[2 * cols], (double)n_rows);
}

__global__ __launch_bounds__(256) void k_cmvn_apply(float* __restrict__ x, int64_t n_rows, int n_mfcc,
                                                    const double* __restrict__ stats, double eps) {
    const int64_t total = n_rows * n_mfcc;
    const double cnt = stats[2 * n_mfcc];
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % n_mfcc);
        const double mean = stats[c] / cnt;
        double var = stats[n_mfcc + c] / cnt - mean * mean;
        if (var < eps) var = eps;
        x[i] = (float)(((double)x[i] - mean) / sqrt(var));
    }
}

template <typename T>
int upload(const std::vector<T>& h, T** d) {
    *d = nullptr;
    if (h.empty()) return MFCC_OK;
    HIP_TRY(hipMalloc((void**)d, h.size() * sizeof(T)));
    HIP_TRY(hipMemcpy(*d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
    return MFCC_OK;
}

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

struct WsLayout {
    size_t hdr, tile_off, gmax, tiles, mel, total;
    int64_t ub_tiles;
};

WsLayout ws_layout(int64_t total_frames, int32_t n_utts, int n_mels, bool need_mel) {
    WsLayout w;
    w.ub_tiles = total_frames / TILE_FRAMES + n_utts;
    size_t off = 0;
    w.hdr = off; off += align_up(sizeof(WsHeader), 256);
    w.tile_off = off; off += align_up(((size_t)n_utts + 1) * sizeof(int32_t), 256);
    w.gmax = off; off += align_up((size_t)std::max(n_utts, 1) * sizeof(uint32_t), 256);
    w.tiles = off; off += align_up((size_t)std::max<int64_t>(w.ub_tiles, 1) * sizeof(Tile), 256);
    w.mel = off;
    if (need_mel) off += align_up((size_t)total_frames * n_mels * sizeof(float), 256);
    w.total = off;
    return w;
}

}  // namespace

// ------------------------------------------------------------------------------------------
// Plan
// ------------------------------------------------------------------------------------------
struct FrontVariant {
    void (*fn)(const FrontArgs);
    int waves;
    bool win_regs, w64_regs, melw_lds;
};

struct mfcc_plan {
    mfcc_host::Resolved r;
    FrontVariant front{};
    int device = 0;
    int n_cu = 0;
    bool fast2048 = false;
    int variant = 0;
    size_t front_lds_bytes = 0;
    mfcc_host::SparseMel sm;
    int n_rt = 0, n_mels_pad = 0;
    float ref_db = 0.0f;
    // device tables
    float* d_window = nullptr;
    float2* d_tw1 = nullptr;
    float2* d_w64 = nullptr;
    float2* d_tw128 = nullptr;
    float4* d_melw = nullptr;
    int32_t* d_mel_start = nullptr;
    int32_t* d_mel_filt = nullptr;
    MelPass* d_mel_pass = nullptr;
    float* d_dct_frag = nullptr;
    float* d_lifter = nullptr;
    // optional per-kernel timing (mfcc_plan_set_profiling)
    bool profiling = false;
    hipEvent_t ev[3] = {nullptr, nullptr, nullptr};  // before front, after front, after dB+DCT
};

namespace {

// Front-end kernel variants: wavefronts per workgroup (one workgroup per CU; LDS-limited) and
// where the per-lane constant tables live.  MFCC_HIP_VARIANT selects one for experiments.
const FrontVariant kFrontVariants[] = {
    {k_melspec_2048<12, false, false, false, 0>, 12, false, false, false},
    {k_melspec_2048<12, true, true, true, 0>, 12, true, true, true},
    {k_melspec_2048<12, false, false, true, 0>, 12, false, false, true},
    {k_melspec_2048<12, true, false, true, 0>, 12, true, false, true},
    {k_melspec_2048<8, true, true, true, 0>, 8, true, true, true},
    {k_melspec_2048<8, false, false, true, 0>, 8, false, false, true},
};
// power != 2: same structure as variants 0 (mel weights streamed) and 2 (mel weights in LDS).
const FrontVariant kFrontGeneralPower[] = {
    {k_melspec_2048<12, false, false, false, 1>, 12, false, false, false},
    {k_melspec_2048<12, false, false, true, 1>, 12, false, false, true},
};
constexpr int kNumFrontVariants = sizeof(kFrontVariants) / sizeof(kFrontVariants[0]);
constexpr int kDefaultFrontVariant = 2;

int select_variant(mfcc_plan* p) {
    int v = kDefaultFrontVariant;
    if (const char* e = std::getenv("MFCC_HIP_VARIANT")) v = std::atoi(e);
    if (v < 0 || v >= kNumFrontVariants) return fail(MFCC_E_INVALID, "MFCC_HIP_VARIANT out of range");
    const int melw_floats = p->sm.total_rounds * 64;
    auto bytes = [&](int i) {
        const FrontVariant& f = kFrontVariants[i];
        return k2048::lds_bytes(f.waves, f.win_regs, f.w64_regs, f.melw_lds ? melw_floats : 0);
    };
    if (bytes(v) > (size_t)k2048::LDS_LIMIT) v = 0;  // mel weights too large for LDS: stream them
    if (bytes(v) > (size_t)k2048::LDS_LIMIT) return fail(MFCC_E_UNSUPPORTED, "LDS budget exceeded");
    p->front = kFrontVariants[v];
    if (p->r.c.power != 2.0f) p->front = kFrontGeneralPower[p->front.melw_lds ? 1 : 0];
    p->variant = v;
    p->front_lds_bytes = k2048::lds_bytes(p->front.waves, p->front.win_regs, p->front.w64_regs,
                                         p->front.melw_lds ? melw_floats : 0);
    hipError_t e2 = hipFuncSetAttribute(reinterpret_cast<const void*>(p->front.fn),
                                        hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)p->front_lds_bytes);
    if (e2 != hipSuccess) return fail(MFCC_E_HIP, std::string("hipFuncSetAttribute: ") + hipGetErrorString(e2));
    return MFCC_OK;
}

}  // namespace

extern "C" {

void mfcc_config_default(mfcc_config* c) {
    if (!c) return;
    std::memset(c, 0, sizeof(*c));
    c->struct_size = (int32_t)sizeof(mfcc_config);
    c->n_fft = 2048; c->hop_length = 512; c->win_length = 0; c->center = 1;
    c->pad_mode = MFCC_PAD_CONSTANT; c->power = 2.0f;
    c->sr = 22050.0; c->n_mels = 128; c->fmin = 0.0; c->fmax = 0.0; c->htk = 0;
    c->mel_norm = MFCC_MEL_NORM_SLANEY;
    c->ref = 1.0f; c->amin = 1e-10f; c->top_db = 80.0f;
    c->n_mfcc = 20; c->dct_type = 2; c->dct_norm = MFCC_DCT_NORM_ORTHO; c->lifter = 0.0f;
    c->window = nullptr; c->mel_basis = nullptr;
}

int mfcc_abi_version(void) { return MFCC_ABI_VERSION; }

const char* mfcc_last_error(void) { return g_err.c_str(); }

int mfcc_host_window(const mfcc_config* cfg, double* h_out) {
    mfcc_host::Resolved r;
    const std::string e = mfcc_host::resolve(cfg, &r);
    if (!e.empty()) return fail(MFCC_E_INVALID, e);
    if (!h_out) return fail(MFCC_E_INVALID, "h_out is NULL");
    std::vector<double> w;
    mfcc_host::build_window(r, w);
    std::memcpy(h_out, w.data(), w.size() * sizeof(double));
    return MFCC_OK;
}

int mfcc_host_mel_basis(const mfcc_config* cfg, float* h_out) {
    mfcc_host::Resolved r;
    const std::string e = mfcc_host::resolve(cfg, &r);
    if (!e.empty()) return fail(MFCC_E_INVALID, e);
    if (!h_out) return fail(MFCC_E_INVALID, "h_out is NULL");
    std::vector<float> m;
    mfcc_host::build_mel_basis(r, m);
    std::memcpy(h_out, m.data(), m.size() * sizeof(float));
    return MFCC_OK;
}

int mfcc_host_dct_basis(const mfcc_config* cfg, float* h_out) {
    mfcc_host::Resolved r;
    const std::string e = mfcc_host::resolve(cfg, &r);
    if (!e.empty()) return fail(MFCC_E_INVALID, e);
    if (!h_out) return fail(MFCC_E_INVALID, "h_out is NULL");
    std::vector<float> d;
    mfcc_host::build_dct_basis(r, d);
    std::memcpy(h_out, d.data(), d.size() * sizeof(float));
    return MFCC_OK;
}

int64_t mfcc_num_frames(const mfcc_config* cfg, int64_t n_samples) {
    mfcc_host::Resolved r;
    const std::string e = mfcc_host::resolve(cfg, &r);
    if (!e.empty()) return fail(MFCC_E_INVALID, e);
    const int64_t F = mfcc_host::num_frames(r, n_samples);
    if (F < 0) return fail(MFCC_E_INVALID, "n_fft is too large for an uncentered signal of this length");
    return F;
}

int mfcc_batch_layout(const mfcc_config* cfg, const int64_t* h_sample_offsets, int32_t n_utts,
                      int64_t* h_frame_offsets) {
    mfcc_host::Resolved r;
    const std::string e = mfcc_host::resolve(cfg, &r);
    if (!e.empty()) return fail(MFCC_E_INVALID, e);
    if (n_utts < 0 || !h_sample_offsets || !h_frame_offsets)
        return fail(MFCC_E_INVALID, "bad batch arguments");
    h_frame_offsets[0] = 0;
    for (int u = 0; u < n_utts; ++u) {
        const int64_t T = h_sample_offsets[u + 1] - h_sample_offsets[u];
        if (T < 0) return fail(MFCC_E_INVALID, "sample offsets must be non-decreasing");
        if (T > 0x7fffffffLL) return fail(MFCC_E_UNSUPPORTED, "utterances of 2^31 samples or more");
        const int64_t F = mfcc_host::num_frames(r, T);
        if (F < 0)
            return fail(MFCC_E_INVALID, "n_fft is too large for uncentered analysis of utterance " +
                                            std::to_string(u));
        h_frame_offsets[u + 1] = h_frame_offsets[u] + F;
    }
    return MFCC_OK;
}

int mfcc_plan_create(const mfcc_config* cfg, mfcc_plan** out_plan) {
    if (!out_plan) return fail(MFCC_E_INVALID, "out_plan is NULL");
    *out_plan = nullptr;
    mfcc_host::Resolved r;
    const std::string e = mfcc_host::resolve(cfg, &r);
    if (!e.empty()) return fail(MFCC_E_INVALID, e);
    if (r.c.n_fft != k2048::NFFT)
        return fail(MFCC_E_UNSUPPORTED, "this build has kernels for n_fft = 2048 only");

    mfcc_plan* p = new (std::nothrow) mfcc_plan();
    if (!p) return fail(MFCC_E_NOMEM, "out of host memory");
    p->r = r;
    p->r.c.window = nullptr;      // the copies below are what the plan keeps
    p->r.c.mel_basis = nullptr;
    p->fast2048 = true;

    hipError_t he = hipGetDevice(&p->device);
    hipDeviceProp_t prop;
    if (he == hipSuccess) he = hipGetDeviceProperties(&prop, p->device);
    if (he != hipSuccess) {
        delete p;
        return fail(MFCC_E_HIP, std::string("no usable HIP device: ") + hipGetErrorString(he));
    }
    p->n_cu = prop.multiProcessorCount;

    std::vector<double> win;
    mfcc_host::build_window(r, win);
    std::vector<float> winf(win.begin(), win.end());

    const double PI = 3.14159265358979323846;
    std::vector<float2> tw1(15 * 64), w64(64), tw128(64);
    for (int k2 = 1; k2 < 16; ++k2)
        for (int n1 = 0; n1 < 64; ++n1) {
            const double ang = -2.0 * PI * (double)(n1 * k2) / 2048.0;
            const double sc = (k2 == 8) ? 1.0 : 0.5;  // rdft32 leaves 2 A[k] except at k = 8
            tw1[(k2 - 1) * 64 + n1] = make_float2((float)(sc * std::cos(ang)), (float)(sc * std::sin(ang)));
        }
    for (int q = 0; q < 4; ++q)
        for (int ka = 0; ka < 16; ++ka) {
            const double ang = -2.0 * PI * (double)(q * ka) / 64.0;
            w64[q * 16 + ka] = make_float2((float)std::cos(ang), (float)std::sin(ang));
        }
    for (int j = 0; j < 64; ++j) {
        const double ang = -2.0 * PI * (double)j / 128.0;
        tw128[j] = make_float2((float)std::cos(ang), (float)std::sin(ang));
    }

    std::vector<float> basis;
    mfcc_host::build_mel_basis(r, basis);
    mfcc_host::build_sparse_mel(basis, r.c.n_mels, r.n_bins, k2048::MEL_CHUNK, p->sm);
    if (p->sm.n_pass > k2048::MAX_MEL_PASS) {
        delete p;
        return fail(MFCC_E_UNSUPPORTED, "n_mels > 256 is not supported");
    }
    if (p->sm.max_read >= k2048::XCH_FLOATS) {
        delete p;
        return fail(MFCC_E_UNSUPPORTED, "mel filter support too wide for the LDS staging region");
    }
    std::vector<MelPass> passes(p->sm.n_pass);
    for (int i = 0; i < p->sm.n_pass; ++i) passes[i] = MelPass{p->sm.rounds[i], p->sm.round_off[i]};

    std::vector<float> dct, lifter;
    mfcc_host::build_dct_basis(r, dct);
    mfcc_host::build_lifter(r, lifter);
    p->n_rt = (r.c.n_mfcc + 15) / 16;
    p->n_mels_pad = (r.c.n_mels + 63) / 64 * 64;
    const int steps = p->n_mels_pad / 4;
    std::vector<float> frag((size_t)p->n_rt * steps * 64, 0.0f);
    for (int rt = 0; rt < p->n_rt; ++rt)
        for (int s = 0; s < steps; ++s)
            for (int l = 0; l < 64; ++l) {
                const int row = rt * 16 + (l & 15), kq = l >> 4;
                const int k = 16 * (s >> 2) + 4 * kq + (s & 3);
                if (row < r.c.n_mfcc && k < r.c.n_mels)
                    frag[((size_t)rt * steps + s) * 64 + l] = dct[(size_t)row * r.c.n_mels + k];
            }
    p->ref_db = (float)(10.0 * std::log10(std::max((double)r.c.amin, std::fabs((double)r.c.ref))));

    int rc = MFCC_OK;
    if (rc == MFCC_OK) rc = upload(winf, &p->d_window);
    if (rc == MFCC_OK) rc = upload(tw1, &p->d_tw1);
    if (rc == MFCC_OK) rc = upload(w64, &p->d_w64);
    if (rc == MFCC_OK) rc = upload(tw128, &p->d_tw128);
    if (rc == MFCC_OK) {
        std::vector<float4> w4(p->sm.w.size() / 4);
        for (size_t i = 0; i < w4.size(); ++i)
            w4[i] = make_float4(p->sm.w[4 * i], p->sm.w[4 * i + 1], p->sm.w[4 * i + 2], p->sm.w[4 * i + 3]);
        rc = upload(w4, &p->d_melw);
    }
    if (rc == MFCC_OK) rc = upload(p->sm.start, &p->d_mel_start);
    if (rc == MFCC_OK) rc = upload(p->sm.filt, &p->d_mel_filt);
    if (rc == MFCC_OK) rc = upload(passes, &p->d_mel_pass);
    if (rc == MFCC_OK) rc = upload(frag, &p->d_dct_frag);
    if (rc == MFCC_OK) rc = upload(lifter, &p->d_lifter);
    if (rc == MFCC_OK) rc = select_variant(p);
    if (rc != MFCC_OK) {
        const std::string keep = g_err;
        mfcc_plan_destroy(p);
        g_err = keep;
        return rc;
    }
    *out_plan = p;
    return MFCC_OK;
}

void mfcc_plan_destroy(mfcc_plan* p) {
    if (!p) return;
    (void)hipFree(p->d_window); (void)hipFree(p->d_tw1); (void)hipFree(p->d_w64);
    (void)hipFree(p->d_tw128); (void)hipFree(p->d_melw); (void)hipFree(p->d_mel_start);
    (void)hipFree(p->d_mel_filt); (void)hipFree(p->d_mel_pass); (void)hipFree(p->d_dct_frag);
    (void)hipFree(p->d_lifter);
    for (int i = 0; i < 3; ++i) if (p->ev[i]) (void)hipEventDestroy(p->ev[i]);
    delete p;
}

size_t mfcc_workspace_bytes(const mfcc_plan* plan, int64_t total_frames, int32_t n_utts) {
    if (!plan || total_frames < 0 || n_utts < 0) return 0;
    return ws_layout(total_frames, n_utts, plan->r.c.n_mels, true).total;
}

static int launch_front(const mfcc_plan* plan, const float* d_pcm, const int64_t* d_sample_offsets,
                        const int64_t* d_frame_offsets, const int32_t* d_group, int32_t n_utts,
                        int64_t total_frames, float* mel_out, float* spec_out, char* ws,
                        const WsLayout& L, hipStream_t st) {
    WsHeader* hdr = reinterpret_cast<WsHeader*>(ws + L.hdr);
    int32_t* tile_off = reinterpret_cast<int32_t*>(ws + L.tile_off);
    uint32_t* gmax = reinterpret_cast<uint32_t*>(ws + L.gmax);
    Tile* tiles = reinterpret_cast<Tile*>(ws + L.tiles);

    hipLaunchKernelGGL(k_prepare_scan, dim3(1), dim3(1024), 0, st, d_frame_offsets, n_utts, tile_off,
                       gmax, hdr);
    const int pt_blocks = (int)((L.ub_tiles + 255) / 256);
    hipLaunchKernelGGL(k_prepare_tiles, dim3(pt_blocks), dim3(256), 0, st, d_sample_offsets,
                       d_frame_offsets, d_group, tile_off, n_utts, tiles);

    FrontArgs fa;
    fa.pcm = d_pcm; fa.tiles = tiles; fa.hdr = hdr;
    fa.window = plan->d_window; fa.tw1 = plan->d_tw1; fa.w64 = plan->d_w64; fa.tw128 = plan->d_tw128;
    fa.melw = plan->d_melw; fa.mel_start = plan->d_mel_start; fa.mel_filt = plan->d_mel_filt;
    fa.mel_pass = plan->d_mel_pass;
    fa.mel_out = mel_out; fa.spec_out = spec_out; fa.gmax = gmax;
    fa.n_pass = plan->sm.n_pass; fa.n_mels = plan->r.c.n_mels;
    fa.hop = plan->r.c.hop_length;
    fa.pad = plan->r.c.center ? plan->r.c.n_fft / 2 : 0;
    fa.reflect = plan->r.c.pad_mode == MFCC_PAD_REFLECT;
    fa.ablate = std::getenv("MFCC_HIP_ABLATE") ? std::atoi(std::getenv("MFCC_HIP_ABLATE")) : 0;
    fa.pw_mode = plan->r.c.power == 2.0f ? 0 : (plan->r.c.power == 1.0f ? 1 : 2);
    fa.half_power = 0.5f * plan->r.c.power;
    fa.total_rounds = plan->sm.total_rounds;
    const FrontVariant& fv = plan->front;
    const int64_t wg_needed = (L.ub_tiles * TILE_FRAMES + fv.waves - 1) / fv.waves;
    const int grid = (int)std::max<int64_t>(1, std::min<int64_t>(wg_needed, plan->n_cu));
    if (plan->profiling) HIP_TRY(hipEventRecord(plan->ev[0], st));
    hipLaunchKernelGGL(fv.fn, dim3(grid), dim3(fv.waves * 64), plan->front_lds_bytes, st, fa);
    if (plan->profiling) HIP_TRY(hipEventRecord(plan->ev[1], st));
    (void)total_frames;
    HIP_TRY(hipGetLastError());
    return MFCC_OK;
}

static int check_batch(const mfcc_plan* plan, const void* d_pcm, const void* so, const void* fo,
                       int32_t n_utts, int64_t total_frames) {
    if (!plan) return fail(MFCC_E_INVALID, "plan is NULL");
    if (n_utts < 0 || total_frames < 0) return fail(MFCC_E_INVALID, "negative batch size");
    if (n_utts > 0 && (!so || !fo)) return fail(MFCC_E_INVALID, "offset arrays are NULL");
    if (total_frames > 0 && !d_pcm && plan->r.c.center == 0)
        return fail(MFCC_E_INVALID, "d_pcm is NULL");
    if (total_frames / TILE_FRAMES + n_utts > 0x7fffffffLL)
        return fail(MFCC_E_UNSUPPORTED, "batch too large: split it");
    return MFCC_OK;
}

int mfcc_execute(const mfcc_plan* plan, const float* d_pcm, const int64_t* d_sample_offsets,
                 const int64_t* d_frame_offsets, const int32_t* d_group, int32_t n_utts,
                 int64_t total_frames, float* d_out, float* d_melspec_out, void* d_workspace,
                 size_t workspace_bytes, void* stream) {
    int rc = check_batch(plan, d_pcm, d_sample_offsets, d_frame_offsets, n_utts, total_frames);
    if (rc != MFCC_OK) return rc;
    if (n_utts == 0 || total_frames == 0) return MFCC_OK;
    if (!d_out) return fail(MFCC_E_INVALID, "d_out is NULL");
    const bool need_mel = d_melspec_out == nullptr;
    const WsLayout L = ws_layout(total_frames, n_utts, plan->r.c.n_mels, need_mel);
    if (!d_workspace || workspace_bytes < L.total)
        return fail(MFCC_E_WORKSPACE, "workspace too small: need " + std::to_string(L.total) + " bytes");
    if (reinterpret_cast<uintptr_t>(d_workspace) % 256)
        return fail(MFCC_E_INVALID, "workspace must be 256-byte aligned");
    hipStream_t st = static_cast<hipStream_t>(stream);
    char* ws = static_cast<char*>(d_workspace);
    float* mel = need_mel ? reinterpret_cast<float*>(ws + L.mel) : d_melspec_out;

    rc = launch_front(plan, d_pcm, d_sample_offsets, d_frame_offsets, d_group, n_utts, total_frames,
                      mel, nullptr, ws, L, st);
    if (rc != MFCC_OK) return rc;

    DbDctArgs da;
    da.tiles = reinterpret_cast<const Tile*>(ws + L.tiles);
    da.hdr = reinterpret_cast<const WsHeader*>(ws + L.hdr);
    da.mel = mel;
    da.gmax = reinterpret_cast<const uint32_t*>(ws + L.gmax);
    da.dct_frag = plan->d_dct_frag; da.lifter = plan->d_lifter; da.out = d_out;
    da.n_mels = plan->r.c.n_mels; da.n_mels_pad = plan->n_mels_pad; da.n_mfcc = plan->r.c.n_mfcc;
    da.n_rt = plan->n_rt;
    da.vec_ok = (plan->r.c.n_mels % 4 == 0) && (reinterpret_cast<uintptr_t>(mel) % 16 == 0);
    da.amin = plan->r.c.amin; da.ref_db = plan->ref_db; da.top_db = plan->r.c.top_db;
    const int blocks = (int)((L.ub_tiles + 3) / 4);
    hipLaunchKernelGGL(k_db_dct, dim3(blocks), dim3(256), 0, st, da);
    if (plan->profiling) HIP_TRY(hipEventRecord(plan->ev[2], st));
    HIP_TRY(hipGetLastError());
    return MFCC_OK;
}

int mfcc_plan_set_profiling(mfcc_plan* plan, int enable) {
    if (!plan) return fail(MFCC_E_INVALID, "plan is NULL");
    if (enable && !plan->ev[0])
        for (int i = 0; i < 3; ++i) HIP_TRY(hipEventCreate(&plan->ev[i]));
    plan->profiling = enable != 0;
    return MFCC_OK;
}

int mfcc_plan_last_timings(const mfcc_plan* plan, float* ms_front, float* ms_db_dct) {
    if (!plan || !plan->ev[0]) return fail(MFCC_E_INVALID, "profiling was not enabled on this plan");
    HIP_TRY(hipEventSynchronize(plan->ev[2]));
    if (ms_front) HIP_TRY(hipEventElapsedTime(ms_front, plan->ev[0], plan->ev[1]));
    if (ms_db_dct) HIP_TRY(hipEventElapsedTime(ms_db_dct, plan->ev[1], plan->ev[2]));
    return MFCC_OK;
}

int mfcc_execute_spectrogram(const mfcc_plan* plan, const float* d_pcm,
                             const int64_t* d_sample_offsets, const int64_t* d_frame_offsets,
                             int32_t n_utts, int64_t total_frames, float* d_spec_out,
                             void* d_workspace, size_t workspace_bytes, void* stream) {
    int rc = check_batch(plan, d_pcm, d_sample_offsets, d_frame_offsets, n_utts, total_frames);
    if (rc != MFCC_OK) return rc;
    if (n_utts == 0 || total_frames == 0) return MFCC_OK;
    if (!d_spec_out) return fail(MFCC_E_INVALID, "d_spec_out is NULL");
    const WsLayout L = ws_layout(total_frames, n_utts, plan->r.c.n_mels, false);
    if (!d_workspace || workspace_bytes < L.total)
        return fail(MFCC_E_WORKSPACE, "workspace too small: need " + std::to_string(L.total) + " bytes");
    if (reinterpret_cast<uintptr_t>(d_workspace) % 256)
        return fail(MFCC_E_INVALID, "workspace must be 256-byte aligned");
    return launch_front(plan, d_pcm, d_sample_offsets, d_frame_offsets, nullptr, n_utts, total_frames,
                        nullptr, d_spec_out, static_cast<char*>(d_workspace), L,
                        static_cast<hipStream_t>(stream));
}

int mfcc_cmvn_accumulate(const float* d_feats, int64_t n_rows, int32_t n_mfcc, double* d_stats,
                         void* stream) {
    if (n_rows < 0 || n_mfcc < 1 || n_mfcc > 256) return fail(MFCC_E_INVALID, "bad CMVN shape");
    if (n_rows == 0) return MFCC_OK;
    if (!d_feats || !d_stats) return fail(MFCC_E_INVALID, "NULL CMVN buffer");
    const int threads = 256;
    const int rpb = threads / n_mfcc;
    const int64_t want = (n_rows + rpb - 1) / rpb;
    const int blocks = (int)std::max<int64_t>(1, std::min<int64_t>(want, 2048));
    hipLaunchKernelGGL(k_cmvn_accumulate, dim3(blocks), dim3(threads), 2 * threads * sizeof(double),
                       static_cast<hipStream_t>(stream), d_feats, n_rows, n_mfcc, d_stats);
    HIP_TRY(hipGetLastError());
    return MFCC_OK;
}

int mfcc_cmvn_apply(float* d_feats, int64_t n_rows, int32_t n_mfcc, const double* d_stats, double eps,
                    void* stream) {
    if (n_rows < 0 || n_mfcc < 1) return fail(MFCC_E_INVALID, "bad CMVN shape");
    if (n_rows == 0) return MFCC_OK;
    if (!d_feats || !d_stats) return fail(MFCC_E_INVALID, "NULL CMVN buffer");
    const int64_t total = n_rows * n_mfcc;
    const int blocks = (int)std::max<int64_t>(1, std::min<int64_t>((total + 255) / 256, 4096));
    hipLaunchKernelGGL(k_cmvn_apply, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), d_feats,
                       n_rows, n_mfcc, d_stats, eps);
    HIP_TRY(hipGetLastError());
    return MFCC_OK;
}

}  // extern "C"
