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
//  mel     the power spectrum is written to the wavefront's LDS region in bin order (rows of 64
//          bins); every lane runs the same number of jobs, each a dot product of 12 consecutive
//          bins with one filter's weights, and each filter's partial sums are added up
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "fft_core.h"

namespace k2048 {
constexpr int NFFT = 2048;
constexpr int NBIN = 1025;
constexpr int TILE_FRAMES = 16;
// Exchange: 16 columns of 64 complex, no padding (8 KB per wavefront, so that 16 wavefronts and
// the tables fit the CU's 160 KB).  Step 2 reads a column two complex (16 bytes) at a time; to
// keep those ds_read_b128 bank-conflict free the pairs of a column are stored swizzled: pair
// 8 q + p (q = 0..3, p = 0..7) of column k2 sits at pair slot 8 q + (p ^ g), g = 2 (k2 & 3) +
// (q >> 1).  The 16 lanes the hardware services together then hit 16 different bank groups,
// and the ds_write_b64 of step 1 stay a permutation within the column (conflict free).
constexpr int COL_STRIDE = 64;
constexpr int XCH_FLOATS = 16 * COL_STRIDE * 2;   // 2048 floats per wavefront
__host__ __device__ constexpr int xch_slot(int k2, int c) {   // complex index of element c of column k2
    return k2 * COL_STRIDE + 16 * (c >> 4) + 2 * (((c >> 1) & 7) ^ (2 * (k2 & 3) + (c >> 5))) + (c & 1);
}
// After step 2 the same region holds, per wavefront:
//   the power spectrum in rows of 64 bins, 80 floats apart: bin k at k + 16 (k >> 6).  With the
//   16 floats of padding the 64 lanes of one power-spectrum store (bins 32 kb + k2c + 128 ka)
//   hit 64 different banks.
//   the column-0 scratch (4 rows of 24 complex)
//   the mel cells: rows of 64 partial sums (MelJobs in host_tables.h), the last row for idle jobs
constexpr int P_ROW_BINS = 64;
constexpr int P_ROW = 80;
constexpr int P_END = 16 * P_ROW + 16;            // bin 1024 sits at 1280; reads may run on to here
constexpr int SPECIAL_OFF = P_END;                // float offset of the column-0 scratch
constexpr int SPECIAL_ROW = 24;                   // complex per kb row of that scratch
constexpr int MCELL_OFF = SPECIAL_OFF + 4 * SPECIAL_ROW * 2;
constexpr int MCELL_ROWS = (XCH_FLOATS - MCELL_OFF) / 64;   // 8, one of them for idle jobs
static_assert(MCELL_ROWS >= 5, "per-wavefront LDS region too small");
constexpr int MEL_J = 3;                          // quads (of 4 bins) per mel job
constexpr int MEL_REG_JOBS = 4;                   // jobs per lane that are kept in registers
#ifndef MFCC_AB_MEL_BATCH
#define MFCC_AB_MEL_BATCH 2
#endif
constexpr int MEL_BATCH = MFCC_AB_MEL_BATCH;      // jobs whose reads are in flight together (2 or 4)
constexpr int MEL_MAX_ROWS = 4;                   // cell rows per group of 64 filters
constexpr int TW1_FLOATS = 15 * 64 * 2;
constexpr int W64_STRIDE = 18;                    // complex per kb row (16 + 2 pad)
constexpr int W64_FLOATS = 4 * W64_STRIDE * 2;
constexpr int WIN_FLOATS = 2048;
constexpr int LDS_LIMIT = 160 * 1024;
constexpr int MAX_MEL_PASS = 4;                   // n_mels <= 256
constexpr int MEL_CHUNK = 4;                      // generic kernel: ds_read_b128 pairs in flight

enum TableHome { IN_LDS = 0, IN_REGS = 1, IN_GLOBAL = 2 };

// LDS bytes of a variant: per-wavefront exchange regions, then the tables that live in LDS.
// melw_floats: mel weights + job table (2 ints per job and lane) when they live in LDS, else 0
constexpr size_t lds_bytes(int waves, int win, int tw1, bool w64_regs, int melw_floats) {
    return sizeof(float) * ((size_t)waves * XCH_FLOATS + (tw1 == IN_LDS ? TW1_FLOATS : 0) +
                            (w64_regs ? 0 : W64_FLOATS) + (win == IN_LDS ? WIN_FLOATS : 0) +
                            (size_t)melw_floats + 16 /* slot counters */);
}
__host__ __device__ constexpr int p_index(int bin) { return bin + (P_ROW - P_ROW_BINS) * (bin >> 6); }
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
    int32_t bad_groups;   // utterances whose top_db group id was outside [0, n_utts)
    int32_t pad[62];
};

struct MelPass {
    int32_t rounds;      // bins each lane reads in this pass (multiple of 4 * MEL_CHUNK)
    int32_t round_off;   // first round of this pass in the weight table
};

struct FrontArgs {
    const void* pcm;         // float32 samples, or int16 PCM (value / 32768) when pcm16 is set
    const Tile* tiles;
    const WsHeader* hdr;
    const float* window;     // 2048
    const float4* window_t;  // the window lane-major: float 32 lane + n2 = window[lane + 64 n2]
    const float2* tw1;       // 15 x 64, scaled
    const float4* tw1_t;     // step-1 twiddles lane-major: complex 16 lane + k2 - 1, k2 = 1..15
    const float2* w64;       // 4 x 16
    const float2* tw128;     // 64
    const float4* melw;      // (n_jobs * MEL_J) x 64 quads of weights
    const int2* mel_jobs;    // n_jobs x 64: byte offsets, within the wavefront's LDS region, of
                             // the job's first quad (.x) and of its cell (.y); bit 0 of .y: the
                             // job continues the chain of the previous slot
    float* mel_out;          // (rows, n_mels) or nullptr
    float* spec_out;         // (rows, 1025) or nullptr
    uint32_t* gmax;
    int32_t n_jobs;
    int32_t n_mels;
    int32_t mel_rows[4];     // cell rows per group of 64 filters
    uint32_t mel_zero_rows;  // cell rows to clear before each frame's jobs
    int32_t mel_shared;      // the chains of a filter add into one cell (LDS atomics)
    int32_t hop;
    int32_t pad;             // n_fft/2 if center else 0
    int32_t reflect;
    int32_t pw_mode;         // 0: power 2, 1: power 1, 2: general
    int32_t pcm16;           // pcm holds int16 samples
    int32_t share;           // wavefronts that walk one run of frames together (1, 2, 4, 8 or 16)
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

// v[i] += s * (v[i] of the lane's quad partner), eight registers per call: one v_fmac_f32 with a
// DPP source operand each (the compiler only forms v_mov_b32_dpp + v_fma_f32).  CTRL 2: partner
// lane ^ 2, CTRL 1: partner lane ^ 1.  A DPP operand may not be read within two wait states of
// the VALU instruction that wrote it; the leading s_nop covers whatever precedes the block.
#define MFCC_DPP_FMAC8(PERM)                                                                       \
    asm("s_nop 1\n\t"                                                                             \
        "v_fmac_f32_dpp %0, %0, %8 " PERM " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"          \
        "v_fmac_f32_dpp %1, %1, %8 " PERM " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"          \
        "v_fmac_f32_dpp %2, %2, %8 " PERM " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"          \
        "v_fmac_f32_dpp %3, %3, %8 " PERM " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"          \
        "v_fmac_f32_dpp %4, %4, %8 " PERM " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"          \
        "v_fmac_f32_dpp %5, %5, %8 " PERM " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"          \
        "v_fmac_f32_dpp %6, %6, %8 " PERM " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"          \
        "v_fmac_f32_dpp %7, %7, %8 " PERM " row_mask:0xf bank_mask:0xf bound_ctrl:1"               \
        : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]),     \
          "+v"(v[7])                                                                               \
        : "v"(s))
template <int CTRL>
__device__ __forceinline__ void quad_fmac8(float* v, float s) {
    if (CTRL == 2) MFCC_DPP_FMAC8("quad_perm:[2,3,0,1]");
    else MFCC_DPP_FMAC8("quad_perm:[1,0,3,2]");
}

// Lanes of bank group BANKS (bit g: lanes 4 g .. 4 g + 3 of every row of 16) take `src` from the
// lane SHIFT below them in their row; the other lanes keep `old`.
template <int SHIFT, int BANKS>
__device__ __forceinline__ float dpp_row_shr(float old, float src) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, src),
                                                                 0x110 + SHIFT, 0xF, BANKS, false));
}

// |X|^power from |X|^2.  PW = 0: power 2 (the reference's); PW = 1: any other power, decided at
// run time (kept out of the power-2 instantiations: powf is ~200 instructions per call site).
template <int PW>
__device__ __forceinline__ float apply_power(float mag2, int pw_mode, float half_power) {
    if (PW == 0) return mag2;
    return pw_mode == 1 ? sqrtf(mag2) : powf(mag2, half_power);
}

// numpy's 'reflect' extension of the samples lo .. T-1 (the edge sample is not repeated; a pad
// longer than the slice reflects back and forth inside it), as an index into the utterance.
__device__ __forceinline__ int64_t reflect_index(int64_t i, int64_t T, int64_t lo = 0) {
    const int64_t L = T - lo;
    if (L <= 1) return lo;
    const int64_t period = 2 * (L - 1);
    int64_t m = (i - lo) % period;
    if (m < 0) m += period;
    return lo + (m < L ? m : period - m);
}

// librosa 0.10's stft pads the frames that reach past the end of the signal from the SLICE
// y[tail_k hop - n_fft/2 :] (right side only) whenever those frames are not also frames that touch
// the left padding; the left side and the overlapping case pad from the signal's start.  Returns
// the first sample of the slice the frame starting at `start` (= t hop - n_fft/2) reflects in: 0
// unless it is such a tail frame.  Only for hop >= n_fft/2 can the slice be shorter than the
// padding, and only then does this differ from reflecting the whole signal.
__device__ __forceinline__ int reflect_slice_start(int start, int T, int hop, int n_fft) {
    const int half = n_fft / 2;
    const int start_k = (half + hop - 1) / hop;           // frames that touch the left padding
    const int tail_k = (T + half - n_fft) / hop + 1;       // first frame that touches the right padding
    if (tail_k <= start_k) return 0;                       // (also whenever T + half < n_fft)
    const int lo = tail_k * hop - half;
    return start >= lo ? lo : 0;
}

// Wave-uniform read of a tile descriptor through the scalar cache (tiles are written by an
// earlier kernel and never by this one).
__device__ __forceinline__ Tile load_tile(const Tile* tiles, int idx) {
    typedef const Tile __attribute__((address_space(4))) ConstTile;
    const ConstTile* p = (const ConstTile*)(uintptr_t)(tiles + __builtin_amdgcn_readfirstlane(idx));
    Tile t;
    t.soff = p->soff; t.row0 = p->row0; t.T = p->T; t.f0 = p->f0; t.F = p->F; t.grp = p->grp;
    return t;
}

__device__ __forceinline__ void flush_max(float runmax, int grp, uint32_t* gmax, int lane) {
    if (grp < 0) return;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) runmax = fmaxf(runmax, __shfl_xor(runmax, o));
    if (lane == 0 && runmax > 0.0f) atomicMax(&gmax[grp], __float_as_uint(runmax));
}

// Edge frames (those that overlap the zero / reflect padding) are rare: stage them through the
// wavefront's LDS region with a rolled loop so the hot path stays small.
// One sample as float32: int16 PCM maps to value / 32768 (exact), the way torchaudio decodes
// LibriSpeech FLAC for /root/reference/src/preprocess.py:75.
constexpr float PCM16_SCALE = 1.0f / 32768.0f;
__device__ __forceinline__ float load_sample(const void* base, int64_t idx, bool pcm16) {
    return pcm16 ? PCM16_SCALE * (float)static_cast<const int16_t*>(base)[idx]
                 : static_cast<const float*>(base)[idx];
}

// `p`: first sample of the utterance.
__device__ __noinline__ void stage_edge_frame(float* X, const void* pcm, int64_t soff, bool pcm16, int start,
                                              int T, int reflect, int lane, int hop) {
    const int lo = reflect ? reflect_slice_start(start, T, hop, NFFT) : 0;
#pragma unroll 1
    for (int n2 = 0; n2 < 32; ++n2) {
        const int idx = start + lane + 64 * n2;
        float v = 0.0f;
        if (reflect) {
            if (T > 0) v = load_sample(pcm, soff + reflect_index(idx, T, lo), pcm16);
        } else if (idx >= 0 && idx < T) {
            v = load_sample(pcm, soff + idx, pcm16);
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
// PCM16     the input is int16 PCM (HOPR = 0 only)
template <int WAVES, int WIN, int TW1, bool W64_REGS, bool MELW_LDS, int HOPR, int PW, bool PCM16 = false>
__global__ __launch_bounds__(WAVES * 64) void k_melspec_2048(const FrontArgs a) {
    using namespace k2048;
    constexpr int THREADS = WAVES * 64;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // Tables first, per-wavefront regions after them: a DS instruction's immediate offset reaches
    // 64 KB, so every table read addresses off one per-lane register (tables behind the 128 KB of
    // regions needed a register per read: 15 for the step-1 twiddles alone).
    float* tab = lds;
    float2* tw1_l = reinterpret_cast<float2*>(tab);
    if (TW1 == IN_LDS) tab += TW1_FLOATS;
    float2* w64_l = reinterpret_cast<float2*>(tab);
    if (!W64_REGS) tab += W64_FLOATS;
    float* win_l = tab;
    if (WIN == IN_LDS) tab += WIN_FLOATS;
    float4* melw_l = reinterpret_cast<float4*>(tab);
    int2* jobs_l = reinterpret_cast<int2*>(melw_l + a.n_jobs * MEL_J * 64);
    // one slot counter per group of wavefronts that share a run (after the tables)
    int* run_ctr = MELW_LDS ? reinterpret_cast<int*>(jobs_l + a.n_jobs * 64) : reinterpret_cast<int*>(melw_l);
    float* regions = reinterpret_cast<float*>(run_ctr + 16);
    float* X = regions + wave * XCH_FLOATS;
    float2* X2 = reinterpret_cast<float2*>(X);

    for (int i = tid; i < WAVES * XCH_FLOATS; i += THREADS) regions[i] = 0.0f;
    if (TW1 == IN_LDS) { for (int i = tid; i < 15 * 64; i += THREADS) tw1_l[i] = a.tw1[i]; }
    // step-2 twiddles W_64^(m kb), m = 1..15, at slot m - 1 of row kb (m = 0 is 1 and never read):
    // the eight 16-byte reads of a row then carry only values that are used, so the compiler
    // keeps them as ds_read_b128
    if (!W64_REGS) {
        if (tid < 64) {
            const int m = tid & 15;
            w64_l[(tid >> 4) * W64_STRIDE + (m ? m - 1 : 15)] = m ? a.w64[tid] : make_float2(0.0f, 0.0f);
        }
    }
    // window in LDS as quads: float4 (j, lane) = w[lane + 64 (4 j + c)], c = 0..3, so that a lane's
    // 32 values come in 8 conflict-free ds_read_b128 (4 LDS cycles each; 32 ds_read_b32 take 2 each,
    // and pairs of ds_read_b64 get merged into half-rate ds_read2st64_b64)
    if (WIN == IN_LDS) {
        for (int i = tid; i < 512; i += THREADS) {
            const int j = i >> 6, l = i & 63;
            reinterpret_cast<float4*>(win_l)[i] = make_float4(a.window[l + 256 * j], a.window[l + 256 * j + 64],
                                                              a.window[l + 256 * j + 128], a.window[l + 256 * j + 192]);
        }
    }
    if (tid < WAVES) run_ctr[tid] = 0;
    if (MELW_LDS) {
        for (int i = tid; i < a.n_jobs * MEL_J * 64; i += THREADS) melw_l[i] = a.melw[i];
        if (a.n_jobs <= MEL_REG_JOBS) {
            // short job lists: one uint4 per lane, a job per component (first quad in the low
            // half, cell and chain bit in the high half), fetched with one read per frame
            if (tid < 64) {
                uint32_t pk[MEL_REG_JOBS];
#pragma unroll
                for (int j = 0; j < MEL_REG_JOBS; ++j) {
                    const int2 jb = a.mel_jobs[(j < a.n_jobs ? j : 0) * 64 + tid];
                    pk[j] = (uint32_t)jb.x | ((uint32_t)jb.y << 16);
                }
                reinterpret_cast<uint4*>(jobs_l)[tid] = make_uint4(pk[0], pk[1], pk[2], pk[3]);
            }
        } else {
            for (int i = tid; i < a.n_jobs * 64; i += THREADS) jobs_l[i] = a.mel_jobs[i];
        }
    }
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

    // Quad DFT-4 signs.  Each butterfly is own + sign * partner (one v_fmac with a DPP operand),
    // which leaves lanes 1..3 of the quad with their outputs negated; only |X|^2 is used.
    const float sgn1 = (q & 2) ? -1.0f : 1.0f;
    const float sgn2 = (q == 1 || q == 2) ? -1.0f : 1.0f;
    const bool lane3 = (q == 3);
    // exchange addressing (complex index c of column k2 lives at k2 * 72 + c + 2 * (c >> 4))
    // step-1 write of element `lane` of column k2: k2 * COL_STRIDE + xw[k2 & 3] (complex index)
    const int xw[4] = {xch_slot(0, lane), xch_slot(1, lane) - COL_STRIDE, xch_slot(2, lane) - 2 * COL_STRIDE,
                       xch_slot(3, lane) - 3 * COL_STRIDE};
    // column 0 as 128 reals y[n] = component n & 1 of complex n >> 1: y[lane] and y[64 + lane]
    const int y0_off = 2 * xch_slot(0, lane >> 1) + (lane & 1);
    const int y1_off = 2 * xch_slot(0, 32 + (lane >> 1)) + (lane & 1);
    // step-2 read of pairs 8 q + j, j = 0..7, of column k2c: float index xr_base ^ (4 j)
    const int xr_base = 2 * (k2c * COL_STRIDE + 16 * q) + 4 * (2 * (k2c & 3) + (q >> 1));
    // power spectrum addressing: ka < 8 -> bin 128 ka + 32 kb + k2c; ka >= 8 -> its mirror image
    // bin 128 (15 - ka) + 128 - 32 kb - k2c (k2c > 0 there); 128 bins = 2 rows = 2 P_ROW floats
    const int pb0 = p_index(32 * kb + k2c);
    const int pb1 = p_index(128 - 32 * kb - k2c);
    // column-0 scratch: C[j] at (j & 3) * SPECIAL_ROW + (j >> 2)
    const int sj = (lane & 3) * SPECIAL_ROW + (lane >> 2);
    const int sm = (((64 - lane) & 63) & 3) * SPECIAL_ROW + (((64 - lane) & 63) >> 2);

    const int n_jobs = a.n_jobs;
    const int n_acc = (a.n_mels + 63) >> 6;          // groups of 64 filters
    float runmax = 0.0f;
    int cur_grp = -1;
    const int n_tiles = a.hdr->n_tiles;

    // Work unit = one frame slot (tile, frame-in-tile).  The slots are cut into contiguous runs,
    // one per group of g = a.share neighbouring wavefronts (by default the whole workgroup; g = 1
    // with HOPR > 0).  The wavefronts
    // of a group take the slots of their run one at a time from a counter in LDS, so at any moment
    // they work on neighbouring frames: consecutive frames share 75 % of their samples, the group
    // keeps one window of samples warm in L2 instead of g, and a wavefront stays within one
    // utterance for many frames (one atomicMax per utterance change).  The next slot is requested
    // a frame ahead.  Every wavefront leaves the loop as soon as the counter passes the run's end.
    const int g = HOPR > 0 ? 1 : (a.share >= WAVES ? WAVES : (WAVES % a.share != 0 ? 1 : a.share));
    const int64_t n_slots = (int64_t)n_tiles * TILE_FRAMES;
    const int64_t n_runs = (int64_t)gridDim.x * WAVES / g;
    const int64_t per_run = (n_slots + n_runs - 1) / n_runs;
    const int64_t run_base = (((int64_t)blockIdx.x * WAVES + wave) / g) * per_run;
    const int count = run_base < n_slots ? (int)min(per_run, n_slots - run_base) : 0;
    int* my_ctr = run_ctr + wave / g;
    int next_static = 0;
    auto request_slot = [&]() -> int {     // index within the run; the same in all lanes once read
        if (g == 1) return next_static++;
        int v = 0;
        if (lane == 0) v = atomicAdd(my_ctr, 1);
        return v;
    };

    // Tile descriptors are read with scalar loads, one tile ahead of the frames being processed,
    // so the frame loop never waits on them (a vector load here would make every frame wait for
    // the previous frame's stores to be acknowledged).
    int tix = -2;
    Tile ti = {}, ti_next = {};
    auto is_interior = [&](const Tile& c, int tt) -> bool {
        const int start = tt * a.hop - a.pad;
        return start >= 0 && start + NFFT <= c.T;
    };

    // HOPR > 0 (hop = 64 * HOPR): consecutive frames of a run share their rows n2 >= HOPR, so the
    // unwindowed samples stay in registers: once a frame has been windowed they shift down by
    // HOPR rows and the HOPR new rows of the next frame are requested, a whole frame ahead.
    float raw[HOPR > 0 ? 32 : 1];
    bool raw_ok = false;               // raw[] already holds the samples of the frame about to run

    // Mel outputs of a frame: each group's cell rows summed and stored as one coalesced row.
    const float* cell0 = X + MCELL_OFF + lane;
    auto collect_outputs = [&](int64_t row) {
        const float* cells = cell0;
        float outv[MAX_MEL_PASS];
#pragma unroll
        for (int p = 0; p < MAX_MEL_PASS; ++p) {
            outv[p] = 0.0f;
            if (p < n_acc) {
                // rows of the group: nested so that all reads are issued before the first add
                const int R = a.mel_rows[p];
                float s0 = cells[0];
                if (R > 1) {
                    const float s1 = cells[64];
                    if (R > 2) {
                        const float s2 = cells[128];
                        if (R > 3) s0 += cells[192];
                        s0 += s2;
                    }
                    s0 += s1;
                }
                outv[p] = s0;
                cells += 64 * R;
            }
        }
        float* mo = a.mel_out + row * a.n_mels;
#pragma unroll
        for (int p = 0; p < MAX_MEL_PASS; ++p) {
            const int f = 64 * p + lane;
            if (p < n_acc && f < a.n_mels) {
                mo[f] = outv[p];
                runmax = fmaxf(runmax, outv[p]);
            }
        }
    };

    int next_v = 0;
    for (int cur = __builtin_amdgcn_readfirstlane(request_slot()); cur < count;
         cur = __builtin_amdgcn_readfirstlane(next_v)) {
        next_v = request_slot();
        const int64_t slot = run_base + cur;
        const int tidx = (int)(slot >> 4), fr = (int)(slot & 15);
        if (tidx != tix) {
            ti = (tidx == tix + 1) ? ti_next : load_tile(a.tiles, tidx);
            tix = tidx;
            ti_next = load_tile(a.tiles, min(tix + 1, n_tiles - 1));
        }
        const int t = ti.f0 + fr;
        const bool valid = t < ti.F;        // the last tile of an utterance may be partly empty
        // (HOPR > 0, so g = 1) does the next slot of this run hold frame t + 1 of the utterance?
        const bool next_same = cur + 1 < count &&
            (fr < TILE_FRAMES - 1 ? t + 1 < ti.F : (ti_next.soff == ti.soff && ti_next.f0 == t + 1));
        const Tile tc = ti;
        if (!valid) continue;
        const bool interior = is_interior(tc, t);
        if (tc.grp != cur_grp) {
            flush_max(runmax, cur_grp, a.gmax, lane);
            cur_grp = tc.grp;
            runmax = 0.0f;
        }
        const int64_t row = tc.row0 + (t - tc.f0);

        // ---- samples, window ----------------------------------------------------------------------
        float r[32];
        {
            float* src = HOPR > 0 ? raw : r;
            const int64_t s0 = tc.soff + (t * a.hop - a.pad) + lane;
            const float* pp = static_cast<const float*>(a.pcm) + s0;
            const int16_t* pp16 = static_cast<const int16_t*>(a.pcm) + s0;
            if (!interior) {
                // edge frames (overlapping the padding) come through LDS
                stage_edge_frame(X, a.pcm, tc.soff, PCM16, t * a.hop - a.pad, tc.T, a.reflect && a.pad, lane, a.hop);
#pragma unroll
                for (int n2 = 0; n2 < 32; ++n2) src[n2] = X[lane + 64 * n2];
                // leave nothing of the staged samples behind: the padding of the power-spectrum
                // rows is read (with zero weights) by the mel stage and must stay finite
#pragma unroll 1
                for (int i = lane; i < XCH_FLOATS; i += 64) X[i] = 0.0f;
            } else if (PCM16) {
                int16_t h[32];
#pragma unroll
                for (int n2 = 0; n2 < 32; ++n2) h[n2] = pp16[64 * n2];
#pragma unroll
                for (int n2 = 0; n2 < 32; ++n2) src[n2] = PCM16_SCALE * (float)h[n2];
            } else if (HOPR == 0 || !raw_ok) {
#pragma unroll
                for (int n2 = 0; n2 < 32; ++n2) src[n2] = pp[64 * n2];
            }
            if (WIN == IN_REGS) {
#pragma unroll
                for (int n2 = 0; n2 < 32; ++n2) r[n2] = src[n2] * win[n2];
            } else {
                float wv[32];
                if (WIN == IN_LDS) {
                    const float4* wq = reinterpret_cast<const float4*>(__builtin_assume_aligned(win_l, 16)) + lane;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float4 w4 = wq[64 * j];
                        wv[4 * j] = w4.x; wv[4 * j + 1] = w4.y; wv[4 * j + 2] = w4.z; wv[4 * j + 3] = w4.w;
                    }
                } else {
                    // from global memory (the CU's vector cache): 8 loads of 16 bytes
                    const float4* wt = a.window_t + 8 * lane;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float4 w4 = wt[j];
                        wv[4 * j] = w4.x; wv[4 * j + 1] = w4.y; wv[4 * j + 2] = w4.z; wv[4 * j + 3] = w4.w;
                    }
                }
                __builtin_amdgcn_sched_barrier(0);   // all table reads in flight before the first use
#pragma unroll
                for (int n2 = 0; n2 < 32; ++n2) r[n2] = src[n2] * wv[n2];
            }
            if (HOPR > 0) {
                raw_ok = interior && next_same && is_interior(tc, t + 1);
                // (the shift is unconditional: when the next frame reloads everything it is dead)
#pragma unroll
                for (int n2 = 0; n2 < 32 - HOPR; ++n2) raw[n2] = raw[n2 + HOPR];
                if (raw_ok) {
                    const float* pn = pp + a.hop + 64 * (32 - HOPR);
#pragma unroll
                    for (int j = 0; j < HOPR; ++j) raw[32 - HOPR + j] = pn[64 * j];
                }
                __builtin_amdgcn_sched_barrier(0);   // the prefetch is issued before the FFT starts
            }
        }

        // ---- step 1: real DFT-32 over n2, twiddle, exchange ------------------------------------
        {
            float zr[16], zi[16];
#pragma unroll
            for (int m = 0; m < 16; ++m) { zr[m] = r[2 * m]; zi[m] = r[2 * m + 1]; }
            fc::cfft16(zr, zi);
            __builtin_amdgcn_sched_barrier(0);
            float2 tw[16];
            if (TW1 == IN_LDS) {
#pragma unroll
                for (int k2 = 1; k2 < 16; ++k2) tw[k2] = tw1_l[(k2 - 1) * 64 + lane];
            } else {
                const float4* tt = a.tw1_t + 8 * lane;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float4 t4 = tt[j];
                    tw[2 * j + 1] = make_float2(t4.x, t4.y);
                    if (j < 7) tw[2 * j + 2] = make_float2(t4.z, t4.w);
                }
            }
            __builtin_amdgcn_sched_barrier(0);   // twiddle reads overlap the conjugate split
            float ar[16], ai[16];
            fc::rdft32_split<true>(zr, zi, ar, ai);      // ai[9..15] come negated
            X[y0_off] = ar[0];                   // y[n1]      = sum over even n2
            X[y1_off] = ai[0];                   // y[64 + n1] = sum over odd n2
#pragma unroll
            for (int k2 = 1; k2 < 16; ++k2) {
                if (k2 <= 8) fc::cmul(ar[k2], ai[k2], tw[k2].x, tw[k2].y);
                else fc::cmul_nim(ar[k2], ai[k2], tw[k2].x, tw[k2].y);
                X2[k2 * COL_STRIDE + xw[k2 & 3]] = make_float2(ar[k2], ai[k2]);
            }
        }

        // ---- step 2: quad DFT-4 (DPP), twiddle, DFT-16 over m -----------------------------------
        float dr[16], di[16];
        {
            float2 w6[16];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float4 v = *reinterpret_cast<const float4*>(X + (xr_base ^ (4 * j)));
                dr[2 * j] = v.x; di[2 * j] = v.y; dr[2 * j + 1] = v.z; di[2 * j + 1] = v.w;
            }
            if (W64_REGS) {
#pragma unroll
                for (int m = 1; m < 16; ++m) w6[m] = w64r[m];
            } else {
                // 16-byte aligned rows: one ds_read_b128 each (unaligned, the compiler splits them into
                // ds_read2_b64, which run at half the rate)
                const float4* wsrc = reinterpret_cast<const float4*>(
                    __builtin_assume_aligned(reinterpret_cast<const char*>(w64_l) + kb * (W64_STRIDE * 8), 16));
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float4 v = wsrc[j];
                    w6[2 * j + 1] = make_float2(v.x, v.y);
                    if (j < 7) w6[2 * j + 2] = make_float2(v.z, v.w);
                }
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

        // ---- power spectrum -> LDS (natural bin order) -------------------------------------------
        // All exchange reads of this wavefront are complete (data dependence above), so the
        // exchange region can be overwritten.  Column 0 goes to its scratch first so that its
        // round trip through LDS overlaps the regular power computation.
        {
            // Column 0's 64 outputs sit in lanes 0..3, 16 each.  An LDS write costs the same with 4
            // active lanes as with 64, so they are first spread over lanes 0..15 with DPP row
            // shifts -- lane 4 g + q takes outputs ka = 4 i + g of lane q -- and leave in four
            // writes instead of sixteen.
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
            if (lane < 16) {
                float2* S = reinterpret_cast<float2*>(X + SPECIAL_OFF) + kb * SPECIAL_ROW + k2c;
#pragma unroll
                for (int i = 0; i < 4; ++i) S[4 * i] = make_float2(cr[i], ci[i]);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (lane >= 4) {
#pragma unroll
            for (int ka = 0; ka < 16; ++ka) {
                const float re = dr[fc::cfft16_pos(ka)], im = di[fc::cfft16_pos(ka)];
                const float pw = apply_power<PW>(re * re + im * im, a.pw_mode, a.half_power);
                if (ka < 8) X[pb0 + 2 * P_ROW * ka] = pw; else X[pb1 + 2 * P_ROW * (15 - ka)] = pw;
            }
        }
        if (a.mel_out) {
            for (uint32_t m = a.mel_zero_rows; m; m &= m - 1)
                X[MCELL_OFF + 64 * __builtin_ctz(m) + lane] = 0.0f;
        }
        // This lane's job descriptors are requested here so that they return while column 0
        // makes its round trip.
        const bool mel_fast = MELW_LDS && n_jobs <= MEL_REG_JOBS && a.mel_out;
        uint4 jq = make_uint4(0u, 0u, 0u, 0u);
        if (mel_fast) jq = reinterpret_cast<const uint4*>(jobs_l)[lane];
        {
            // column 0: C = DFT-64 of c[r] = y[2r] + i y[2r+1]; Y[j] = E[j] + W_128^j O[j]
            const float2* S = reinterpret_cast<const float2*>(X + SPECIAL_OFF);
            const float2 cj = S[sj];
            const float2 cm = S[sm];
            const float er = cj.x + cm.x, ei = cj.y - cm.y;   // 2 E[j]
            float tr = cj.y + cm.y, ti_ = cm.x - cj.x;        // 2 O[j] = -i (C[j] - conj C[64-j])
            fc::cmul(tr, ti_, tw128.x, tw128.y);
            const float yr = er + tr, yi = ei + ti_;          // 2 Y[j]
            X[16 * lane + 16 * (lane >> 2)] =
                apply_power<PW>(0.25f * (yr * yr + yi * yi), a.pw_mode, a.half_power);
            if (lane == 0) {
                const float zr = er - tr, zi = ei - ti_;      // 2 Y[64] (real)
                X[p_index(1024)] = apply_power<PW>(0.25f * (zr * zr + zi * zi), a.pw_mode, a.half_power);
            }
        }

        if (a.spec_out) {
            float* so = a.spec_out + row * NBIN;
#pragma unroll
            for (int i = 0; i < 16; ++i) so[i * 64 + lane] = X[i * P_ROW + lane];
            if (lane == 0) so[1024] = X[p_index(1024)];
        }

        // ---- sparse mel filterbank: n_jobs jobs per lane, MEL_J quads of four bins per job ------
        // A job multiplies MEL_J consecutive quads of the power spectrum with its weights; the
        // jobs of a chain (consecutive slots of one lane) add up in a register and the last one
        // writes the sum to the chain's cell.  A filter's output is the sum of its cells, and the
        // outputs leave as coalesced rows.  (Filterbanks that would need more cell rows than
        // there are share one cell per filter through LDS atomics, which one wavefront executes
        // in program order, so that sum is deterministic too -- but slow.)
        if (a.mel_out) {
            const int2* jp = (MELW_LDS ? jobs_l : a.mel_jobs) + lane;
            const float4* wp = (MELW_LDS ? melw_l : a.melw) + lane;
            char* Xb = reinterpret_cast<char*>(X);
            auto job_sum = [&](const float4 (&w)[MEL_J], const float4 (&v)[MEL_J]) -> float {
                float acc0 = 0.0f, acc1 = 0.0f;
#pragma unroll
                for (int i = 0; i < MEL_J; ++i) {
                    acc0 = fmaf(w[i].x, v[i].x, acc0);
                    acc1 = fmaf(w[i].y, v[i].y, acc1);
                    acc0 = fmaf(w[i].z, v[i].z, acc0);
                    acc1 = fmaf(w[i].w, v[i].w, acc1);
                }
                return acc0 + acc1;
            };
            auto put = [&](int cell_off, float v) {
                float* c = reinterpret_cast<float*>(Xb + cell_off);
                if (!a.mel_shared) *c = v;
                else __hip_atomic_fetch_add(c, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            };
            float chain = 0.0f;
            if (mel_fast) {
                // all reads of MEL_BATCH jobs in flight at once (n_jobs == MEL_REG_JOBS here)
                const uint32_t jobr[MEL_REG_JOBS] = {jq.x, jq.y, jq.z, jq.w};
#pragma unroll
                for (int j = 0; j < MEL_REG_JOBS; j += MEL_BATCH) {
                    float4 w[MEL_BATCH][MEL_J], v[MEL_BATCH][MEL_J];
                    uint32_t jr[MEL_BATCH];
#pragma unroll
                    for (int u = 0; u < MEL_BATCH; ++u) {
                        jr[u] = jobr[j + u];
                        const float4* pp = reinterpret_cast<const float4*>(Xb + (jr[u] & 0xffffu));
#pragma unroll
                        for (int i = 0; i < MEL_J; ++i) {
                            w[u][i] = wp[((j + u) * MEL_J + i) * 64];
                            v[u][i] = pp[i];
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int u = 0; u < MEL_BATCH; ++u) {
                        const float part = job_sum(w[u], v[u]);
                        chain = (jr[u] & 0x10000u) ? chain + part : part;
                        put((jr[u] >> 16) & ~3u, chain);
                    }
                }
            } else {
                int2 job = jp[0];
                for (int j = 0; j < n_jobs; ++j) {
                    const float4* pp = reinterpret_cast<const float4*>(Xb + job.x);
                    const int cell_off = job.y & ~3;
                    const bool cont = job.y & 1;
                    float4 w[MEL_J], v[MEL_J];
#pragma unroll
                    for (int i = 0; i < MEL_J; ++i) { w[i] = wp[(j * MEL_J + i) * 64]; v[i] = pp[i]; }
                    if (j + 1 < n_jobs) job = jp[(j + 1) * 64];
                    __builtin_amdgcn_sched_barrier(0);   // 2 * MEL_J + 1 reads in flight
                    const float part = job_sum(w, v);
                    chain = cont ? chain + part : part;
                    put(cell_off, chain);
                }
            }
            collect_outputs(row);
        }

    }
    flush_max(runmax, cur_grp, a.gmax, lane);
}
