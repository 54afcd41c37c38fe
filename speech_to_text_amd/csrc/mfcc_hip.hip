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

#include <atomic>
#include <type_traits>
#include <utility>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "../../include/mfcc_hip.h"
#include "fft_core.h"
#include "front_2048.h"
#include "front_2048_fast.h"
#include "front_generic.h"
#include "front_pow2.h"
#include "front_400.h"
#include "host_tables.h"

// Workspace of mfcc_execute_noalloc: one copy per GPU, resident from library load until the process
// ends.  Host code never uses this name as an address: mfcc_plan_create asks the runtime for the
// current device's copy (hipGetSymbolAddress) and keeps that pointer in the plan.
static __device__ __align__(256) unsigned char scratch[MFCC_NOALLOC_SCRATCH_BYTES];

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
constexpr int TILE_FRAMES = k2048::TILE_FRAMES;

// One block.  tile_off[u] = number of tiles before utterance u; also zeroes the group maxima.
__global__ __launch_bounds__(1024) void k_prepare_scan(const int64_t* __restrict__ frame_off,
                                                       int n_utts, int32_t* __restrict__ tile_off,
                                                       uint32_t* __restrict__ gmax,
                                                       WsHeader* __restrict__ hdr) {
    __shared__ int32_t wave_sum[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int seg = (n_utts + 1023) / 1024;
    const int lo = min(tid * seg, n_utts), hi = min(lo + seg, n_utts);
    int32_t s = 0;
    for (int u = lo; u < hi; ++u) {
        const int64_t F = frame_off[u + 1] - frame_off[u];
        s += (int32_t)((F + TILE_FRAMES - 1) / TILE_FRAMES);
        gmax[u] = 0u;
    }
    // inclusive scan inside the wavefront, then over the 16 wavefront totals
    int32_t inc = s;
    for (int d = 1; d < 64; d <<= 1) {
        const int32_t v = __shfl_up(inc, d, 64);
        if (lane >= d) inc += v;
    }
    if (lane == 63) wave_sum[wave] = inc;
    __syncthreads();
    int32_t before = 0, total = 0;
    for (int w = 0; w < 16; ++w) {
        const int32_t v = wave_sum[w];
        if (w < wave) before += v;
        total += v;
    }
    if (tid == 0) {
        tile_off[n_utts] = total;
        hdr->n_tiles = total;
        hdr->bad_groups = 0;
    }
    int32_t run = before + inc - s;
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
                                                       int n_utts, Tile* __restrict__ tiles,
                                                       WsHeader* __restrict__ hdr) {
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
    // a group id indexes the running maxima: one outside [0, n_utts) is reported through the
    // header (mfcc_workspace_status) and replaced by the utterance's own slot
    int grp = group ? group[u] : u;
    if ((unsigned)grp >= (unsigned)n_utts) {
        if (tile == tile_off[u]) atomicAdd(&hdr->bad_groups, 1);
        grp = u;
    }
    t.grp = grp;
    tiles[tile] = t;
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

// power_to_db in the log2 domain.  10 log10(max(amin, S)) - ref_db, clamped at the utterance's
// maximum - top_db, equals DB_PER_LOG2 * (log2(max(S, amin, floor)) - ref_l): the clamp becomes a
// floor on the power itself (log2 is monotonic), the factor is applied to the DCT's output and
// ref_l = ref_db / DB_PER_LOG2 is subtracted only when ref is not 1 -- per mel value one
// three-way maximum and one v_log_f32 remain.  amin below the smallest normal number takes the
// library logarithm instead (v_log_f32 does not take subnormal inputs).
constexpr float DB_PER_LOG2 = 3.010299956639812f;   // 10 log10(2)
struct DbScale {
    float floor_pw;   // power below which a value is clamped (0: no clamp)
    float ref_l;      // ref_db / DB_PER_LOG2
};
__device__ __forceinline__ DbScale db_scale(const DbDctArgs& a, float gmax_pw) {
    DbScale d;
    d.ref_l = a.ref_db * (1.0f / DB_PER_LOG2);
    d.floor_pw = 0.0f;
    if (a.top_db >= 0.0f) {
        const float lm = log2f(fmaxf(a.amin, gmax_pw)) - a.top_db * (1.0f / DB_PER_LOG2);
        d.floor_pw = exp2f(lm);
    }
    return d;
}
// MODE 0: amin normal, ref = 1 (the reference's case); 1: amin normal, other ref; 2: library log2
template <int MODE>
__device__ __forceinline__ float db_log2(const DbDctArgs& a, const DbScale& d, float v) {
    const float x = fmaxf(fmaxf(v, a.amin), d.floor_pw);
    if (MODE == 0) return __builtin_amdgcn_logf(x);
    if (MODE == 1) return __builtin_amdgcn_logf(x) - d.ref_l;
    return log2f(x) - d.ref_l;
}
__device__ __forceinline__ int db_mode(const DbDctArgs& a) {
    return a.amin >= 1.0e-37f ? (a.ref_db == 0.0f ? 0 : 1) : 2;
}

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

    const DbScale ds = db_scale(a, a.top_db >= 0.0f ? __uint_as_float(a.gmax[ti.grp]) : 0.0f);
    const int mode = db_mode(a);

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
                for (int i = 0; i < 4; ++i)
                    L[4 * j + i] = mode == 0 ? db_log2<0>(a, ds, v[i]) : mode == 1 ? db_log2<1>(a, ds, v[i]) : db_log2<2>(a, ds, v[i]);
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
                    if (cidx < a.n_mfcc) orow[cidx] = acc[r][i] * (DB_PER_LOG2 * a.lifter[cidx]);
                }
            }
        }
    }
}

// At most 128 mels (a multiple of 4, rows 16-byte aligned) and at most 16 coefficients -- the
// reference's shape is NQ = 8: persistent wavefronts keep the DCT fragments in registers, have a
// whole tile's NQ 16-byte loads in flight at once and request the next tile's before working on
// the current one.  NQ = ceil(n_mels / 16).
template <int NQ>
__global__ __launch_bounds__(256) void k_db_dct_small(const DbDctArgs a) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n_tiles = a.hdr->n_tiles;
    const int n_waves = gridDim.x * 4;
    const int f = lane & 15, kq = lane >> 4;
    float frag[4 * NQ];
#pragma unroll
    for (int s = 0; s < 4 * NQ; ++s) frag[s] = a.dct_frag[s * 64 + lane];
    float lift[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) lift[i] = (4 * kq + i < a.n_mfcc) ? DB_PER_LOG2 * a.lifter[4 * kq + i] : 0.0f;
    const int mode = db_mode(a);
    // mel index of this lane's last quad; past the row's end its DCT weights are zero and the
    // load is redirected to the row's start
    const int last_k = 16 * (NQ - 1) + 4 * kq;
    const int last_off = last_k + 3 < a.n_mels ? last_k : 0;

    auto load_tile = [&](int tile, Tile& ti, float4 (&t)[NQ]) {
        ti = a.tiles[tile];
        const int nvalid = min(TILE_FRAMES, ti.F - ti.f0);
        const float* mrow = a.mel + (ti.row0 + min(f, nvalid - 1)) * a.n_mels;
#pragma unroll
        for (int j = 0; j < NQ - 1; ++j) t[j] = *reinterpret_cast<const float4*>(mrow + 4 * kq + 16 * j);
        t[NQ - 1] = *reinterpret_cast<const float4*>(mrow + last_off);
    };

    int tile = blockIdx.x * 4 + wave;
    if (tile >= n_tiles) return;
    Tile ti, ti_n;
    float4 t[NQ], tn[NQ];
    load_tile(tile, ti, t);
    for (; tile < n_tiles; tile += n_waves) {
        const bool more = tile + n_waves < n_tiles;
        if (more) load_tile(tile + n_waves, ti_n, tn);
        const DbScale ds = db_scale(a, a.top_db >= 0.0f ? __uint_as_float(a.gmax[ti.grp]) : 0.0f);
        f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
        auto contract = [&](auto mode_tag) {
            constexpr int MODE = decltype(mode_tag)::value;
#pragma unroll
            for (int j = 0; j < NQ; ++j) {
                const float v[4] = {t[j].x, t[j].y, t[j].z, t[j].w};
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(frag[4 * j + i], db_log2<MODE>(a, ds, v[i]), acc, 0, 0, 0);
            }
        };
        if (mode == 0) contract(std::integral_constant<int, 0>{});
        else if (mode == 1) contract(std::integral_constant<int, 1>{});
        else contract(std::integral_constant<int, 2>{});
        const int nvalid = min(TILE_FRAMES, ti.F - ti.f0);
        if (f < nvalid) {
            float* orow = a.out + (ti.row0 + f) * a.n_mfcc;
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (4 * kq + i < a.n_mfcc) orow[4 * kq + i] = acc[i] * lift[i];
        }
        if (more) {
            ti = ti_n;
#pragma unroll
            for (int j = 0; j < NQ; ++j) t[j] = tn[j];
        }
    }
}

// DCT (+ lifter) of rows that are already in dB: the S= form of librosa.feature.mfcc.  One
// wavefront per 16 rows, the contraction of k_db_dct without the logarithm.
__global__ __launch_bounds__(256) void k_dct_rows(const float* __restrict__ logmel, int64_t rows,
                                                  const float* __restrict__ dct_frag,
                                                  const float* __restrict__ lifter, float* __restrict__ out,
                                                  int n_mels, int n_mels_pad, int n_mfcc, int n_rt) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t r0 = ((int64_t)blockIdx.x * 4 + wave) * TILE_FRAMES;
    if (r0 >= rows) return;
    const int nvalid = (int)min((int64_t)TILE_FRAMES, rows - r0);
    const int f = lane & 15, kq = lane >> 4;
    const float* mrow = logmel + (r0 + min(f, nvalid - 1)) * n_mels;
    const int steps_total = n_mels_pad / 4;
    for (int rt0 = 0; rt0 < n_rt; rt0 += 4) {
        f32x4 acc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
        for (int c = 0; c < n_mels_pad / 64; ++c) {
            float L[16];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int k = 64 * c + 16 * j + 4 * kq + i;
                    L[4 * j + i] = k < n_mels ? mrow[k] : 0.0f;
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (rt0 + r < n_rt) {
                    const float* fr = dct_frag + ((size_t)(rt0 + r) * steps_total + 16 * c) * 64 + lane;
#pragma unroll
                    for (int s = 0; s < 16; ++s)
                        acc[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(fr[s * 64], L[s], acc[r], 0, 0, 0);
                }
            }
        }
        if (f < nvalid) {
            float* orow = out + (r0 + f) * n_mfcc;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int cidx = (rt0 + r) * 16 + 4 * kq + i;
                    if (cidx < n_mfcc) orow[cidx] = acc[r][i] * lifter[cidx];
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

// ------------------------------------------------------------------------------------------
// power_to_db / amplitude_to_db on an arbitrary array (the spectrogram display path)
// ------------------------------------------------------------------------------------------
// Pass 1: maximum of the (squared, for amplitudes) values, as float bits (the values are >= 0).
__global__ __launch_bounds__(256) void k_to_db_max(const float* __restrict__ x, int64_t n, int amplitude,
                                                   uint32_t* __restrict__ gmax) {
    float m = 0.0f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float v = fabsf(x[i]);
        m = fmaxf(m, amplitude ? v * v : v);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0 && m > 0.0f) atomicMax(gmax, __float_as_uint(m));
}

// Pass 2: 10 log10(max(amin, v)) - 10 log10(max(amin, ref)), clamped at (its maximum - top_db).
// ref_is_max: the reference is the array's maximum (librosa's ref=np.max).
__global__ __launch_bounds__(256) void k_to_db(const float* x, int64_t n, int amplitude, float ref, int ref_is_max,
                                               float amin, float top_db, const uint32_t* __restrict__ gmax,
                                               float* out) {   // out may be x
    const float vmax = __uint_as_float(*gmax);
    const float ref_db = 10.0f * log10f(fmaxf(amin, ref_is_max ? vmax : ref));
    const float thr = top_db >= 0.0f ? 10.0f * log10f(fmaxf(amin, vmax)) - ref_db - top_db : -INFINITY;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float v = fabsf(x[i]);
        if (amplitude) v *= v;
        out[i] = fmaxf(10.0f * log10f(fmaxf(amin, v)) - ref_db, thr);
    }
}

// ------------------------------------------------------------------------------------------
// Delta features (extension): Savitzky-Golay derivative along time, per utterance
// ------------------------------------------------------------------------------------------
constexpr int DELTA_MAX_WIDTH = 65;
struct DeltaArgs {
    const float* x;              // (rows, n_coef)
    const int64_t* frame_off;    // (n_utts + 1)
    float* out;                  // (rows, n_coef)
    int64_t rows;
    int32_t n_utts, n_coef, half;
    float c[DELTA_MAX_WIDTH];    // c[k + half] multiplies x[t + k]
};

// librosa.feature.delta(mode='interp'): the polynomial fitted to the first / last `width` frames
// has a constant derivative of the fitted order, so a frame closer than `half` to an end of its
// utterance takes the value of the nearest frame that has a full window.
__global__ __launch_bounds__(256) void k_delta(const DeltaArgs a) {
    const int64_t total = a.rows * a.n_coef;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = i / a.n_coef;
        const int c = (int)(i - row * a.n_coef);
        int lo = 0, hi = a.n_utts;                 // utterance u: frame_off[u] <= row < frame_off[u + 1]
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (a.frame_off[mid] <= row) lo = mid; else hi = mid;
        }
        const int64_t f0 = a.frame_off[lo], f1 = a.frame_off[lo + 1];
        const int64_t t = min(max(row, f0 + a.half), f1 - 1 - a.half);
        float acc = 0.0f;
        for (int k = -a.half; k <= a.half; ++k) acc = fmaf(a.c[k + a.half], a.x[(t + k) * a.n_coef + c], acc);
        a.out[i] = acc;
    }
}

// Device address of `scratch` on each GPU this process has made a plan on.
constexpr int MAX_DEVICES = 64;
std::mutex g_scratch_mu;
void* g_scratch_ptr[MAX_DEVICES] = {};

// The current device's copy of the static scratch array; `device` must be the current device.
int static_scratch_of(int device, void** out) {
    if (device < 0 || device >= MAX_DEVICES) return fail(MFCC_E_UNSUPPORTED, "device index out of range");
    std::lock_guard<std::mutex> lock(g_scratch_mu);
    if (!g_scratch_ptr[device]) {
        void* ptr = nullptr;
        HIP_TRY(hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch)));
        if (!ptr || reinterpret_cast<uintptr_t>(ptr) % 256)
            return fail(MFCC_E_HIP, "hipGetSymbolAddress returned an unusable address for the static scratch");
        g_scratch_ptr[device] = ptr;
    }
    *out = g_scratch_ptr[device];
    return MFCC_OK;
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
    void (*fn16)(const FrontArgs);   // int16 PCM input; nullptr where the variant has no such build
    int waves;
    int win, tw1;            // k2048::TableHome
    bool w64_regs, melw_lds;
    int hopr;                // 0, or the hop_length / 64 the variant is built for
};

struct mfcc_plan {
    mfcc_host::Resolved r;
    FrontVariant front{};
    int device = 0;
    int n_cu = 0;
    bool fast2048 = false;
    int variant = 0;
    // the streamlined kernel of front_2048_fast.h is usable for this plan
    int fast_mode = 0;
    size_t fast_lds_bytes = 0;
    size_t front_lds_bytes = 0;
    // the variant that takes int16 PCM: `front` itself, or (where that one has no int16 build:
    // the hop-512 variants that keep samples in registers) the best one that has
    FrontVariant front16{};
    size_t front16_lds_bytes = 0;
    mfcc_host::SparseMel sm;       // generic kernel: lane = filter
    mfcc_host::MelJobs mj;         // 2048-point kernel: balanced jobs
    int n_rt = 0, n_mels_pad = 0;
    float ref_db = 0.0f;
    // device tables
    float* d_window = nullptr;
    float4* d_window_t = nullptr;
    float2* d_tw1 = nullptr;
    float4* d_tw1_t = nullptr;
    float2* d_w64 = nullptr;
    float2* d_tw128 = nullptr;
    float4* d_melw = nullptr;
    int32_t* d_mel_start = nullptr;
    int32_t* d_mel_filt = nullptr;
    MelPass* d_mel_pass = nullptr;
    int2* d_mel_jobs = nullptr;
    float* d_dct_frag = nullptr;
    float* d_lifter = nullptr;
    float* d_dfrag = nullptr;
    // in-register FFT front ends other than 2048: 0 (none), 8 / 9 / 10 = log2(n_fft) for the
    // power-of-two kernel (front_pow2.h), or n_fft itself for the 16 A kernels (front_400.h)
    int small_fft = 0, p2_waves = 0, p2_quads = 0, p2_rounds = 0;
    char small_name[16] = {0};                       // "fft<n_fft>" when small_fft names a 16 A length
    size_t p2_lds_bytes = 0;
    float2* d_p2_tw1 = nullptr;
    float4* d_p2_melw = nullptr;
    uint2* d_p2_pairs = nullptr;
    // generic (dense DFT) front end geometry
    int g_kpad = 0, g_lda = 0, g_ldp = 0, g_coltiles = 0, g_nt = 1;
    size_t g_lds_bytes = 0;
    // optional per-kernel timing (mfcc_plan_set_profiling): a ring of event triples (before the
    // front end, after it, after dB+DCT), one triple per execute, so that a caller can run many
    // executes back to back and read all their times afterwards
    // The plan is shared by threads and streams: an execute takes its ring slot with an atomic
    // increment and carries the triple as a local; a slot counts only once all three of its events
    // have been recorded (ev_done), so a failed launch leaves no half-recorded triple to be read.
    bool profiling = false;
    static constexpr int EV_RING = 256;
    hipEvent_t (*ev_ring)[3] = nullptr;
    mutable std::atomic<int64_t> ev_count{0};        // executes that took a slot since profiling was (re)enabled
    mutable std::atomic<int64_t>* ev_done = nullptr; // per slot: 1 + the execute number whose triple is complete, or 0
    // switches for experiments, read from the environment once, when the plan is made
    int share = 16;                                  // MFCC_HIP_SHARE: wavefronts that walk one run of frames together
    int db_per_cu = 2;                               // MFCC_HIP_DB_GRID: dB + DCT workgroups per CU
    // the no-allocation entry point's workspace: this device's copy of the static scratch array
    void* static_scratch = nullptr;
};

namespace {

// Front-end kernel variants: wavefronts per workgroup (one workgroup per CU; LDS-limited) and
// where the per-lane constant tables live.  MFCC_HIP_VARIANT selects one for experiments.
#define FRONT_VARIANT(W, WIN, TW1, W64R, MELW, HOPR, PW) \
    {k_melspec_2048<W, WIN, TW1, W64R, MELW, HOPR, PW>, nullptr, W, WIN, TW1, W64R, MELW, HOPR}
#define FRONT_VARIANT16(W, WIN, TW1, W64R, MELW, PW)                                                   \
    {k_melspec_2048<W, WIN, TW1, W64R, MELW, 0, PW>, k_melspec_2048<W, WIN, TW1, W64R, MELW, 0, PW, true>, \
     W, WIN, TW1, W64R, MELW, 0}
using k2048::IN_GLOBAL;
using k2048::IN_LDS;
using k2048::IN_REGS;
// hopr = 0 variants take any hop_length; hopr = 8 variants need hop_length = 512.
const FrontVariant kFrontVariants[] = {
    FRONT_VARIANT16(12, IN_LDS, IN_LDS, false, false, 0),   // 0: fallback, mel weights streamed
    FRONT_VARIANT16(12, IN_LDS, IN_LDS, false, true, 0),    // 1: any hop
    FRONT_VARIANT(12, IN_LDS, IN_LDS, false, true, 8, 0),   // 2: hop 512, samples reused in registers
    FRONT_VARIANT(12, IN_REGS, IN_LDS, false, true, 8, 0),  // 3
    FRONT_VARIANT(8, IN_REGS, IN_LDS, true, true, 8, 0),    // 4
    FRONT_VARIANT(8, IN_LDS, IN_LDS, false, true, 8, 0),    // 5
    FRONT_VARIANT(12, IN_LDS, IN_LDS, false, false, 8, 0),  // 6: hop 512, mel weights streamed
    FRONT_VARIANT(12, IN_LDS, IN_LDS, true, true, 0, 0),    // 7: step-2 twiddles in registers
    FRONT_VARIANT(12, IN_REGS, IN_LDS, false, true, 0, 0),  // 8: window in registers
    FRONT_VARIANT(12, IN_REGS, IN_LDS, true, true, 0, 0),   // 9: both
    FRONT_VARIANT16(16, IN_LDS, IN_LDS, false, true, 0),    // 10: 16 wavefronts, any hop
    FRONT_VARIANT16(16, IN_LDS, IN_LDS, false, false, 0),   // 11: same, mel weights streamed
    FRONT_VARIANT16(16, IN_GLOBAL, IN_LDS, false, true, 0),   // 12: window from global memory
    FRONT_VARIANT16(16, IN_LDS, IN_GLOBAL, false, true, 0),   // 13: step-1 twiddles from global memory
    FRONT_VARIANT16(16, IN_GLOBAL, IN_GLOBAL, false, true, 0),   // 14: both
};
// power != 2: same structure as variants 0 (mel weights streamed) and 1 (mel weights in LDS).
const FrontVariant kFrontGeneralPower[] = {
    FRONT_VARIANT16(12, IN_LDS, IN_LDS, false, false, 1),
    FRONT_VARIANT16(12, IN_LDS, IN_LDS, false, true, 1),
};
constexpr int kNumFrontVariants = sizeof(kFrontVariants) / sizeof(kFrontVariants[0]);

typedef void (*FrontFn)(const FrontArgs);
FrontFn fast_kernel(bool pcm16) { return pcm16 ? k_melspec_2048_fast<16, true> : k_melspec_2048_fast<16, false>; }

int select_variant(mfcc_plan* p) {
    const int melw_floats = (int)p->mj.w.size() + 2 * 64 * p->mj.n_jobs;   // weights + job table
    auto bytes = [&](int i) {
        const FrontVariant& f = kFrontVariants[i];
        return k2048::lds_bytes(f.waves, f.win, f.tw1, f.w64_regs, f.melw_lds ? melw_floats : 0);
    };
    auto usable = [&](int i) {
        const FrontVariant& f = kFrontVariants[i];
        return bytes(i) <= (size_t)k2048::LDS_LIMIT && (f.hopr == 0 || p->r.c.hop_length == 64 * f.hopr);
    };
    // default: reuse samples in registers when the hop allows it, mel weights in LDS when they fit
    int v = -1;
    for (int cand : {10, 2, 6, 1, 11, 0})
        if (usable(cand)) { v = cand; break; }
    if (const char* e = std::getenv("MFCC_HIP_VARIANT")) {
        const int w = std::atoi(e);
        if (w < 0 || w >= kNumFrontVariants) return fail(MFCC_E_INVALID, "MFCC_HIP_VARIANT out of range");
        if (usable(w)) v = w;
    }
    if (v < 0) return fail(MFCC_E_UNSUPPORTED, "LDS budget exceeded");
    p->front = kFrontVariants[v];
    if (p->r.c.power != 2.0f) p->front = kFrontGeneralPower[p->front.melw_lds ? 1 : 0];
    p->variant = v;
    // The reference's shape takes the streamlined kernel (MFCC_HIP_FAST = 0 keeps the general one, as
    // does MFCC_HIP_VARIANT, which asks for one of its variants).
    p->fast_mode = 0;
    p->fast_lds_bytes = k2048::fast_lds_bytes(16, k2048::MEL_REG_JOBS);
    const bool shape_ok = p->r.c.power == 2.0f && p->mj.n_jobs == k2048::MEL_REG_JOBS && !p->mj.shared &&
                          p->mj.rows[0] == 1 && p->mj.rows[1] == 2 && p->mj.rows[2] == 0 && p->mj.rows[3] == 0 &&
                          (p->mj.zero_rows & ~4u) == 0 && p->r.c.n_mels > 64 && p->r.c.n_mels <= 128 &&
                          p->fast_lds_bytes <= (size_t)k2048::LDS_LIMIT;
    if (shape_ok && !std::getenv("MFCC_HIP_VARIANT")) {
        p->fast_mode = 1;
        if (const char* e = std::getenv("MFCC_HIP_FAST")) p->fast_mode = std::atoi(e) != 0;
    }
    if (p->fast_mode) {
        for (bool pcm16 : {false, true}) {
            hipError_t e2 = hipFuncSetAttribute(reinterpret_cast<const void*>(fast_kernel(pcm16)),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)p->fast_lds_bytes);
            if (e2 != hipSuccess) return fail(MFCC_E_HIP, std::string("hipFuncSetAttribute: ") + hipGetErrorString(e2));
        }
    }
    p->front_lds_bytes = k2048::lds_bytes(p->front.waves, p->front.win, p->front.tw1, p->front.w64_regs,
                                         p->front.melw_lds ? melw_floats : 0);
    p->front16 = p->front;
    if (!p->front16.fn16) {
        for (int cand : {10, 1, 11, 0})
            if (usable(cand) && kFrontVariants[cand].fn16) { p->front16 = kFrontVariants[cand]; break; }
        if (!p->front16.fn16) return fail(MFCC_E_UNSUPPORTED, "LDS budget exceeded");
    }
    p->front16_lds_bytes = k2048::lds_bytes(p->front16.waves, p->front16.win, p->front16.tw1, p->front16.w64_regs,
                                           p->front16.melw_lds ? melw_floats : 0);
    const std::pair<FrontFn, size_t> fns[] = {{p->front.fn, p->front_lds_bytes}, {p->front16.fn16, p->front16_lds_bytes}};
    for (const auto& f : fns) {
        hipError_t e2 = hipFuncSetAttribute(reinterpret_cast<const void*>(f.first),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)f.second);
        if (e2 != hipSuccess) return fail(MFCC_E_HIP, std::string("hipFuncSetAttribute: ") + hipGetErrorString(e2));
    }
    return MFCC_OK;
}

template <int L, int W>
Pow2Fn pow2_pick(bool pw2, bool pcm16) {
    if (pw2) return pcm16 ? k_melspec_pow2<L, W, 0, true> : k_melspec_pow2<L, W, 0, false>;
    return pcm16 ? k_melspec_pow2<L, W, 1, true> : k_melspec_pow2<L, W, 1, false>;
}
// small_fft: 8, 9, 10 for the power-of-two kernels; n_fft itself for the 16 A kernels, which live
// in front_400_inst.hip (lengths and wavefront counts: front_400.h).
Pow2Fn k16a_kernel(int n_fft, int waves, bool pw2, bool pcm16) {
#define MFCC_16A_TRY_PART(K) if (Pow2Fn fn = k16a_lookup_part##K(n_fft, waves, pw2, pcm16)) return fn;
    MFCC_16A_FOR_EACH_PART(MFCC_16A_TRY_PART)
#undef MFCC_16A_TRY_PART
    return nullptr;
}
Pow2Fn pow2_kernel(const mfcc_plan* p, bool pcm16) {
    const bool pw2 = p->r.c.power == 2.0f;
    if (p->small_fft > 10) return k16a_kernel(p->small_fft, p->p2_waves, pw2, pcm16);
    if (p->small_fft == 8) return p->p2_waves == 16 ? pow2_pick<8, 16>(pw2, pcm16) : pow2_pick<8, 8>(pw2, pcm16);
    if (p->small_fft == 9) return p->p2_waves == 16 ? pow2_pick<9, 16>(pw2, pcm16) : pow2_pick<9, 8>(pw2, pcm16);
    return p->p2_waves == 16 ? pow2_pick<10, 16>(pw2, pcm16) : pow2_pick<10, 8>(pw2, pcm16);
}

// Mel tables of the kernels whose mel stage runs lane = (frame, filter) pair (front_pow2.h,
// front_400.h): T frames per pass, frame t's power spectrum at float p_base(t) of the wavefront's
// region with room for row_cap floats.  Returns false when the filterbank does not fit.
template <typename PBase>
bool build_pair_mel(const mfcc_plan* p, const std::vector<float>& basis, int T, PBase p_base, int row_cap,
                    std::vector<float4>& melw, std::vector<uint2>& pairs, int* out_quads, int* out_rounds) {
    const int n_mels = p->r.c.n_mels, n_bins = p->r.n_bins;
    // each filter's span, from a 4-aligned first bin
    std::vector<int> start(n_mels, 0), fq(n_mels, 1);
    for (int f = 0; f < n_mels; ++f) {
        int lo = -1, hi = -1;
        for (int k = 0; k < n_bins; ++k)
            if (basis[(size_t)f * n_bins + k] != 0.0f) { if (lo < 0) lo = k; hi = k; }
        if (lo < 0) continue;
        start[f] = lo & ~3;
        fq[f] = (hi - start[f]) / 4 + 1;
    }
    // round rd: filters rd fpr .. rd fpr + fpr - 1 for all T frames, as many quads as the widest
    const int fpr = 64 / T;
    const int rounds = (n_mels + fpr - 1) / fpr;
    std::vector<int> nq(rounds, 1), qoff(rounds + 1, 0);
    for (int f = 0; f < n_mels; ++f) nq[f / fpr] = std::max(nq[f / fpr], fq[f]);
    for (int rd = 0; rd < rounds; ++rd) qoff[rd + 1] = qoff[rd] + nq[rd];
    for (int f = 0; f < n_mels; ++f)
        if (start[f] + 4 * nq[f / fpr] > row_cap) return false;
    const int quads = qoff[rounds];
    if ((size_t)quads * fpr > 0xffffu) return false;
    melw.assign((size_t)quads * fpr, make_float4(0.0f, 0.0f, 0.0f, 0.0f));
    pairs.assign((size_t)rounds * 64, make_uint2(0u, 0xffffu));
    for (int rd = 0; rd < rounds; ++rd)
        for (int fi = 0; fi < fpr; ++fi) {
            const int f = rd * fpr + fi;
            const bool live = f < n_mels;
            if (live)
                for (int i = 0; i < nq[rd]; ++i) {
                    float w[4];
                    for (int c = 0; c < 4; ++c) {
                        const int k = start[f] + 4 * i + c;
                        w[c] = k < n_bins ? basis[(size_t)f * n_bins + k] : 0.0f;
                    }
                    melw[(size_t)(qoff[rd] + i) * fpr + fi] = make_float4(w[0], w[1], w[2], w[3]);
                }
            for (int t = 0; t < T; ++t) {
                const uint32_t p_off = (uint32_t)(p_base(t) + (live ? start[f] : 0));
                const uint32_t w_idx = (uint32_t)(qoff[rd] * fpr + fi);
                const uint32_t out = live ? (uint32_t)(t * n_mels + f) : 0xffffu;
                pairs[(size_t)rd * 64 + t * fpr + fi] = make_uint2(p_off | (w_idx << 16), out | ((uint32_t)nq[rd] << 16));
            }
        }
    *out_quads = quads;
    *out_rounds = rounds;
    return true;
}

// Tables of the power-of-two front end.  Returns false when the filterbank does not fit its
// layout (the dense-DFT kernel takes over).
template <int L>
bool build_pow2_tables(mfcc_plan* p, const std::vector<float>& basis, std::vector<float2>& tw1,
                       std::vector<float4>& melw, std::vector<uint2>& pairs) {
    using G = kp2::Geo<L>;
    const double PI = 3.14159265358979323846;
    tw1.assign((size_t)std::max(1, G::R - 1) * 64, make_float2(1.0f, 0.0f));
    for (int k2 = 1; k2 < G::R; ++k2)
        for (int n1 = 0; n1 < 64; ++n1) {
            const double ang = -2.0 * PI * (double)(n1 * k2) / (double)G::N;
            const double sc = (k2 == G::R / 2) ? 1.0 : 0.5;   // rdft_cols leaves 2 A[k] except at k = R/2
            tw1[(size_t)(k2 - 1) * 64 + n1] = make_float2((float)(sc * std::cos(ang)), (float)(sc * std::sin(ang)));
        }
    if (!build_pair_mel(p, basis, G::T, [](int t) { return kp2::p_base<L>(t); }, G::row_cap(), melw, pairs,
                        &p->p2_quads, &p->p2_rounds))
        return false;
    p->p2_waves = 0;
    for (int w : {16, 8})
        if (kp2::lds_bytes(G::REGION, w, 64 / G::T, p->p2_quads, p->p2_rounds, G::TW_LDS_FLOATS) <= (size_t)k2048::LDS_LIMIT) {
            p->p2_waves = w;
            break;
        }
    if (!p->p2_waves) return false;
    p->p2_lds_bytes = kp2::lds_bytes(G::REGION, p->p2_waves, 64 / G::T, p->p2_quads, p->p2_rounds, G::TW_LDS_FLOATS);
    return true;
}

// Tables of the n_fft = 16 A front ends (front_400.h); false: the dense-DFT kernel takes over.
template <int A>
bool build_16a_tables(mfcc_plan* p, const std::vector<float>& basis, std::vector<float2>& tw1,
                      std::vector<float4>& melw, std::vector<uint2>& pairs) {
    using G = k16a::Geo<A>;
    const double PI = 3.14159265358979323846;
    tw1.assign((size_t)(G::NC - 1) * 16, make_float2(1.0f, 0.0f));
    for (int c = 1; c < G::NC; ++c)
        for (int b = 0; b < 16; ++b) {
            const double ang = -2.0 * PI * (double)(b * c) / (double)G::N;
            tw1[(size_t)(c - 1) * 16 + b] = make_float2((float)std::cos(ang), (float)std::sin(ang));
        }
    if (!build_pair_mel(p, basis, k16a::T, [](int t) { return G::p_base(t); }, G::P_STRIDE, melw, pairs,
                        &p->p2_quads, &p->p2_rounds))
        return false;
    p->p2_waves = 0;
    const char* force = std::getenv("MFCC_HIP_P2_WAVES");      // experiments: the length's own count or its fallback
    for (int w : {k16a_waves(A), k16a_fallback_waves(A)})
        if (w != 0 && (!force || std::atoi(force) == w) &&
            G::lds_bytes(w, p->p2_quads, p->p2_rounds) <= (size_t)k2048::LDS_LIMIT) {
            p->p2_waves = w;
            break;
        }
    if (!p->p2_waves) return false;
    p->p2_lds_bytes = G::lds_bytes(p->p2_waves, p->p2_quads, p->p2_rounds);
    return true;
}

int setup_pow2(mfcc_plan* p) {
    for (bool pcm16 : {false, true}) {
        hipError_t e2 = hipFuncSetAttribute(reinterpret_cast<const void*>(pow2_kernel(p, pcm16)),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)p->p2_lds_bytes);
        if (e2 != hipSuccess) return fail(MFCC_E_HIP, std::string("hipFuncSetAttribute: ") + hipGetErrorString(e2));
    }
    return MFCC_OK;
}

typedef void (*GenericFn)(const GenArgs);
GenericFn generic_kernel(const mfcc_plan* p) {
    const bool pw2 = p->r.c.power == 2.0f;
    switch (p->g_nt) {
        case 4: return pw2 ? k_melspec_generic<0, 4> : k_melspec_generic<1, 4>;
        case 3: return pw2 ? k_melspec_generic<0, 3> : k_melspec_generic<1, 3>;
        case 2: return pw2 ? k_melspec_generic<0, 2> : k_melspec_generic<1, 2>;
        default: return pw2 ? k_melspec_generic<0, 1> : k_melspec_generic<1, 1>;
    }
}

int setup_generic(mfcc_plan* p) {
    hipError_t e2 = hipFuncSetAttribute(reinterpret_cast<const void*>(generic_kernel(p)),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)p->g_lds_bytes);
    if (e2 != hipSuccess) return fail(MFCC_E_HIP, std::string("hipFuncSetAttribute: ") + hipGetErrorString(e2));
    return MFCC_OK;
}

}  // namespace

// No C++ exception may cross the C ABI: every entry point that can allocate host memory (tables,
// error strings) is a function-try-block that ends in this handler.
#define MFCC_CATCH_ALL                                                       \
    catch (const std::bad_alloc&) {                                          \
        try { g_err = "out of host memory"; } catch (...) {}                 \
        return MFCC_E_NOMEM;                                                 \
    } catch (...) {                                                          \
        try { g_err = "unexpected C++ exception"; } catch (...) {}           \
        return MFCC_E_INVALID;                                               \
    }

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

int mfcc_host_window(const mfcc_config* cfg, double* h_out) try {
    mfcc_host::Resolved r;
    const std::string e = mfcc_host::resolve(cfg, &r);
    if (!e.empty()) return fail(MFCC_E_INVALID, e);
    if (!h_out) return fail(MFCC_E_INVALID, "h_out is NULL");
    std::vector<double> w;
    mfcc_host::build_window(r, w);
    std::memcpy(h_out, w.data(), w.size() * sizeof(double));
    return MFCC_OK;
} MFCC_CATCH_ALL

int mfcc_host_mel_basis(const mfcc_config* cfg, float* h_out) try {
    mfcc_host::Resolved r;
    const std::string e = mfcc_host::resolve(cfg, &r);
    if (!e.empty()) return fail(MFCC_E_INVALID, e);
    if (!h_out) return fail(MFCC_E_INVALID, "h_out is NULL");
    std::vector<float> m;
    mfcc_host::build_mel_basis(r, m);
    std::memcpy(h_out, m.data(), m.size() * sizeof(float));
    return MFCC_OK;
} MFCC_CATCH_ALL

int mfcc_host_dct_basis(const mfcc_config* cfg, float* h_out) try {
    mfcc_host::Resolved r;
    const std::string e = mfcc_host::resolve(cfg, &r);
    if (!e.empty()) return fail(MFCC_E_INVALID, e);
    if (!h_out) return fail(MFCC_E_INVALID, "h_out is NULL");
    std::vector<float> d;
    mfcc_host::build_dct_basis(r, d);
    std::memcpy(h_out, d.data(), d.size() * sizeof(float));
    return MFCC_OK;
} MFCC_CATCH_ALL

int64_t mfcc_num_frames(const mfcc_config* cfg, int64_t n_samples) {
    mfcc_host::Resolved r;
    const std::string e = mfcc_host::resolve(cfg, &r);
    if (!e.empty()) return fail(MFCC_E_INVALID, e);
    const int64_t F = mfcc_host::num_frames(r, n_samples);
    if (F < 0) return fail(MFCC_E_INVALID, "n_fft is too large for an uncentered signal of this length");
    return F;
}

int mfcc_batch_layout(const mfcc_config* cfg, const int64_t* h_sample_offsets, int32_t n_utts,
                      int64_t* h_frame_offsets) try {
    mfcc_host::Resolved r;
    const std::string e = mfcc_host::resolve(cfg, &r);
    if (!e.empty()) return fail(MFCC_E_INVALID, e);
    if (n_utts < 0 || !h_sample_offsets || !h_frame_offsets)
        return fail(MFCC_E_INVALID, "bad batch arguments");
    h_frame_offsets[0] = 0;
    for (int u = 0; u < n_utts; ++u) {
        const int64_t T = h_sample_offsets[u + 1] - h_sample_offsets[u];
        if (T < 0) return fail(MFCC_E_INVALID, "sample offsets must be non-decreasing");
        // the kernels index an utterance's samples with 32-bit integers, up to one transform length
        // past its end
        if (T > 0x7fffffffLL - 4LL * r.c.n_fft)
            return fail(MFCC_E_UNSUPPORTED, "utterances of 2^31 - 4 n_fft samples or more");
        const int64_t F = mfcc_host::num_frames(r, T);
        if (F < 0)
            return fail(MFCC_E_INVALID, "n_fft is too large for uncentered analysis of utterance " +
                                            std::to_string(u));
        h_frame_offsets[u + 1] = h_frame_offsets[u] + F;
    }
    return MFCC_OK;
} MFCC_CATCH_ALL

int mfcc_plan_create(const mfcc_config* cfg, mfcc_plan** out_plan) try {
    if (!out_plan) return fail(MFCC_E_INVALID, "out_plan is NULL");
    *out_plan = nullptr;
    // librosa computes an odd-length transform; this library has no kernel for one -- a limit of
    // the library, not an invalid parameter
    if (cfg && cfg->struct_size == (int32_t)sizeof(mfcc_config) && cfg->n_fft >= 3 && cfg->n_fft % 2)
        return fail(MFCC_E_UNSUPPORTED, "odd n_fft (" + std::to_string(cfg->n_fft) + ") has no kernel in this library");
    mfcc_host::Resolved r;
    const std::string e = mfcc_host::resolve(cfg, &r);
    if (!e.empty()) return fail(MFCC_E_INVALID, e);
    mfcc_plan* p = new (std::nothrow) mfcc_plan();
    if (!p) return fail(MFCC_E_NOMEM, "out of host memory");
    p->r = r;
    p->r.c.window = nullptr;      // the copies below are what the plan keeps
    p->r.c.mel_basis = nullptr;
    p->fast2048 = (r.c.n_fft == k2048::NFFT);

    hipError_t he = hipGetDevice(&p->device);
    hipDeviceProp_t prop;
    if (he == hipSuccess) he = hipGetDeviceProperties(&prop, p->device);
    if (he != hipSuccess) {
        delete p;
        return fail(MFCC_E_HIP, std::string("no usable HIP device: ") + hipGetErrorString(he));
    }
    p->n_cu = prop.multiProcessorCount;
    // experiment switches: read here, once, never on the execute path
    if (const char* e = std::getenv("MFCC_HIP_SHARE")) {
        const int v = std::atoi(e);
        if (v == 1 || v == 2 || v == 4 || v == 8 || v == 16) p->share = v;
    }
    if (const char* e = std::getenv("MFCC_HIP_DB_GRID")) p->db_per_cu = std::max(1, std::atoi(e));

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
    std::vector<int2> jobs;
    if (p->fast2048) {
        mfcc_host::build_mel_jobs(basis, r.c.n_mels, r.n_bins, k2048::MEL_J, k2048::P_ROW_BINS, k2048::P_ROW,
                                  k2048::MCELL_ROWS - 1, k2048::MEL_MAX_ROWS, p->mj);
        if (p->mj.max_read >= k2048::P_END) {
            delete p;
            return fail(MFCC_E_UNSUPPORTED, "mel job reads past the power-spectrum rows");
        }
        jobs.resize(p->mj.at.size());
        for (size_t i = 0; i < jobs.size(); ++i)
            jobs[i] = make_int2(4 * p->mj.at[i], 4 * (k2048::MCELL_OFF + p->mj.cell[i]) | (p->mj.cont[i] ? 1 : 0));
    }
    // n_fft 256 / 512 / 1024 and the 16 A lengths (128 ... 960): the in-register FFT kernels (MFCC_HIP_DENSE_DFT=1 keeps the dense-DFT one)
    std::vector<float2> p2_tw1;
    std::vector<float4> p2_melw;
    std::vector<uint2> p2_pairs;
    if (!std::getenv("MFCC_HIP_DENSE_DFT")) {
#define MFCC_16A_TABLES(A) \
        if (r.c.n_fft == 16 * A && build_16a_tables<A>(p, basis, p2_tw1, p2_melw, p2_pairs)) p->small_fft = 16 * A;
        MFCC_16A_LENGTHS(MFCC_16A_TABLES)
#undef MFCC_16A_TABLES
        if (p->small_fft) std::snprintf(p->small_name, sizeof p->small_name, "fft%d", p->small_fft);
        if (r.c.n_fft == 256 && build_pow2_tables<8>(p, basis, p2_tw1, p2_melw, p2_pairs)) p->small_fft = 8;
        if (r.c.n_fft == 512 && build_pow2_tables<9>(p, basis, p2_tw1, p2_melw, p2_pairs)) p->small_fft = 9;
        if (r.c.n_fft == 1024 && build_pow2_tables<10>(p, basis, p2_tw1, p2_melw, p2_pairs)) p->small_fft = 10;
    }
    std::vector<float> dfrag;
    if (!p->fast2048 && !p->small_fft) {
        // Dense DFT matrices of the folded transform (front_generic.h): cos(2 pi m k / N) for the sums
        // p[m] = x[m] + x[N-m] and -sin(2 pi m k / N) for the differences q[m], m = 0 .. N/2, each
        // [(N/2 + 1) x n_bins] in the B-fragment order of v_mfma_f32_16x16x4_f32: lane l supplies
        // B[k = l >> 4][j = l & 15].
        const int N = r.c.n_fft, K2 = N / 2 + 1;
        p->g_kpad = (K2 + 15) / 16 * 16;
        p->g_lda = 4 * ((p->g_kpad / 16) | 1);      // an odd number of 16-byte quads per row
        p->g_ldp = (std::max(r.n_bins, p->sm.max_read + 1) + 3) / 4 * 4;
        p->g_coltiles = (r.n_bins + 15) / 16;
        const size_t per_tile = sizeof(float) * (128 * (size_t)p->g_lda + 16 * (size_t)p->g_ldp);
        p->g_nt = (int)std::min<size_t>(4, (size_t)k2048::LDS_LIMIT / per_tile);   // tiles per workgroup pass
        if (const char* e = getenv("MFCC_HIP_GENERIC_NT")) p->g_nt = std::min(p->g_nt, std::max(1, atoi(e)));
        p->g_lds_bytes = per_tile * std::max(1, p->g_nt);
        if (p->g_nt < 1) {
            delete p;
            return fail(MFCC_E_UNSUPPORTED, "n_fft too large for the generic (dense DFT) kernel; "
                                            "supported: n_fft = 2048, or even n_fft up to about 1600");
        }
        const int ks4 = p->g_kpad / 16;
        dfrag.assign((size_t)p->g_coltiles * ks4 * 2 * 64 * 4, 0.0f);
        for (int ct = 0; ct < p->g_coltiles; ++ct)
            for (int s4 = 0; s4 < ks4; ++s4)
                for (int l = 0; l < 64; ++l)
                    for (int j = 0; j < 4; ++j) {
                        // four consecutive steps of a lane are adjacent: one 16-byte load
                        const int m = 16 * s4 + 4 * j + (l >> 4), bin = 16 * ct + (l & 15);
                        if (m >= K2 || bin >= r.n_bins) continue;
                        const double ang = 2.0 * PI * (double)(((int64_t)m * bin) % N) / (double)N;
                        const size_t at = ((((size_t)ct * ks4 + s4) * 2) * 64 + l) * 4 + j;
                        dfrag[at] = (float)std::cos(ang);
                        if (m > 0 && m < N / 2) dfrag[at + 64 * 4] = (float)-std::sin(ang);
                    }
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

    int rc = static_scratch_of(p->device, &p->static_scratch);
    if (rc == MFCC_OK) rc = upload(winf, &p->d_window);
    if (rc == MFCC_OK) rc = upload(tw1, &p->d_tw1);
    if (rc == MFCC_OK && p->fast2048) {
        // lane-major copies for the kernel variants that read these tables from global memory
        std::vector<float4> wt(64 * 8), tt(64 * 8);
        for (int l = 0; l < 64; ++l)
            for (int j = 0; j < 8; ++j) {
                wt[l * 8 + j] = make_float4(winf[l + 64 * (4 * j)], winf[l + 64 * (4 * j + 1)], winf[l + 64 * (4 * j + 2)],
                                            winf[l + 64 * (4 * j + 3)]);
                const float2 a = tw1[(2 * j) * 64 + l];                                  // k2 = 2 j + 1
                const float2 b = j < 7 ? tw1[(2 * j + 1) * 64 + l] : make_float2(0.0f, 0.0f);   // k2 = 2 j + 2
                tt[l * 8 + j] = make_float4(a.x, a.y, b.x, b.y);
            }
        rc = upload(wt, &p->d_window_t);
        if (rc == MFCC_OK) rc = upload(tt, &p->d_tw1_t);
    }
    if (rc == MFCC_OK) rc = upload(w64, &p->d_w64);
    if (rc == MFCC_OK) rc = upload(tw128, &p->d_tw128);
    if (rc == MFCC_OK) {
        const std::vector<float>& w = p->fast2048 ? p->mj.w : p->sm.w;
        std::vector<float4> w4(w.size() / 4);
        for (size_t i = 0; i < w4.size(); ++i)
            w4[i] = make_float4(w[4 * i], w[4 * i + 1], w[4 * i + 2], w[4 * i + 3]);
        rc = upload(w4, &p->d_melw);
    }
    if (rc == MFCC_OK) rc = upload(jobs, &p->d_mel_jobs);
    if (rc == MFCC_OK) rc = upload(p->sm.start, &p->d_mel_start);
    if (rc == MFCC_OK) rc = upload(p->sm.filt, &p->d_mel_filt);
    if (rc == MFCC_OK) rc = upload(passes, &p->d_mel_pass);
    if (rc == MFCC_OK) rc = upload(frag, &p->d_dct_frag);
    if (rc == MFCC_OK) rc = upload(lifter, &p->d_lifter);
    if (rc == MFCC_OK) rc = upload(dfrag, &p->d_dfrag);
    if (rc == MFCC_OK && p->small_fft) rc = upload(p2_tw1, &p->d_p2_tw1);
    if (rc == MFCC_OK && p->small_fft) rc = upload(p2_melw, &p->d_p2_melw);
    if (rc == MFCC_OK && p->small_fft) rc = upload(p2_pairs, &p->d_p2_pairs);
    if (rc == MFCC_OK) rc = p->fast2048 ? select_variant(p) : (p->small_fft ? setup_pow2(p) : setup_generic(p));
    if (rc != MFCC_OK) {
        const std::string keep = g_err;
        mfcc_plan_destroy(p);
        g_err = keep;
        return rc;
    }
    *out_plan = p;
    return MFCC_OK;
} MFCC_CATCH_ALL

void mfcc_plan_destroy(mfcc_plan* p) {
    if (!p) return;
    (void)hipFree(p->d_window); (void)hipFree(p->d_tw1); (void)hipFree(p->d_w64);
    (void)hipFree(p->d_window_t); (void)hipFree(p->d_tw1_t);
    (void)hipFree(p->d_tw128); (void)hipFree(p->d_melw); (void)hipFree(p->d_mel_start);
    (void)hipFree(p->d_mel_filt); (void)hipFree(p->d_mel_pass); (void)hipFree(p->d_mel_jobs);
    (void)hipFree(p->d_dct_frag);
    (void)hipFree(p->d_lifter); (void)hipFree(p->d_dfrag);
    (void)hipFree(p->d_p2_tw1); (void)hipFree(p->d_p2_melw); (void)hipFree(p->d_p2_pairs);
    if (p->ev_ring) {
        for (int i = 0; i < mfcc_plan::EV_RING; ++i)
            for (int j = 0; j < 3; ++j) if (p->ev_ring[i][j]) (void)hipEventDestroy(p->ev_ring[i][j]);
        delete[] p->ev_ring;
        delete[] p->ev_done;
    }
    delete p;
}

size_t mfcc_workspace_bytes(const mfcc_plan* plan, int64_t total_frames, int32_t n_utts) {
    if (!plan || total_frames < 0 || n_utts < 0) return 0;
    return ws_layout(total_frames, n_utts, plan->r.c.n_mels, true).total;
}

// One execute's profiling slot: the event triple and the number it will be known by.
struct ProfSlot {
    hipEvent_t* ev = nullptr;
    int64_t seq = 0;
};
static ProfSlot take_prof_slot(const mfcc_plan* plan) {
    ProfSlot s;
    if (!plan->profiling || !plan->ev_ring) return s;
    s.seq = plan->ev_count.fetch_add(1, std::memory_order_relaxed);
    const int slot = (int)(s.seq % mfcc_plan::EV_RING);
    plan->ev_done[slot].store(0, std::memory_order_release);
    s.ev = plan->ev_ring[slot];
    return s;
}
static int finish_prof_slot(const mfcc_plan* plan, const ProfSlot& s, hipStream_t st) {
    if (!s.ev) return MFCC_OK;
    HIP_TRY(hipEventRecord(s.ev[2], st));
    plan->ev_done[s.seq % mfcc_plan::EV_RING].store(s.seq + 1, std::memory_order_release);
    return MFCC_OK;
}

static int launch_front(const mfcc_plan* plan, const void* d_pcm, bool pcm16, const int64_t* d_sample_offsets,
                        const int64_t* d_frame_offsets, const int32_t* d_group, int32_t n_utts,
                        int64_t total_frames, float* mel_out, float* spec_out, char* ws,
                        const WsLayout& L, hipStream_t st, hipEvent_t* ev) {
    WsHeader* hdr = reinterpret_cast<WsHeader*>(ws + L.hdr);
    int32_t* tile_off = reinterpret_cast<int32_t*>(ws + L.tile_off);
    uint32_t* gmax = reinterpret_cast<uint32_t*>(ws + L.gmax);
    Tile* tiles = reinterpret_cast<Tile*>(ws + L.tiles);

    hipLaunchKernelGGL(k_prepare_scan, dim3(1), dim3(1024), 0, st, d_frame_offsets, n_utts, tile_off,
                       gmax, hdr);
    const int pt_blocks = (int)((L.ub_tiles + 255) / 256);
    hipLaunchKernelGGL(k_prepare_tiles, dim3(pt_blocks), dim3(256), 0, st, d_sample_offsets,
                       d_frame_offsets, d_group, tile_off, n_utts, tiles, hdr);

    if (plan->small_fft) {
        Pow2Args pa;
        pa.pcm = d_pcm; pa.tiles = tiles; pa.hdr = hdr;
        pa.window = plan->d_window; pa.tw1 = plan->d_p2_tw1; pa.w64 = plan->d_w64; pa.tw128 = plan->d_tw128;
        pa.melw = plan->d_p2_melw; pa.mel_pairs = plan->d_p2_pairs;
        pa.mel_out = mel_out; pa.spec_out = spec_out; pa.gmax = gmax;
        pa.n_mels = plan->r.c.n_mels; pa.mel_quads = plan->p2_quads; pa.mel_rounds = plan->p2_rounds;
        pa.hop = plan->r.c.hop_length;
        pa.pad = plan->r.c.center ? plan->r.c.n_fft / 2 : 0;
        pa.reflect = plan->r.c.pad_mode == MFCC_PAD_REFLECT;
        pa.pw_mode = plan->r.c.power == 2.0f ? 0 : (plan->r.c.power == 1.0f ? 1 : 2);
        pa.half_power = 0.5f * plan->r.c.power;
        const int passes_per_tile = plan->small_fft > 10 ? TILE_FRAMES / k16a::T : (1 << plan->small_fft) / 128;
        const int64_t wg_needed = (L.ub_tiles * passes_per_tile + plan->p2_waves - 1) / plan->p2_waves;
        const int grid = (int)std::max<int64_t>(1, std::min<int64_t>(wg_needed, plan->n_cu));
        if (ev) HIP_TRY(hipEventRecord(ev[0], st));
        hipLaunchKernelGGL(pow2_kernel(plan, pcm16), dim3(grid), dim3(plan->p2_waves * 64), plan->p2_lds_bytes, st, pa);
        if (ev) HIP_TRY(hipEventRecord(ev[1], st));
        HIP_TRY(hipGetLastError());
        return MFCC_OK;
    }
    if (!plan->fast2048) {
        GenArgs ga;
        ga.pcm = d_pcm; ga.pcm16 = pcm16 ? 1 : 0; ga.tiles = tiles; ga.hdr = hdr;
        ga.window = plan->d_window; ga.dfrag = plan->d_dfrag;
        ga.melw = plan->d_melw; ga.mel_start = plan->d_mel_start; ga.mel_filt = plan->d_mel_filt;
        ga.mel_pass = plan->d_mel_pass;
        ga.mel_out = mel_out; ga.spec_out = spec_out; ga.gmax = gmax;
        ga.n_pass = plan->sm.n_pass; ga.n_mels = plan->r.c.n_mels;
        ga.n_fft = plan->r.c.n_fft; ga.n_bins = plan->r.n_bins;
        ga.kpad = plan->g_kpad; ga.n_coltiles = plan->g_coltiles; ga.lda = plan->g_lda; ga.ldp = plan->g_ldp;
        ga.hop = plan->r.c.hop_length;
        ga.pad = plan->r.c.center ? plan->r.c.n_fft / 2 : 0;
        ga.reflect = plan->r.c.pad_mode == MFCC_PAD_REFLECT;
        ga.pw_mode = plan->r.c.power == 2.0f ? 0 : (plan->r.c.power == 1.0f ? 1 : 2);
        ga.half_power = 0.5f * plan->r.c.power;
        const int nt = plan->g_nt;
        const int per_cu = (int)std::max<size_t>(1, std::min<size_t>(8, (size_t)k2048::LDS_LIMIT / plan->g_lds_bytes));
        const int64_t n_super = (L.ub_tiles + nt - 1) / nt;
        const int grid = (int)std::max<int64_t>(1, std::min<int64_t>(n_super, (int64_t)plan->n_cu * per_cu));
        if (ev) HIP_TRY(hipEventRecord(ev[0], st));
        hipLaunchKernelGGL(generic_kernel(plan), dim3(grid), dim3(256 * nt), plan->g_lds_bytes, st, ga);
        if (ev) HIP_TRY(hipEventRecord(ev[1], st));
        HIP_TRY(hipGetLastError());
        return MFCC_OK;
    }
    FrontArgs fa;
    fa.pcm = d_pcm; fa.pcm16 = pcm16 ? 1 : 0; fa.tiles = tiles; fa.hdr = hdr;
    fa.window = plan->d_window; fa.tw1 = plan->d_tw1; fa.w64 = plan->d_w64; fa.tw128 = plan->d_tw128;
    fa.window_t = plan->d_window_t; fa.tw1_t = plan->d_tw1_t;
    fa.melw = plan->d_melw; fa.mel_jobs = plan->d_mel_jobs;
    fa.mel_out = mel_out; fa.spec_out = spec_out; fa.gmax = gmax;
    fa.n_jobs = plan->mj.n_jobs; fa.n_mels = plan->r.c.n_mels;
    fa.mel_shared = plan->mj.shared ? 1 : 0;
    fa.mel_zero_rows = plan->mj.zero_rows;
    for (int i = 0; i < 4; ++i) fa.mel_rows[i] = plan->mj.rows[i];
    fa.hop = plan->r.c.hop_length;
    fa.pad = plan->r.c.center ? plan->r.c.n_fft / 2 : 0;
    fa.reflect = plan->r.c.pad_mode == MFCC_PAD_REFLECT;
    // all wavefronts of a workgroup draw the frames of one run from a counter in LDS: the CU
    // keeps one window of samples warm in L2 (measured 1.005 x the algorithmic HBM traffic, 1.9 x
    // with a private run per wavefront, MFCC_HIP_SHARE = 1) and its wavefronts finish together
    fa.share = plan->share;
    fa.pw_mode = plan->r.c.power == 2.0f ? 0 : (plan->r.c.power == 1.0f ? 1 : 2);
    fa.half_power = 0.5f * plan->r.c.power;
    const FrontVariant& fv = pcm16 ? plan->front16 : plan->front;
    const int64_t wg_needed = (L.ub_tiles * TILE_FRAMES + fv.waves - 1) / fv.waves;
    const int grid = (int)std::max<int64_t>(1, std::min<int64_t>(wg_needed, plan->n_cu));
    if (ev) HIP_TRY(hipEventRecord(ev[0], st));
    if (plan->fast_mode && mel_out && !spec_out) {
        FrontFn fn = fast_kernel(pcm16);
        hipLaunchKernelGGL(fn, dim3(grid), dim3(16 * 64), plan->fast_lds_bytes, st, fa);
        if (ev) HIP_TRY(hipEventRecord(ev[1], st));
        HIP_TRY(hipGetLastError());
        return MFCC_OK;
    }
    hipLaunchKernelGGL(pcm16 ? fv.fn16 : fv.fn, dim3(grid), dim3(fv.waves * 64),
                       pcm16 ? plan->front16_lds_bytes : plan->front_lds_bytes, st, fa);
    if (ev) HIP_TRY(hipEventRecord(ev[1], st));
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

static int execute_impl(const mfcc_plan* plan, const void* d_pcm, bool pcm16, const int64_t* d_sample_offsets,
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

    const ProfSlot prof = take_prof_slot(plan);
    rc = launch_front(plan, d_pcm, pcm16, d_sample_offsets, d_frame_offsets, d_group, n_utts, total_frames,
                      mel, nullptr, ws, L, st, prof.ev);
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
    if (da.vec_ok && da.n_mels <= 128 && da.n_mfcc <= 16) {
        // two workgroups (eight persistent wavefronts) per CU stream fastest: 0.121 ms per 10 h at
        // 128 mels against 0.131 with 4 to 16 per CU and 0.175 with one
        const int grid = std::max(1, std::min(blocks, plan->n_cu * plan->db_per_cu));
        void (*fn)(const DbDctArgs) = nullptr;
        switch ((da.n_mels + 15) / 16) {
            case 1: fn = k_db_dct_small<1>; break;
            case 2: fn = k_db_dct_small<2>; break;
            case 3: fn = k_db_dct_small<3>; break;
            case 4: fn = k_db_dct_small<4>; break;
            case 5: fn = k_db_dct_small<5>; break;
            case 6: fn = k_db_dct_small<6>; break;
            case 7: fn = k_db_dct_small<7>; break;
            default: fn = k_db_dct_small<8>; break;
        }
        hipLaunchKernelGGL(fn, dim3(grid), dim3(256), 0, st, da);
    } else {
        hipLaunchKernelGGL(k_db_dct, dim3(blocks), dim3(256), 0, st, da);
    }
    HIP_TRY(hipGetLastError());
    return finish_prof_slot(plan, prof, st);
}

int mfcc_execute(const mfcc_plan* plan, const float* d_pcm, const int64_t* d_sample_offsets,
                 const int64_t* d_frame_offsets, const int32_t* d_group, int32_t n_utts,
                 int64_t total_frames, float* d_out, float* d_melspec_out, void* d_workspace,
                 size_t workspace_bytes, void* stream) try {
    return execute_impl(plan, d_pcm, false, d_sample_offsets, d_frame_offsets, d_group, n_utts, total_frames,
                        d_out, d_melspec_out, d_workspace, workspace_bytes, stream);
} MFCC_CATCH_ALL

int mfcc_execute_pcm16(const mfcc_plan* plan, const int16_t* d_pcm16, const int64_t* d_sample_offsets,
                       const int64_t* d_frame_offsets, const int32_t* d_group, int32_t n_utts,
                       int64_t total_frames, float* d_out, float* d_melspec_out, void* d_workspace,
                       size_t workspace_bytes, void* stream) try {
    return execute_impl(plan, d_pcm16, true, d_sample_offsets, d_frame_offsets, d_group, n_utts, total_frames,
                        d_out, d_melspec_out, d_workspace, workspace_bytes, stream);
} MFCC_CATCH_ALL

int mfcc_execute_noalloc(const mfcc_plan* plan, const float* d_pcm, const int64_t* d_sample_offsets,
                         const int64_t* d_frame_offsets, const int32_t* d_group, int32_t n_utts,
                         int64_t total_frames, float* d_out, void* stream) try {
    if (!plan) return fail(MFCC_E_INVALID, "plan is NULL");
    if (n_utts < 0 || total_frames < 0) return fail(MFCC_E_INVALID, "negative batch size");
    int dev = -1;
    HIP_TRY(hipGetDevice(&dev));
    if (dev != plan->device)
        return fail(MFCC_E_INVALID, "the plan was made on device " + std::to_string(plan->device) +
                                        " but the current device is " + std::to_string(dev));
    const size_t need = ws_layout(total_frames, n_utts, plan->r.c.n_mels, true).total;
    if (need > (size_t)MFCC_NOALLOC_SCRATCH_BYTES)
        return fail(MFCC_E_WORKSPACE, "batch needs " + std::to_string(need) + " bytes of workspace, the static scratch of "
                                      "mfcc_execute_noalloc holds " + std::to_string((size_t)MFCC_NOALLOC_SCRATCH_BYTES) +
                                      ": split the batch or call mfcc_execute with a workspace of your own");
    return execute_impl(plan, d_pcm, false, d_sample_offsets, d_frame_offsets, d_group, n_utts, total_frames, d_out,
                        nullptr, plan->static_scratch, (size_t)MFCC_NOALLOC_SCRATCH_BYTES, stream);
} MFCC_CATCH_ALL

int mfcc_plan_set_profiling(mfcc_plan* plan, int enable) try {
    if (!plan) return fail(MFCC_E_INVALID, "plan is NULL");
    // (switching profiling on or off is for a quiet plan: no execute of it may be in flight on
    // another thread; the executes themselves may then run concurrently)
    if (enable && !plan->ev_ring) {
        plan->ev_done = new (std::nothrow) std::atomic<int64_t>[mfcc_plan::EV_RING];
        plan->ev_ring = new (std::nothrow) hipEvent_t[mfcc_plan::EV_RING][3]();
        if (!plan->ev_ring || !plan->ev_done) return fail(MFCC_E_NOMEM, "out of host memory");
        for (int i = 0; i < mfcc_plan::EV_RING; ++i)
            for (int j = 0; j < 3; ++j) HIP_TRY(hipEventCreate(&plan->ev_ring[i][j]));
    }
    plan->profiling = enable != 0;
    plan->ev_count.store(0);
    if (plan->ev_done)
        for (int i = 0; i < mfcc_plan::EV_RING; ++i) plan->ev_done[i].store(0);
    return MFCC_OK;
} MFCC_CATCH_ALL

int mfcc_plan_last_timings(const mfcc_plan* plan, float* ms_front, float* ms_db_dct) try {
    if (!plan || !plan->ev_ring) return fail(MFCC_E_INVALID, "profiling was not enabled on this plan");
    const int64_t count = plan->ev_count.load();
    if (count == 0) return fail(MFCC_E_INVALID, "no execute has been recorded since profiling was enabled");
    const int slot = (int)((count - 1) % mfcc_plan::EV_RING);
    if (plan->ev_done[slot].load(std::memory_order_acquire) != count)
        return fail(MFCC_E_INVALID, "the last execute did not finish recording its events (it failed, or is still being issued)");
    hipEvent_t* ev = plan->ev_ring[slot];
    HIP_TRY(hipEventSynchronize(ev[2]));
    if (ms_front) HIP_TRY(hipEventElapsedTime(ms_front, ev[0], ev[1]));
    if (ms_db_dct) HIP_TRY(hipEventElapsedTime(ms_db_dct, ev[1], ev[2]));
    return MFCC_OK;
} MFCC_CATCH_ALL

int mfcc_plan_timing_history(const mfcc_plan* plan, int32_t max_n, float* ms_front, float* ms_db_dct) try {
    if (!plan || !plan->ev_ring) return fail(MFCC_E_INVALID, "profiling was not enabled on this plan");
    if (max_n < 0) return fail(MFCC_E_INVALID, "max_n is negative");
    const int64_t count = plan->ev_count.load();
    const int64_t have = std::min<int64_t>(count, mfcc_plan::EV_RING);
    const int n = (int)std::min<int64_t>(have, max_n);
    int got = 0;
    for (int i = 0; i < n; ++i) {
        const int64_t seq = count - n + i;
        const int slot = (int)(seq % mfcc_plan::EV_RING);
        if (plan->ev_done[slot].load(std::memory_order_acquire) != seq + 1) continue;   // a launch that failed
        hipEvent_t* ev = plan->ev_ring[slot];
        HIP_TRY(hipEventSynchronize(ev[2]));
        if (ms_front) HIP_TRY(hipEventElapsedTime(&ms_front[got], ev[0], ev[1]));
        if (ms_db_dct) HIP_TRY(hipEventElapsedTime(&ms_db_dct[got], ev[1], ev[2]));
        ++got;
    }
    return got;
} MFCC_CATCH_ALL

int mfcc_workspace_status(const void* d_workspace, void* stream) try {
    if (!d_workspace) return fail(MFCC_E_INVALID, "d_workspace is NULL");
    WsHeader h;
    HIP_TRY(hipMemcpyAsync(&h, d_workspace, sizeof(h), hipMemcpyDeviceToHost, static_cast<hipStream_t>(stream)));
    HIP_TRY(hipStreamSynchronize(static_cast<hipStream_t>(stream)));
    if (h.bad_groups != 0)
        return fail(MFCC_E_INVALID, std::to_string(h.bad_groups) + " utterance(s) had a d_group value outside "
                                    "[0, n_utts); each was given a top_db maximum of its own");
    return MFCC_OK;
} MFCC_CATCH_ALL

const char* mfcc_plan_front_kernel(const mfcc_plan* plan) {
    if (!plan) return nullptr;
    if (plan->fast2048) return "fft2048";
    if (plan->small_fft > 10) return plan->small_name;      // "fft<n_fft>" of a 16 A length
    if (plan->small_fft == 8) return "fft256";
    if (plan->small_fft == 9) return "fft512";
    if (plan->small_fft == 10) return "fft1024";
    return "dense_dft";
}

int mfcc_execute_spectrogram(const mfcc_plan* plan, const float* d_pcm,
                             const int64_t* d_sample_offsets, const int64_t* d_frame_offsets,
                             int32_t n_utts, int64_t total_frames, float* d_spec_out,
                             void* d_workspace, size_t workspace_bytes, void* stream) try {
    int rc = check_batch(plan, d_pcm, d_sample_offsets, d_frame_offsets, n_utts, total_frames);
    if (rc != MFCC_OK) return rc;
    if (n_utts == 0 || total_frames == 0) return MFCC_OK;
    if (!d_spec_out) return fail(MFCC_E_INVALID, "d_spec_out is NULL");
    const WsLayout L = ws_layout(total_frames, n_utts, plan->r.c.n_mels, false);
    if (!d_workspace || workspace_bytes < L.total)
        return fail(MFCC_E_WORKSPACE, "workspace too small: need " + std::to_string(L.total) + " bytes");
    if (reinterpret_cast<uintptr_t>(d_workspace) % 256)
        return fail(MFCC_E_INVALID, "workspace must be 256-byte aligned");
    const ProfSlot prof = take_prof_slot(plan);
    rc = launch_front(plan, d_pcm, false, d_sample_offsets, d_frame_offsets, nullptr, n_utts, total_frames,
                      nullptr, d_spec_out, static_cast<char*>(d_workspace), L, static_cast<hipStream_t>(stream), prof.ev);
    if (rc != MFCC_OK) return rc;
    return finish_prof_slot(plan, prof, static_cast<hipStream_t>(stream));
} MFCC_CATCH_ALL

int mfcc_cmvn_accumulate(const float* d_feats, int64_t n_rows, int32_t n_mfcc, double* d_stats,
                         void* stream) try {
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
} MFCC_CATCH_ALL

int mfcc_cmvn_apply(float* d_feats, int64_t n_rows, int32_t n_mfcc, const double* d_stats, double eps,
                    void* stream) try {
    if (n_rows < 0 || n_mfcc < 1) return fail(MFCC_E_INVALID, "bad CMVN shape");
    if (n_rows == 0) return MFCC_OK;
    if (!d_feats || !d_stats) return fail(MFCC_E_INVALID, "NULL CMVN buffer");
    const int64_t total = n_rows * n_mfcc;
    const int blocks = (int)std::max<int64_t>(1, std::min<int64_t>((total + 255) / 256, 4096));
    hipLaunchKernelGGL(k_cmvn_apply, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), d_feats,
                       n_rows, n_mfcc, d_stats, eps);
    HIP_TRY(hipGetLastError());
    return MFCC_OK;
} MFCC_CATCH_ALL

int mfcc_to_db(const float* d_in, int64_t n, int32_t amplitude, float ref, int32_t ref_is_max, float amin,
               float top_db, float* d_out, void* d_workspace, void* stream) try {
    if (n < 0) return fail(MFCC_E_INVALID, "negative size");
    if (!(amin > 0.0f)) return fail(MFCC_E_INVALID, "amin must be strictly positive");
    if (n == 0) return MFCC_OK;
    if (!d_in || !d_out || !d_workspace) return fail(MFCC_E_INVALID, "NULL buffer");
    hipStream_t st = static_cast<hipStream_t>(stream);
    uint32_t* gmax = static_cast<uint32_t*>(d_workspace);
    HIP_TRY(hipMemsetAsync(gmax, 0, sizeof(uint32_t), st));
    const int blocks = (int)std::max<int64_t>(1, std::min<int64_t>((n + 255) / 256, 8192));
    // amplitudes: ref and amin enter squared (amplitude_to_db = power_to_db(S^2, ref^2, amin^2))
    const float r = amplitude ? ref * ref : fabsf(ref), am = amplitude ? amin * amin : amin;
    if (top_db >= 0.0f || ref_is_max)
        hipLaunchKernelGGL(k_to_db_max, dim3(blocks), dim3(256), 0, st, d_in, n, amplitude, gmax);
    hipLaunchKernelGGL(k_to_db, dim3(blocks), dim3(256), 0, st, d_in, n, amplitude, r, ref_is_max, am, top_db, gmax, d_out);
    HIP_TRY(hipGetLastError());
    return MFCC_OK;
} MFCC_CATCH_ALL

int mfcc_dct(const mfcc_plan* plan, const float* d_logmel, int64_t rows, float* d_out, void* stream) try {
    if (!plan) return fail(MFCC_E_INVALID, "plan is NULL");
    if (rows < 0) return fail(MFCC_E_INVALID, "rows must be non-negative");
    if (rows == 0) return MFCC_OK;
    if (!d_logmel || !d_out) return fail(MFCC_E_INVALID, "NULL buffer");
    if (rows > 0x7fffffffLL * 16) return fail(MFCC_E_UNSUPPORTED, "batch too large: split it");
    int dev = -1;
    HIP_TRY(hipGetDevice(&dev));      // like every entry point: the caller's current device is left alone
    if (dev != plan->device)
        return fail(MFCC_E_INVALID, "the plan was made on device " + std::to_string(plan->device) +
                                        " but the current device is " + std::to_string(dev));
    const int64_t tiles = (rows + TILE_FRAMES - 1) / TILE_FRAMES;
    hipLaunchKernelGGL(k_dct_rows, dim3((unsigned)((tiles + 3) / 4)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       d_logmel, rows, plan->d_dct_frag, plan->d_lifter, d_out, plan->r.c.n_mels, plan->n_mels_pad,
                       plan->r.c.n_mfcc, plan->n_rt);
    HIP_TRY(hipGetLastError());
    return MFCC_OK;
} MFCC_CATCH_ALL

int mfcc_delta(const float* d_feats, const int64_t* d_frame_offsets, const int64_t* h_frame_offsets, int32_t n_utts,
               int32_t n_coef, int32_t width, int32_t order, float* d_out, void* stream) try {
    if (n_utts < 0 || n_coef < 1) return fail(MFCC_E_INVALID, "bad delta shape");
    if (width < 3 || width % 2 == 0 || width > DELTA_MAX_WIDTH)
        return fail(MFCC_E_INVALID, "width must be an odd integer >= 3 (and at most 65)");
    if (order < 1 || order >= width) return fail(MFCC_E_INVALID, "order must be a positive integer below width");
    if (n_utts == 0) return MFCC_OK;
    if (!d_feats || !d_frame_offsets || !h_frame_offsets || !d_out) return fail(MFCC_E_INVALID, "NULL delta buffer");
    for (int u = 0; u < n_utts; ++u) {
        const int64_t F = h_frame_offsets[u + 1] - h_frame_offsets[u];
        if (F > 0 && F < width)
            return fail(MFCC_E_INVALID, "when mode='interp', width=" + std::to_string(width) +
                                            " cannot exceed the " + std::to_string(F) + " frames of utterance " +
                                            std::to_string(u));
    }
    // Savitzky-Golay coefficients of the order-th derivative of a degree-`order` least-squares fit:
    // with A[k][j] = k^j (k = -half .. half), c = order! * (row `order` of (A^T A)^-1 A^T)
    DeltaArgs a;
    a.half = width / 2;
    const int m = order + 1;
    std::vector<double> N((size_t)m * m, 0.0), inv((size_t)m * m, 0.0);
    for (int k = -a.half; k <= a.half; ++k)
        for (int i = 0; i < m; ++i)
            for (int j = 0; j < m; ++j) N[(size_t)i * m + j] += std::pow((double)k, i + j);
    for (int i = 0; i < m; ++i) inv[(size_t)i * m + i] = 1.0;
    for (int col = 0; col < m; ++col) {                     // Gauss-Jordan with partial pivoting
        int piv = col;
        for (int r2 = col + 1; r2 < m; ++r2)
            if (std::fabs(N[(size_t)r2 * m + col]) > std::fabs(N[(size_t)piv * m + col])) piv = r2;
        for (int j = 0; j < m; ++j) {
            std::swap(N[(size_t)col * m + j], N[(size_t)piv * m + j]);
            std::swap(inv[(size_t)col * m + j], inv[(size_t)piv * m + j]);
        }
        const double d = N[(size_t)col * m + col];
        if (std::fabs(d) < 1e-300) return fail(MFCC_E_INVALID, "order too large for this width");
        for (int j = 0; j < m; ++j) { N[(size_t)col * m + j] /= d; inv[(size_t)col * m + j] /= d; }
        for (int r2 = 0; r2 < m; ++r2) {
            if (r2 == col) continue;
            const double f = N[(size_t)r2 * m + col];
            for (int j = 0; j < m; ++j) {
                N[(size_t)r2 * m + j] -= f * N[(size_t)col * m + j];
                inv[(size_t)r2 * m + j] -= f * inv[(size_t)col * m + j];
            }
        }
    }
    double fact = 1.0;
    for (int i = 2; i <= order; ++i) fact *= i;
    for (int k = -a.half; k <= a.half; ++k) {
        double v = 0.0;
        for (int j = 0; j < m; ++j) v += inv[(size_t)order * m + j] * std::pow((double)k, j);
        a.c[k + a.half] = (float)(fact * v);
    }
    a.x = d_feats; a.frame_off = d_frame_offsets; a.out = d_out;
    a.rows = h_frame_offsets[n_utts]; a.n_utts = n_utts; a.n_coef = n_coef;
    if (a.rows == 0) return MFCC_OK;
    const int64_t total = a.rows * n_coef;
    const int blocks = (int)std::max<int64_t>(1, std::min<int64_t>((total + 255) / 256, 8192));
    hipLaunchKernelGGL(k_delta, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), a);
    HIP_TRY(hipGetLastError());
    return MFCC_OK;
} MFCC_CATCH_ALL

int mfcc_host_pcm16_from_float(const float* h_src, int64_t n, int16_t* h_dst) {
    if (n < 0 || (n > 0 && (!h_src || !h_dst))) return fail(MFCC_E_INVALID, "bad host buffers");
    // Branch-free so that the host compiler vectorises it.  Adding and subtracting 1.5 * 2^23
    // rounds to the nearest integer for |v| < 2^22; anything larger (or NaN) fails the range test.
    int ok = 1;
    for (int64_t i = 0; i < n; ++i) {
        const float v = h_src[i] * 32768.0f;             // exact: a power of two
        const float c = fminf(fmaxf(v, -32768.0f), 32767.0f);
        const float r = (c + 12582912.0f) - 12582912.0f;
        ok &= (r == v);
        h_dst[i] = (int16_t)(int32_t)r;
    }
    return ok;
}

}  // extern "C"
