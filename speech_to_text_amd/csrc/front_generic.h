// Generic front end for gfx950: any even n_fft that fits LDS, as a dense DFT on the matrix cores.
//
//   pad / frame / window -> DFT as an fp32 MFMA GEMM -> |X|^power -> sparse mel filterbank
//
// Used for every n_fft without an in-register FFT kernel (e.g. 600, 200, 1104), where the DFT is run
// as a genuine dense contraction on v_mfma_f32_16x16x4_f32.  Same role as front_2048.h: the stft +
// abs**power + mel einsum part of librosa.feature.mfcc (/root/reference/src/preprocess.py:134) for
// such n_fft.  The real input is folded first: with p[m] = x[m] + x[N-m] and q[m] = x[m] - x[N-m]
// (0 < m < N/2; p[0] = x[0], p[N/2] = x[N/2], q[0] = q[N/2] = 0),
//     Re X[k] = sum_m p[m] cos(2 pi m k / N),    Im X[k] = -sum_m q[m] sin(2 pi m k / N),
// i.e. two contractions [16 frames x (N/2 + 1)] . [(N/2 + 1) x n_bins], half the multiply-adds of
// the unfolded [16 x N] . [N x 2 n_bins].
//
// One workgroup (4 wavefronts) per 16-frame tile:
//   stage   each wavefront windows and folds 4 frames into LDS, p and q apart, laid out so that
//           the four A operands of four consecutive MFMA steps are one 16-byte read: element m of
//           frame f sits at float ((m & 3) * 16 + f) * lda + (m >> 2), lda = 4 * (odd number of
//           quads) (conflict-free)
//   DFT     wavefront w takes the 16-bin column tiles w, w+4, ...: one accumulator chain for the
//           real parts (p . cos), one for the imaginary parts (q . -sin), so a lane ends up with
//           both parts of its (frame, bin) and squares them in place
//   mel     wavefront w runs the lane-per-filter contraction on frames 4w .. 4w+3 from LDS P
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "front_2048.h"

struct GenArgs {
    const void* pcm;           // float32 samples, or int16 PCM (value / 32768) when pcm16 is set
    const Tile* tiles;
    const WsHeader* hdr;
    const float* window;       // n_fft
    const float* dfrag;        // [col tile][k step / 4][cos, -sin][lane][4] DFT matrices in MFMA B-fragment order
    const float4* melw;
    const int32_t* mel_start;
    const int32_t* mel_filt;
    const MelPass* mel_pass;
    float* mel_out;
    float* spec_out;
    uint32_t* gmax;
    int32_t n_pass;
    int32_t n_mels;
    int32_t n_fft;
    int32_t n_bins;
    int32_t kpad;              // n_fft / 2 + 1 (the folded length) rounded up to a multiple of 16
    int32_t n_coltiles;        // ceil(n_bins / 16)
    int32_t lda;               // floats per (m & 3, frame) row of folded samples
    int32_t ldp;               // floats per power-spectrum row (multiple of 4, + slack for mel over-read)
    int32_t hop;
    int32_t pad;
    int32_t reflect;
    int32_t pw_mode;
    int32_t pcm16;
    float half_power;
};

typedef float gen_f32x4 __attribute__((ext_vector_type(4)));

// NT: 16-frame tiles per workgroup pass (4 NT wavefronts).  Every fragment of the DFT matrix a
// wavefront fetches is used for all NT tiles, so the matrix traffic per frame falls by NT.
template <int PW, int NT>
__global__ __launch_bounds__(256 * NT) void k_melspec_generic(const GenArgs a) {
    using namespace k2048;
    constexpr int THREADS = 256 * NT, WAVES = 4 * NT;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* A = lds;                            // NT x {p, q} x 64 x lda
    float* P = lds + NT * 128 * a.lda;         // NT x 16 x ldp
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int tl = wave >> 2, wq = wave & 3;   // this wavefront stages / finishes frames 4 wq .. 4 wq + 3 of tile tl
    const int n_tiles = a.hdr->n_tiles;
    const int n_super = (n_tiles + NT - 1) / NT;
    const int per_wg = (n_super + gridDim.x - 1) / gridDim.x;
    const int super_end = min(n_super, (int)(blockIdx.x + 1) * per_wg);

    int mel_st[MAX_MEL_PASS], mel_fi[MAX_MEL_PASS], mel_rounds[MAX_MEL_PASS], mel_roff[MAX_MEL_PASS];
#pragma unroll
    for (int p = 0; p < MAX_MEL_PASS; ++p) {
        mel_st[p] = 0; mel_fi[p] = -1; mel_rounds[p] = 0; mel_roff[p] = 0;
        if (p < a.n_pass) {
            mel_st[p] = a.mel_start[p * 64 + lane];
            mel_fi[p] = a.mel_filt[p * 64 + lane];
            mel_rounds[p] = __builtin_amdgcn_readfirstlane(a.mel_pass[p].rounds);
            mel_roff[p] = __builtin_amdgcn_readfirstlane(a.mel_pass[p].round_off);
        }
    }
    // zero the power rows once: the slack past n_bins is read (with zero weights) by the mel stage
    for (int i = threadIdx.x; i < NT * 16 * a.ldp; i += THREADS) P[i] = 0.0f;

    float runmax = 0.0f;
    int cur_grp = -1;
    float* At = A + tl * 128 * a.lda;           // p rows; the q rows follow 64 lda further on
    float* Pt = P + tl * 16 * a.ldp;

    for (int sup = blockIdx.x * per_wg; sup < super_end; ++sup) {
        const int tile = sup * NT + tl;
        const bool have = tile < n_tiles;
        Tile ti = {};
        if (have) ti = a.tiles[tile];
        const int nvalid = have ? min(TILE_FRAMES, ti.F - ti.f0) : 0;
        if (have && ti.grp != cur_grp) {
            flush_max(runmax, cur_grp, a.gmax, lane);
            cur_grp = ti.grp;
            runmax = 0.0f;
        }
        __syncthreads();   // the previous pass's mel stage is done with P, its DFT with A

        // ---- stage: pad + frame + window + fold -------------------------------------------------
        const bool pcm16 = a.pcm16 != 0;
        const int half = a.n_fft >> 1;
        for (int fi = 0; fi < 4; ++fi) {
            const int f = 4 * wq + fi;
            const int start = (ti.f0 + f) * a.hop - a.pad;
            const int lo = (a.reflect && a.pad) ? reflect_slice_start(start, ti.T, a.hop, a.n_fft) : 0;
            auto sample = [&](int n) -> float {          // windowed sample n of the frame (0 <= n < n_fft)
                const int idx = start + n;
                float v = 0.0f;
                if (a.reflect) {
                    if (ti.T > 0) v = load_sample(a.pcm, ti.soff + reflect_index(idx, ti.T, lo), pcm16);
                } else if (idx >= 0 && idx < ti.T) {
                    v = load_sample(a.pcm, ti.soff + idx, pcm16);
                }
                return v * a.window[n];
            };
            for (int m = lane; m < a.kpad; m += 64) {
                float pv = 0.0f, qv = 0.0f;
                if (f < nvalid && m <= half) {
                    const float v = sample(m);
                    if (m > 0 && m < half) {
                        const float u = sample(a.n_fft - m);
                        pv = v + u;
                        qv = v - u;
                    } else {
                        pv = v;
                    }
                }
                const int at = ((m & 3) * 16 + f) * a.lda + (m >> 2);
                At[at] = pv;
                At[64 * a.lda + at] = qv;
            }
        }
        __syncthreads();

        // ---- DFT on the matrix cores, power, -> P ------------------------------------------------
        // lane supplies A[frame = lane & 15][k = 4 s + (lane >> 4)]: four steps per 16-byte read
        const float4* arow = reinterpret_cast<const float4*>(A + ((lane >> 4) * 16 + (lane & 15)) * a.lda);
        const int tile_q = 32 * a.lda;        // float4 units between the tiles' folded samples
        const int q_off = 16 * a.lda;         // ... and from a tile's p rows to its q rows
        const int ks4 = a.kpad >> 4;
        for (int ct = wave; ct < a.n_coltiles; ct += WAVES) {
            const float4* frag = reinterpret_cast<const float4*>(a.dfrag) + (size_t)ct * ks4 * 128 + lane;
            gen_f32x4 acc_re[NT], acc_im[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) { acc_re[t] = (gen_f32x4){0.0f, 0.0f, 0.0f, 0.0f}; acc_im[t] = acc_re[t]; }
            // The matrix fragments come from the L2 / memory-side cache, hundreds of cycles away, and a
            // workgroup has one or two wavefronts per SIMD: they are requested a group of G steps
            // (G x 8 NT products) ahead of their use, in registers.
            constexpr int G = NT <= 2 ? 4 : 1;      // (short transforms, where NT is 3 or 4, have few steps: one ahead is enough)
            float4 bc[G], bs[G];
#pragma unroll
            for (int j = 0; j < G; ++j) {
                const int sj = min(j, ks4 - 1);
                bc[j] = frag[(size_t)sj * 128];
                bs[j] = frag[(size_t)sj * 128 + 64];
            }
            for (int s0 = 0; s0 < ks4; s0 += G) {
                float4 bc_next[G], bs_next[G];
#pragma unroll
                for (int j = 0; j < G; ++j) {
                    const int sj = min(s0 + G + j, ks4 - 1);
                    bc_next[j] = frag[(size_t)sj * 128];
                    bs_next[j] = frag[(size_t)sj * 128 + 64];
                }
#pragma unroll
                for (int j = 0; j < G; ++j) {
                    if (s0 + j < ks4) {
                        float4 ap[NT], aq[NT];
#pragma unroll
                        for (int t = 0; t < NT; ++t) { ap[t] = arow[t * tile_q + s0 + j]; aq[t] = arow[t * tile_q + q_off + s0 + j]; }
#pragma unroll
                        for (int t = 0; t < NT; ++t) {
                            acc_re[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[t].x, bc[j].x, acc_re[t], 0, 0, 0);
                            acc_im[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(aq[t].x, bs[j].x, acc_im[t], 0, 0, 0);
                        }
#pragma unroll
                        for (int t = 0; t < NT; ++t) {
                            acc_re[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[t].y, bc[j].y, acc_re[t], 0, 0, 0);
                            acc_im[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(aq[t].y, bs[j].y, acc_im[t], 0, 0, 0);
                        }
#pragma unroll
                        for (int t = 0; t < NT; ++t) {
                            acc_re[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[t].z, bc[j].z, acc_re[t], 0, 0, 0);
                            acc_im[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(aq[t].z, bs[j].z, acc_im[t], 0, 0, 0);
                        }
#pragma unroll
                        for (int t = 0; t < NT; ++t) {
                            acc_re[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[t].w, bc[j].w, acc_re[t], 0, 0, 0);
                            acc_im[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(aq[t].w, bs[j].w, acc_im[t], 0, 0, 0);
                        }
                    }
                }
#pragma unroll
                for (int j = 0; j < G; ++j) { bc[j] = bc_next[j]; bs[j] = bs_next[j]; }
            }
            // acc[i]: frame 4 (lane >> 4) + i, bin 16 ct + (lane & 15)
            const int bin = 16 * ct + (lane & 15);
            if (bin < a.n_bins) {
#pragma unroll
                for (int t = 0; t < NT; ++t) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float v = acc_re[t][i] * acc_re[t][i] + acc_im[t][i] * acc_im[t][i];
                        P[(t * 16 + 4 * (lane >> 4) + i) * a.ldp + bin] = apply_power<PW>(v, a.pw_mode, a.half_power);
                    }
                }
            }
        }
        __syncthreads();

        // ---- per frame: optional spectrogram tap, sparse mel -----------------------------------
        for (int fi = 0; fi < 4; ++fi) {
            const int f = 4 * wq + fi;
            if (f >= nvalid) break;
            const float* X = Pt + f * a.ldp;
            const int64_t row = ti.row0 + f;
            if (a.spec_out) {
                float* so = a.spec_out + row * a.n_bins;
                for (int k = lane; k < a.n_bins; k += 64) so[k] = X[k];
            }
            if (a.mel_out) {
                float* mo = a.mel_out + row * a.n_mels;
#pragma unroll
                for (int pp_ = 0; pp_ < MAX_MEL_PASS; ++pp_) {
                    if (pp_ < a.n_pass) {
                        const float4* wp = a.melw + (mel_roff[pp_] >> 2) * 64 + lane;
                        const float4* xp = reinterpret_cast<const float4*>(X + mel_st[pp_]);
                        float acc0 = 0.0f, acc1 = 0.0f;
                        const int nq = mel_rounds[pp_] >> 2;
                        for (int i = 0; i < nq; ++i) {
                            const float4 w = wp[i * 64];
                            const float4 v = xp[i];
                            acc0 = fmaf(w.x, v.x, acc0);
                            acc1 = fmaf(w.y, v.y, acc1);
                            acc0 = fmaf(w.z, v.z, acc0);
                            acc1 = fmaf(w.w, v.w, acc1);
                        }
                        const float acc = acc0 + acc1;
                        if (mel_fi[pp_] >= 0) {
                            mo[mel_fi[pp_]] = acc;
                            runmax = fmaxf(runmax, acc);
                        }
                    }
                }
            }
        }
    }
    flush_max(runmax, cur_grp, a.gmax, lane);
}
