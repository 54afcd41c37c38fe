// 2048-point front end, the reference's shape, with everything that is a run-time choice in
// front_2048.h fixed at compile time:
//
//   power 2, mel output only, 65..128 mel filters whose job table has four jobs per lane and the
//   cell rows (1, 2) -- what librosa's Slaney / HTK filterbanks of 65..128 bands give at
//   n_fft = 2048 (/root/reference/src/preprocess.py:134 uses 128) -- all tables in LDS, 16
//   wavefronts per workgroup drawing the frames of one run from one counter.
//
// The arithmetic is that of k_melspec_2048 (same helpers, same order of operations) except that
// the window multiplications are fused into the first butterflies (fc::cfft16_windowed: one
// rounding fewer per sample); the frame loop is straight-line code apart from the edge-frame and
// tile / utterance changes.
#pragma once
#include "front_2048.h"

namespace k2048 {
// LDS bytes: tables (step-1 twiddles, step-2 twiddles, window, mel weights, slot counter), then one
// exchange region per wavefront.
constexpr size_t fast_lds_bytes(int waves, int n_jobs) {
    return sizeof(float) * ((size_t)TW1_FLOATS + W64_FLOATS + WIN_FLOATS + (size_t)n_jobs * MEL_J * 64 * 4 + 16 +
                            (size_t)waves * XCH_FLOATS);
}
}  // namespace k2048

// PCM16: the input is int16 PCM.
// (Measured and dropped: requesting the next frame's samples before the mel stage of the current
// one -- 126 registers, no spills, 1 % slower; storing a frame's outputs at the top of the next
// frame, after its loads, so that the loads are not queued behind the stores -- no change; writing
// column 0's outputs from its four lanes directly, without the DPP spreading -- 2 % slower.)
#ifndef MFCC_FAST_WINFOLD
#define MFCC_FAST_WINFOLD 1
#endif
#ifndef MFCC_FAST_TW1Q
#define MFCC_FAST_TW1Q 1
#endif
#ifndef MFCC_FAST_EDGE_DIRECT
#define MFCC_FAST_EDGE_DIRECT 1
#endif
// experiment: issue priority per stage, two bits each: step 1 (bits 0-1), step 2 (2-3), power store and
// column 0 (4-5), mel stage (6-7)
#ifndef MFCC_FAST_PRIO
#define MFCC_FAST_PRIO 0x54
#endif
#define MFCC_SETPRIO(stage)                                                                            \
    do {                                                                                               \
        if (MFCC_FAST_PRIO) __builtin_amdgcn_s_setprio((MFCC_FAST_PRIO >> (2 * (stage))) & 3);          \
    } while (0)
template <int WAVES, bool PCM16>
__global__ __launch_bounds__(WAVES * 64) void k_melspec_2048_fast(const FrontArgs a) {
    using namespace k2048;
    constexpr int THREADS = WAVES * 64;
    constexpr int NJ = MEL_REG_JOBS;            // jobs per lane
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // tables first: their reads then address off one per-lane register plus an immediate offset
    float2* tw1_l = reinterpret_cast<float2*>(lds);
    float2* w64_l = reinterpret_cast<float2*>(lds + TW1_FLOATS);
    float4* win_l = reinterpret_cast<float4*>(lds + TW1_FLOATS + W64_FLOATS);
    float4* melw_l = reinterpret_cast<float4*>(lds + TW1_FLOATS + W64_FLOATS + WIN_FLOATS);
    int* run_ctr = reinterpret_cast<int*>(melw_l + NJ * MEL_J * 64);
    float* regions = reinterpret_cast<float*>(run_ctr + 16);
    float* X = regions + wave * XCH_FLOATS;
    float2* X2 = reinterpret_cast<float2*>(X);

    for (int i = tid; i < WAVES * XCH_FLOATS; i += THREADS) regions[i] = 0.0f;
#if MFCC_FAST_TW1Q
    // step-1 twiddles as quads: float4 (j, lane) = {tw[2j+1][lane], tw[2j+2][lane]}, j = 0..6, then
    // tw[15][lane] alone: seven 16-byte reads and one 8-byte read per frame (pairs of 8-byte reads
    // get merged into ds_read2st64_b64, which takes twice the LDS cycles)
    for (int i = tid; i < 15 * 64; i += THREADS) {
        const int k2 = 1 + (i >> 6), l = i & 63;
        const int j = (k2 - 1) >> 1;
        tw1_l[j < 7 ? (j * 64 + l) * 2 + ((k2 - 1) & 1) : 7 * 128 + l] = a.tw1[i];
    }
#else
    for (int i = tid; i < 15 * 64; i += THREADS) tw1_l[i] = a.tw1[i];
#endif
    if (tid < 64) {
        // W_64^(m kb), m = 1..15, at slot m - 1 of row kb: eight aligned 16-byte reads per row
        const int m = tid & 15;
        w64_l[(tid >> 4) * W64_STRIDE + (m ? m - 1 : 15)] = m ? a.w64[tid] : make_float2(0.0f, 0.0f);
    }
    // window quads: float4 (j, lane) = w[lane + 64 (4 j + c)], c = 0..3
    for (int i = tid; i < 512; i += THREADS) {
        const int j = i >> 6, l = i & 63;
        win_l[i] = make_float4(a.window[l + 256 * j], a.window[l + 256 * j + 64], a.window[l + 256 * j + 128],
                               a.window[l + 256 * j + 192]);
    }
    for (int i = tid; i < NJ * MEL_J * 64; i += THREADS) melw_l[i] = a.melw[i];
    if (tid == 0) run_ctr[0] = 0;
    __syncthreads();

    const int q = lane & 3, k2c = lane >> 2;
    const int kb = ((q & 1) << 1) | (q >> 1);
    const float2 tw128 = a.tw128[lane];
    const float sgn1 = (q & 2) ? -1.0f : 1.0f;
    const float sgn2 = (q == 1 || q == 2) ? -1.0f : 1.0f;
    const bool lane3 = (q == 3);
    const int xw[4] = {xch_slot(0, lane), xch_slot(1, lane) - COL_STRIDE, xch_slot(2, lane) - 2 * COL_STRIDE,
                       xch_slot(3, lane) - 3 * COL_STRIDE};
    const int y0_off = 2 * xch_slot(0, lane >> 1) + (lane & 1);
    const int y1_off = 2 * xch_slot(0, 32 + (lane >> 1)) + (lane & 1);
    const int xr_base = 2 * (k2c * COL_STRIDE + 16 * q) + 4 * (2 * (k2c & 3) + (q >> 1));
    const int pb0 = p_index(32 * kb + k2c);
    const int pb1 = p_index(128 - 32 * kb - k2c);
    const int sj = (lane & 3) * SPECIAL_ROW + (lane >> 2);
    const int sm = (((64 - lane) & 63) & 3) * SPECIAL_ROW + (((64 - lane) & 63) >> 2);
    const bool has_hi = 64 + lane < a.n_mels;

    float runmax = 0.0f;
    int cur_grp = -1;
    const int n_tiles = a.hdr->n_tiles;
    // one run of frame slots per workgroup, handed out one slot at a time (see front_2048.h)
    const int64_t n_slots = (int64_t)n_tiles * TILE_FRAMES;
    const int64_t per_run = (n_slots + gridDim.x - 1) / gridDim.x;
    const int64_t run_base = (int64_t)blockIdx.x * per_run;
    const int count = run_base < n_slots ? (int)min(per_run, n_slots - run_base) : 0;
    auto request_slot = [&]() -> int {
        int v = 0;
        if (lane == 0) v = atomicAdd(run_ctr, 1);
        return v;
    };
    int tix = -2;
    Tile ti = {}, ti_next = {};
    auto is_interior = [&](const Tile& c, int tt) -> bool {
        const int start = tt * a.hop - a.pad;
        return start >= 0 && start + NFFT <= c.T;
    };

    // samples of the frame about to run (float32, or int16 PCM widened by the load)
    float smp[32];
    auto load_samples = [&](int64_t soff, int tt) {
        const int64_t s0 = soff + (tt * a.hop - a.pad) + lane;
        if (PCM16) {
            const int16_t* pp = static_cast<const int16_t*>(a.pcm) + s0;
#pragma unroll
            for (int n2 = 0; n2 < 32; ++n2) smp[n2] = __builtin_bit_cast(float, (int)pp[64 * n2]);
        } else {
            const float* pp = static_cast<const float*>(a.pcm) + s0;
#pragma unroll
            for (int n2 = 0; n2 < 32; ++n2) smp[n2] = pp[64 * n2];
        }
    };

    // This lane's four mel jobs, decoded once: where each job's 12 power values start, which cell
    // takes the chain's sum so far, and whether the job continues the previous job's chain (as a
    // factor 1 or 0 on the running sum) -- 12 registers instead of a read and its decoding per frame.
    const float4* job_src[NJ];
    float* job_cell[NJ];
    float job_cont[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int2 jb = a.mel_jobs[j * 64 + lane];
        job_src[j] = reinterpret_cast<const float4*>(reinterpret_cast<char*>(X) + jb.x);
        job_cell[j] = reinterpret_cast<float*>(reinterpret_cast<char*>(X) + (jb.y & ~3));
        job_cont[j] = (jb.y & 1) ? 1.0f : 0.0f;
    }

    // outputs of one frame: filter f = cell row 0 (f < 64), rows 1 + 2 (f >= 64); coalesced stores
    const float* cells = X + MCELL_OFF + lane;
    auto collect = [&](int64_t out_row) {
        const float c0 = cells[0], c1 = cells[64], c2 = cells[128];
        const float lo = c0, hi = c1 + c2;
        float* mo = a.mel_out + out_row * a.n_mels;
        mo[lane] = lo;
        runmax = fmaxf(runmax, lo);
        if (has_hi) {
            mo[64 + lane] = hi;
            runmax = fmaxf(runmax, hi);
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
        const Tile tc = ti;
        if (t >= tc.F) continue;                // the last tile of an utterance may be partly empty
        const bool interior = is_interior(tc, t);
        if (tc.grp != cur_grp) {
            flush_max(runmax, cur_grp, a.gmax, lane);
            cur_grp = tc.grp;
            runmax = 0.0f;
        }
        const int64_t row = tc.row0 + (t - tc.f0);
        MFCC_SETPRIO(0);

        // ---- samples, window, step 1: real DFT-32 over n2, twiddle, exchange -------------------------
        // both paths give all 32 sample registers new values, so last frame's are dead here
#if MFCC_FAST_EDGE_DIRECT
        // Zero padding (librosa 0.10's default, the reference's): an edge frame loads like an
        // interior one, from addresses clamped into the utterance, and the lanes that fell outside
        // it take 0 -- all 32 loads in flight at once instead of a rolled loop through LDS.
        const bool direct = !interior && !(a.reflect && a.pad) && tc.T > 0;
#else
        const bool direct = false;
#endif
        if (interior) {
            load_samples(tc.soff, t);
        } else if (direct) {
            const int first = t * a.hop - a.pad + lane;
            const int last = tc.T - 1;
            if (PCM16) {
                const int16_t* pp = static_cast<const int16_t*>(a.pcm) + tc.soff;
#pragma unroll
                for (int n2 = 0; n2 < 32; ++n2)
                    smp[n2] = __builtin_bit_cast(float, (int)pp[min(max(first + 64 * n2, 0), last)]);
            } else {
                const float* pp = static_cast<const float*>(a.pcm) + tc.soff;
#pragma unroll
                for (int n2 = 0; n2 < 32; ++n2) smp[n2] = pp[min(max(first + 64 * n2, 0), last)];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int n2 = 0; n2 < 32; ++n2)
                smp[n2] = ((unsigned)(first + 64 * n2) <= (unsigned)last) ? smp[n2] : 0.0f;
        } else {
            stage_edge_frame(X, a.pcm, tc.soff, PCM16, t * a.hop - a.pad, tc.T, a.reflect && a.pad, lane, a.hop);
#pragma unroll
            for (int n2 = 0; n2 < 32; ++n2) smp[n2] = X[lane + 64 * n2];
#pragma unroll 1
            for (int i = lane; i < XCH_FLOATS; i += 64) X[i] = 0.0f;
        }
        {
            float wv[32];
            const float4* wq = win_l + lane;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float4 w4 = wq[64 * j];
                wv[4 * j] = w4.x; wv[4 * j + 1] = w4.y; wv[4 * j + 2] = w4.z; wv[4 * j + 3] = w4.w;
            }
            __builtin_amdgcn_sched_barrier(0);
            if (PCM16 && (interior || direct)) {
#pragma unroll
                for (int n2 = 0; n2 < 32; ++n2) smp[n2] = PCM16_SCALE * (float)__builtin_bit_cast(int, smp[n2]);
            }
            float zr[16], zi[16];
#if MFCC_FAST_WINFOLD
            // the window multiplications are folded into the first butterflies
            fc::cfft16_windowed(smp, wv, zr, zi);
#else
#pragma unroll
            for (int m = 0; m < 16; ++m) { zr[m] = smp[2 * m] * wv[2 * m]; zi[m] = smp[2 * m + 1] * wv[2 * m + 1]; }
            fc::cfft16(zr, zi);
#endif
            __builtin_amdgcn_sched_barrier(0);
            float2 tw[16];
#if MFCC_FAST_TW1Q
            {
                const float4* tq = reinterpret_cast<const float4*>(tw1_l) + lane;
#pragma unroll
                for (int j = 0; j < 7; ++j) {
                    const float4 v = tq[64 * j];
                    tw[2 * j + 1] = make_float2(v.x, v.y);
                    tw[2 * j + 2] = make_float2(v.z, v.w);
                }
                tw[15] = tw1_l[7 * 128 + lane];
            }
#else
#pragma unroll
            for (int k2 = 1; k2 < 16; ++k2) tw[k2] = tw1_l[(k2 - 1) * 64 + lane];
#endif
            __builtin_amdgcn_sched_barrier(0);
            float ar[16], ai[16];
            fc::rdft32_split<true>(zr, zi, ar, ai);      // ai[9..15] come negated
            X[y0_off] = ar[0];
            X[y1_off] = ai[0];
#pragma unroll
            for (int k2 = 1; k2 < 16; ++k2) {
                if (k2 <= 8) fc::cmul(ar[k2], ai[k2], tw[k2].x, tw[k2].y);
                else fc::cmul_nim(ar[k2], ai[k2], tw[k2].x, tw[k2].y);
                X2[k2 * COL_STRIDE + xw[k2 & 3]] = make_float2(ar[k2], ai[k2]);
            }
        }

        // ---- step 2: quad DFT-4 (DPP), twiddle, DFT-16 over m -----------------------------------
        MFCC_SETPRIO(1);
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
            __builtin_amdgcn_sched_barrier(0);
            quad_fmac8<2>(dr, sgn1); quad_fmac8<2>(dr + 8, sgn1);
            quad_fmac8<2>(di, sgn1); quad_fmac8<2>(di + 8, sgn1);
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float xr = dr[i], xi = di[i];
                dr[i] = lane3 ? -xi : xr;
                di[i] = lane3 ? xr : xi;
            }
            quad_fmac8<1>(dr, sgn2); quad_fmac8<1>(dr + 8, sgn2);
            quad_fmac8<1>(di, sgn2); quad_fmac8<1>(di + 8, sgn2);
#pragma unroll
            for (int m = 1; m < 16; ++m) fc::cmul(dr[m], di[m], w6[m].x, w6[m].y);
            fc::cfft16(dr, di);
        }

        // ---- power spectrum -> LDS (natural bin order) -------------------------------------------
        MFCC_SETPRIO(2);
        {
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
                const float pw = re * re + im * im;
                if (ka < 8) X[pb0 + 2 * P_ROW * ka] = pw; else X[pb1 + 2 * P_ROW * (15 - ka)] = pw;
            }
        }
        X[MCELL_OFF + 64 * 2 + lane] = 0.0f;       // cell row 2: second chains, absent for most filters
        {
            const float2* S = reinterpret_cast<const float2*>(X + SPECIAL_OFF);
            const float2 cj = S[sj];
            const float2 cm = S[sm];
            const float er = cj.x + cm.x, ei = cj.y - cm.y;
            float tr = cj.y + cm.y, ti_ = cm.x - cj.x;
            fc::cmul(tr, ti_, tw128.x, tw128.y);
            const float yr = er + tr, yi = ei + ti_;
            X[16 * lane + 16 * (lane >> 2)] = 0.25f * (yr * yr + yi * yi);
            if (lane == 0) {
                const float zr = er - tr, zi = ei - ti_;
                X[p_index(1024)] = 0.25f * (zr * zr + zi * zi);
            }
        }

        // ---- sparse mel filterbank: four jobs per lane (see front_2048.h) -------------------------
        MFCC_SETPRIO(3);
        {
            const float4* wp = melw_l + lane;
            float chain = 0.0f;
#pragma unroll
            for (int j = 0; j < NJ; j += MEL_BATCH) {
                float4 w[MEL_BATCH][MEL_J], v[MEL_BATCH][MEL_J];
#pragma unroll
                for (int u = 0; u < MEL_BATCH; ++u) {
                    const float4* pp = job_src[j + u];
#pragma unroll
                    for (int i = 0; i < MEL_J; ++i) {
                        w[u][i] = wp[((j + u) * MEL_J + i) * 64];
                        v[u][i] = pp[i];
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < MEL_BATCH; ++u) {
                    float acc0 = 0.0f, acc1 = 0.0f;
#pragma unroll
                    for (int i = 0; i < MEL_J; ++i) {
                        acc0 = fmaf(w[u][i].x, v[u][i].x, acc0);
                        acc1 = fmaf(w[u][i].y, v[u][i].y, acc1);
                        acc0 = fmaf(w[u][i].z, v[u][i].z, acc0);
                        acc1 = fmaf(w[u][i].w, v[u][i].w, acc1);
                    }
                    const float part = acc0 + acc1;
                    chain = fmaf(chain, job_cont[j + u], part);      // chain + part, or part alone
                    *job_cell[j + u] = chain;
                }
            }
            collect(row);
        }
    }
    flush_max(runmax, cur_grp, a.gmax, lane);
}
