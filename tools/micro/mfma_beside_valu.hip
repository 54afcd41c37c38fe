// What an fp32 MFMA costs a wavefront stream that is VALU-issue bound (gfx950).
//
// Each wavefront runs groups of NV independent v_fma_f32 followed by NM matrix instructions
// (v_mfma_f32_16x16x4_f32, or v_mfma_f32_4x4x1_16B_f32), W wavefronts per SIMD.  Printed: SHADER
// cycles (s_memtime) of the SIMD per group, and what each MFMA adds over the same stream without
// MFMAs -- the price of moving work from the VALU to the (otherwise idle) matrix pipe.
//   hipcc -O3 --offload-arch=gfx950 tools/micro/mfma_beside_valu.hip -o tools/micro/mfma_beside_valu
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float v4f __attribute__((ext_vector_type(4)));

template <int KIND, int NV, int NM>
__global__ __launch_bounds__(256) void k(float* out, int iters, float s, long long* clk) {
    const long long c0 = clock64(), w0 = wall_clock64();
    float a[16];
    for (int i = 0; i < 16; ++i) a[i] = threadIdx.x * 0.001f + i;
    v4f acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = (v4f){0.f, 0.f, 0.f, 0.f};
    float ma = a[0], mb = a[1];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
#pragma unroll
            for (int i = 0; i < NV; ++i) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a[i & 15]) : "v"(s));
#pragma unroll
            for (int i = 0; i < NM; ++i) {
                if (KIND == 0) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[(r * NM + i) & 3]) : "v"(ma), "v"(mb));
                if (KIND == 1) asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, %0" : "+v"(acc[(r * NM + i) & 3]) : "v"(ma), "v"(mb));
            }
        }
    }
    float t = 0;
    for (int i = 0; i < 16; ++i) t += a[i];
    for (int i = 0; i < 4; ++i) t += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = clock64() - c0; clk[1] = wall_clock64() - w0; }
    if (t == 12345.678f) out[0] = t;
}

template <int KIND, int NV, int NM>
double run(float* d, int wavesPerSimd, double* mhz) {
    static long long* clk = nullptr;
    if (!clk && hipMalloc(&clk, 16) != hipSuccess) return 0;
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int iters = 4000, blocks = prop.multiProcessorCount * wavesPerSimd;
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return 0;
    hipLaunchKernelGGL((k<KIND, NV, NM>), dim3(blocks), dim3(256), 0, 0, d, 50, 1.0001f, clk);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<KIND, NV, NM>), dim3(blocks), dim3(256), 0, 0, d, iters, 1.0001f, clk);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    long long h[2] = {0, 0};
    if (hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost) != hipSuccess) return 0;
    *mhz = h[1] > 0 ? 100.0 * h[0] / h[1] : 0;
    // kernel time x the real shader clock / groups per SIMD (a wavefront's own span is not the SIMD's
    // busy time: the arbiter serves the oldest wavefronts first)
    return ms * 1e-3 * *mhz * 1e6 / ((double)iters * 8 * wavesPerSimd);
}

template <int KIND, int NV>
void row(const char* name, float* d, int w) {
    double m0, m1, m2, m4;
    const double g0 = run<KIND, NV, 0>(d, w, &m0), g1 = run<KIND, NV, 1>(d, w, &m1), g2 = run<KIND, NV, 2>(d, w, &m2),
                 g4 = run<KIND, NV, 4>(d, w, &m4);
    printf("%-26s %d waves/SIMD, %2d v_fma per group: %6.1f cycles alone (%.0f MHz); +1 MFMA %6.1f (+%.1f); +2 %6.1f (+%.1f each); +4 %6.1f (+%.1f each, %.0f MHz)\n",
           name, w, NV, g0, m0, g1, g1 - g0, g2, (g2 - g0) / 2, g4, (g4 - g0) / 4, m4);
}

int main() {
    float* d; hipMalloc(&d, 4);
    for (int w : {1, 2, 4}) {
        row<0, 8>("v_mfma_f32_16x16x4_f32", d, w);
        row<0, 16>("v_mfma_f32_16x16x4_f32", d, w);
        row<0, 32>("v_mfma_f32_16x16x4_f32", d, w);
        row<1, 8>("v_mfma_f32_4x4x1_16b_f32", d, w);
        row<1, 16>("v_mfma_f32_4x4x1_16b_f32", d, w);
        row<1, 32>("v_mfma_f32_4x4x1_16b_f32", d, w);
    }
    return 0;
}
