// Issue rate of wave64 VALU instructions on gfx950: v_fma_f32, v_pk_fma_f32, v_pk_add_f32,
// v_pk_mul_f32, v_mov_b32, v_fmac_f32_dpp, v_add_f32.  Prints SIMD cycles per instruction.
//   hipcc -O3 --offload-arch=gfx950 tools/micro/valu_rate.hip -o /tmp/valu_rate && /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float s, long long* clk) {
    const long long c0 = clock64(), w0 = wall_clock64();
    float a[8]; f2 p[8];
    for (int i = 0; i < 8; ++i) { a[i] = threadIdx.x * 0.001f + i; p[i] = (f2){a[i], a[i] + 1.0f}; }
    f2 s2 = (f2){s, s};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (MODE == 0) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a[i]) : "v"(s));
                if (MODE == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(p[i]) : "v"(s2));
                if (MODE == 2) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(s2));
                if (MODE == 3) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(s2));
                if (MODE == 4) asm volatile("v_mov_b32 %0, %1" : "+v"(a[i]) : "v"(s));
                if (MODE == 5) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(s));
                if (MODE == 6) asm volatile("v_fmac_f32_dpp %0, %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(a[i]) : "v"(s));
                if (MODE == 7) asm volatile("v_pk_fma_f32 %0, %0, %1, %0 op_sel_hi:[1,0,1]" : "+v"(p[i]) : "v"(s2));
            }
        }
    }
    float acc = 0;
    for (int i = 0; i < 8; ++i) acc += a[i] + p[i].x + p[i].y;
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = clock64() - c0; clk[1] = wall_clock64() - w0; }
    if (acc == 12345.678f) out[0] = acc;
}

template <int MODE>
void run(const char* name, float* d, int wavesPerSimd) {
    static long long* clk = nullptr;
    if (!clk && hipMalloc(&clk, 16) != hipSuccess) return;
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int iters = 20000;
    const int blocks = prop.multiProcessorCount * wavesPerSimd;   // 256 threads = 4 waves = 1 per SIMD
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 100, 1.0001f, clk);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0001f, clk);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double instr_per_simd = (double)iters * 32 * wavesPerSimd;
    const double clkhz = prop.clockRate * 1e3;   // Hz
    long long h[2] = {0, 0};
    if (hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost) != hipSuccess) return;
    // the shader clock counter against the constant 100 MHz counter: the clock the chip really ran at
    const double mhz = h[1] > 0 ? 100.0 * (double)h[0] / (double)h[1] : 0.0;
    printf("%-16s waves/SIMD %d: %.3f ms, %.2f cycles/instr at the nominal %.0f MHz, one wavefront issues one every %.2f cycles of its own counter (ran at %.0f MHz)\n",
           name, wavesPerSimd, ms, ms * 1e-3 * clkhz / instr_per_simd, clkhz / 1e6,
           (double)h[0] / ((double)iters * 32), mhz);
}

int main() {
    float* d; hipMalloc(&d, 4);
    for (int w : {1, 4}) {
        run<0>("v_fma_f32", d, w); run<1>("v_pk_fma_f32", d, w); run<2>("v_pk_add_f32", d, w);
        run<3>("v_pk_mul_f32", d, w); run<4>("v_mov_b32", d, w); run<5>("v_add_f32", d, w);
        run<6>("v_fmac_f32_dpp", d, w); run<7>("v_pk_fma opsel", d, w);
    }
    return 0;
}
