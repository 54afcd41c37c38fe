// Issue rate of wave64 VALU instructions on gfx950: v_fma_f32, v_pk_fma_f32, v_pk_add_f32,
// v_pk_mul_f32, v_mov_b32, v_fmac_f32_dpp, v_add_f32, v_cndmask_b32, v_mov_b32_dpp, v_permlane*_swap.  Prints SIMD
// cycles per instruction in SHADER clocks: the kernel's duration is taken from the shader-clock
// counter (s_memtime) of one wavefront, not from wall time and a nominal frequency -- the chip
// lowers its clock under a dense VALU load.
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
    const unsigned long long msk = __ballot((threadIdx.x & 3) == 3);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 32; ++r) {      // 256 instructions per trip: the loop's branch is < 2 % of the stream
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
                if (MODE == 10) asm volatile("v_fmac_f32_e32 %0, %1, %0" : "+v"(a[i]) : "v"(s));
                if (MODE == 11) asm volatile("v_fmamk_f32 %0, %1, 0x3f3504f3, %0" : "+v"(a[i]) : "v"(s));
                if (MODE == 12) asm volatile("v_mul_f32_e32 %0, %1, %0" : "+v"(a[i]) : "v"(s));
                if (MODE == 13) asm volatile("v_fma_f32 %0, %1, %1, -%0" : "+v"(a[i]) : "v"(s));
                if (MODE == 14) asm volatile("v_fma_f32 %0, %0, %1, -%0" : "+v"(a[i]) : "v"(s));
                if (MODE == 15) asm volatile("v_fma_f32 %0, %1, %1, %0" : "+v"(a[i]) : "v"(s));
                if (MODE == 16) asm volatile("v_sub_f32_e32 %0, %1, %0" : "+v"(a[i]) : "v"(s));
                if (MODE == 17) asm volatile("v_fma_f32 %0, %0, %1, -%2" : "+v"(a[i]) : "s"(s), "v"(a[(i + 1) & 7]));
                if (MODE == 18) asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(s));
                if (MODE == 19) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(a[(i + 1) & 7]), "v"(a[(i + 2) & 7]));
                if (MODE == 20) asm volatile("v_mul_f32_e32 %0, 0x3e800000, %0" : "+v"(a[i]));
                if (MODE == 21) asm volatile("v_fmac_f32_e32 %0, 0xbf3504f3, %1" : "+v"(a[i]) : "v"(a[(i + 1) & 7]));
                if (MODE == 22) asm volatile("v_fma_f32 %0, %1, %2, -%0" : "+v"(a[i]) : "v"(a[(i + 1) & 7]), "v"(a[(i + 2) & 7]));
                if (MODE == 23) asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(a[i]) : "v"(a[(i + 1) & 7]), "v"(a[(i + 2) & 7]));
                if (MODE == 24) asm volatile("v_add_f32_e32 %0, %1, %2" : "=v"(a[i]) : "v"(a[(i + 1) & 7]), "v"(a[(i + 2) & 7]));
                if (MODE == 25) asm volatile("v_fma_f32 %0, -%1, %2, %0" : "+v"(a[i]) : "v"(a[(i + 1) & 7]), "v"(a[(i + 2) & 7]));
                if (MODE == 26) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(a[i]), "+v"(a[(i + 1) & 7]));
                if (MODE == 27) asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(a[i]), "+v"(a[(i + 1) & 7]));
                if (MODE == 8) asm volatile("v_cndmask_b32_e64 %0, %0, -%1, %2" : "+v"(a[i]) : "v"(s), "s"(msk));
                if (MODE == 9) asm volatile("v_mov_b32_dpp %0, %1 row_shr:4 row_mask:0xf bank_mask:0x2" : "+v"(a[i]) : "v"(s));
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
    const int iters = 2500;
    const int blocks = prop.multiProcessorCount * wavesPerSimd;   // 256 threads = 4 waves = 1 per SIMD
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 100, 1.0001f, clk);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0001f, clk);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double instr_per_simd = (double)iters * 256 * wavesPerSimd;
    const double clkhz = prop.clockRate * 1e3;   // Hz
    long long h[2] = {0, 0};
    if (hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost) != hipSuccess) return;
    // the shader clock counter against the constant 100 MHz counter: the clock the chip really ran at
    const double mhz = h[1] > 0 ? 100.0 * (double)h[0] / (double)h[1] : 0.0;
    // SIMD cycles per instruction = kernel time (HIP events) x the shader clock the chip really ran at
    // (one wavefront's s_memtime span over its s_memrealtime span) / instructions per SIMD.  One
    // wavefront's own span is NOT the SIMD's busy time: the issue arbiter serves the oldest wavefronts
    // first, so with 4 per SIMD the first two finish in about half the kernel's time.
    printf("%-16s waves/SIMD %d: %.3f ms at %.0f MHz = %.2f SHADER cycles of the SIMD per instruction (%.2f at the nominal %.0f MHz); wavefront 0 issued one every %.2f shader cycles\n",
           name, wavesPerSimd, ms, mhz, ms * 1e-3 * mhz * 1e6 / instr_per_simd, ms * 1e-3 * clkhz / instr_per_simd, clkhz / 1e6,
           (double)h[0] / ((double)iters * 256));
}

int main() {
    float* d; hipMalloc(&d, 4);
    for (int w : {4}) {
        run<0>("v_fma_f32", d, w); run<1>("v_pk_fma_f32", d, w); run<2>("v_pk_add_f32", d, w);
        run<3>("v_pk_mul_f32", d, w); run<4>("v_mov_b32", d, w); run<5>("v_add_f32", d, w);
        run<6>("v_fmac_f32_dpp", d, w); run<7>("v_pk_fma opsel", d, w);
        run<8>("v_cndmask_b32", d, w); run<9>("v_mov_b32_dpp", d, w);
        run<10>("v_fmac_f32_e32", d, w); run<11>("v_fmamk_f32", d, w); run<12>("v_mul_f32_e32", d, w);
        run<13>("v_fma s,s,-a", d, w); run<14>("v_fma a,s,-a", d, w); run<15>("v_fma s,s,a", d, w);
        run<16>("v_sub_f32_e32", d, w); run<17>("v_fma a,sgpr,-b", d, w); run<18>("v_cndmask_e32 vcc", d, w);
        run<19>("v_fma b,c,a", d, w); run<20>("v_mul literal", d, w); run<21>("v_fmac literal", d, w);
        run<22>("v_fma b,c,-a", d, w); run<23>("v_fmac b,c", d, w); run<24>("v_add b,c", d, w); run<25>("v_fma -b,c,a", d, w);
        run<26>("v_permlane32_swap", d, w); run<27>("v_permlane16_swap", d, w);
    }
    return 0;
}
