// Does the VGPR bank (register index mod 4) of the source operands change the issue rate of a
// three-operand VALU instruction on gfx950?  4 wavefronts per SIMD, independent instructions.
//   hipcc -O3 --offload-arch=gfx950 tools/micro/vgpr_bank.hip -o tools/micro/vgpr_bank
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    asm volatile(
        "v_mov_b32 v1, 1.0\n v_mov_b32 v2, 1.0\n v_mov_b32 v3, 1.0\n v_mov_b32 v4, 1.0\n v_mov_b32 v5, 1.0\n"
        "v_mov_b32 v6, 1.0\n v_mov_b32 v7, 1.0\n v_mov_b32 v8, 1.0\n v_mov_b32 v9, 1.0\n v_mov_b32 v10, 1.0\n"
        "v_mov_b32 v11, 1.0\n v_mov_b32 v12, 1.0\n v_mov_b32 v13, 1.0\n v_mov_b32 v14, 1.0\n v_mov_b32 v15, 1.0\n"
        "v_mov_b32 v16, 1.0\n v_mov_b32 v17, 1.0\n v_mov_b32 v18, 1.0\n v_mov_b32 v19, 1.0\n v_mov_b32 v20, 1.0\n"
        "v_mov_b32 v21, 1.0\n v_mov_b32 v22, 1.0\n v_mov_b32 v23, 1.0\n v_mov_b32 v24, 1.0\n"
        ::: "v1","v2","v3","v4","v5","v6","v7","v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24");
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0)   // three sources in three different banks
            asm volatile(
                "v_fma_f32 v16, v1, v2, v3\n v_fma_f32 v17, v2, v3, v4\n v_fma_f32 v18, v3, v4, v5\n v_fma_f32 v19, v5, v6, v7\n"
                "v_fma_f32 v20, v6, v7, v8\n v_fma_f32 v21, v7, v8, v9\n v_fma_f32 v22, v9, v10, v11\n v_fma_f32 v23, v10, v11, v12\n"
                ::: "v16","v17","v18","v19","v20","v21","v22","v23");
        if (MODE == 1)   // three sources in one bank
            asm volatile(
                "v_fma_f32 v16, v0, v4, v8\n v_fma_f32 v17, v1, v5, v9\n v_fma_f32 v18, v2, v6, v10\n v_fma_f32 v19, v3, v7, v11\n"
                "v_fma_f32 v20, v4, v8, v12\n v_fma_f32 v21, v5, v9, v13\n v_fma_f32 v22, v6, v10, v14\n v_fma_f32 v23, v7, v11, v15\n"
                ::: "v16","v17","v18","v19","v20","v21","v22","v23");
        if (MODE == 2)   // two sources, different banks
            asm volatile(
                "v_add_f32 v16, v1, v2\n v_add_f32 v17, v2, v3\n v_add_f32 v18, v3, v4\n v_add_f32 v19, v5, v6\n"
                "v_add_f32 v20, v6, v7\n v_add_f32 v21, v7, v8\n v_add_f32 v22, v9, v10\n v_add_f32 v23, v10, v11\n"
                ::: "v16","v17","v18","v19","v20","v21","v22","v23");
        if (MODE == 3)   // two sources, same bank
            asm volatile(
                "v_add_f32 v16, v0, v4\n v_add_f32 v17, v1, v5\n v_add_f32 v18, v2, v6\n v_add_f32 v19, v3, v7\n"
                "v_add_f32 v20, v4, v8\n v_add_f32 v21, v5, v9\n v_add_f32 v22, v6, v10\n v_add_f32 v23, v7, v11\n"
                ::: "v16","v17","v18","v19","v20","v21","v22","v23");
        if (MODE == 4)   // dependent chain: each instruction needs the previous result
            asm volatile(
                "v_fma_f32 v16, v16, v2, v3\n v_fma_f32 v16, v16, v3, v4\n v_fma_f32 v16, v16, v2, v5\n v_fma_f32 v16, v16, v6, v7\n"
                "v_fma_f32 v16, v16, v7, v8\n v_fma_f32 v16, v16, v6, v9\n v_fma_f32 v16, v16, v10, v11\n v_fma_f32 v16, v16, v11, v12\n"
                ::: "v16");
    }
    if (iters < 0) out[0] = 1.0f;
}

template <int MODE>
void run(const char* name, float* d) {
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int iters = 100000, blocks = prop.multiProcessorCount * 4;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 100);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-44s %.3f ms, %.2f cycles per instruction per SIMD (at %.0f MHz)\n", name, ms,
           ms * 1e-3 * prop.clockRate * 1e3 / ((double)iters * 8 * 4), prop.clockRate / 1e3);
}

int main() {
    float* d; hipMalloc(&d, 4);
    run<0>("v_fma_f32, sources in 3 banks", d);
    run<1>("v_fma_f32, sources in 1 bank", d);
    run<2>("v_add_f32, sources in 2 banks", d);
    run<3>("v_add_f32, sources in 1 bank", d);
    run<4>("v_fma_f32, dependent chain (per wavefront)", d);
    return 0;
}
