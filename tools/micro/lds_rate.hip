// LDS instruction cost on gfx950: cycles of the CU's LDS pipe per wave64 instruction, for reads
// and writes of 4 / 8 / 16 bytes per lane, with all 64 lanes or only 16 lanes active, conflict-free
// addresses.  16 wavefronts per CU issue back-to-back so that the pipe, not latency, is measured.
// Cycles are SHADER clocks (s_memtime span of one workgroup), not wall time x a nominal frequency.
//   hipcc -O3 --offload-arch=gfx950 tools/micro/lds_rate.hip -o tools/micro/lds_rate
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

template <int BYTES, bool WRITE, int ACTIVE>
__global__ __launch_bounds__(1024) void k(float* out, int iters, long long* clk) {
    const long long c0 = clock64(), w0 = wall_clock64();
    __shared__ __attribute__((aligned(16))) float lds[16 * 1024];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 16 * 1024; i += 1024) lds[i] = (float)i;
    __syncthreads();
    // BYTES 9 / 17 stand for ds_write2_b32 / ds_write2_b64 (two separate addresses per lane)
    float* base = lds + wave * 1024 + lane * (BYTES == 9 ? 1 : BYTES == 17 ? 2 : BYTES / 4);
    v4f acc = {0, 0, 0, 0};
    v2f acc2 = {0, 0};
    if (lane < ACTIVE) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                float* p = base + ((r & 3) * 256);
                if (WRITE) {
                    if (BYTES == 4) asm volatile("ds_write_b32 %0, %1" ::"v"((uint32_t)(uintptr_t)p), "v"(acc.x) : "memory");
                    if (BYTES == 8) asm volatile("ds_write_b64 %0, %1" ::"v"((uint32_t)(uintptr_t)p), "v"(acc2) : "memory");
                    if (BYTES == 16) asm volatile("ds_write_b128 %0, %1" ::"v"((uint32_t)(uintptr_t)p), "v"(acc) : "memory");
                    if (BYTES == 9) asm volatile("ds_write2_b32 %0, %1, %2 offset0:0 offset1:160" ::"v"((uint32_t)(uintptr_t)p), "v"(acc.x), "v"(acc.y) : "memory");
                    if (BYTES == 17) asm volatile("ds_write2_b64 %0, %1, %2 offset0:0 offset1:80" ::"v"((uint32_t)(uintptr_t)p), "v"(acc2), "v"(acc2) : "memory");
                } else {
                    if (BYTES == 4) { float v; asm volatile("ds_read_b32 %0, %1" : "=v"(v) : "v"((uint32_t)(uintptr_t)p) : "memory"); acc.x += v; }
                    if (BYTES == 8) { v2f v; asm volatile("ds_read_b64 %0, %1" : "=v"(v) : "v"((uint32_t)(uintptr_t)p) : "memory"); acc.y += v.x; }
                    if (BYTES == 16) { v4f v; asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"((uint32_t)(uintptr_t)p) : "memory"); acc.z += v.x; }
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    }
    if (acc.x + acc.y + acc.z == 12345.678f) out[0] = acc.x;
    __syncthreads();
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = clock64() - c0; clk[1] = wall_clock64() - w0; }
}

template <int BYTES, bool WRITE, int ACTIVE>
void run(float* d) {
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int iters = 4000, blocks = prop.multiProcessorCount;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    static long long* clk = nullptr;
    if (!clk && hipMalloc(&clk, 16) != hipSuccess) return;
    hipLaunchKernelGGL((k<BYTES, WRITE, ACTIVE>), dim3(blocks), dim3(1024), 0, 0, d, 10, clk);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<BYTES, WRITE, ACTIVE>), dim3(blocks), dim3(1024), 0, 0, d, iters, clk);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double instr_per_cu = (double)iters * 8 * 16;
    long long h[2] = {0, 0};
    if (hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost) != hipSuccess) return;
    // shader-clock span of one workgroup (s_memtime) / instructions = SHADER cycles of the CU per instruction
    printf("%s %s%-3d %2d lanes: %.3f ms, %.2f SHADER cycles of the CU per instruction (ran at %.0f MHz; %.2f at the nominal %.0f MHz)\n", WRITE ? "write" : "read ", BYTES == 9 || BYTES == 17 ? "2xb" : "b", (BYTES & ~1) * 8, ACTIVE, ms, (double)h[0] / instr_per_cu, h[1] > 0 ? 100.0 * h[0] / h[1] : 0.0, ms * 1e-3 * prop.clockRate * 1e3 / instr_per_cu, prop.clockRate / 1e3);
}

int main() {
    float* d; hipMalloc(&d, 4);
    run<4, false, 64>(d); run<8, false, 64>(d); run<16, false, 64>(d);
    run<4, true, 64>(d); run<8, true, 64>(d); run<16, true, 64>(d);
    run<9, true, 64>(d); run<17, true, 64>(d);
    run<8, true, 16>(d); run<16, true, 16>(d); run<8, false, 16>(d); run<16, false, 16>(d);
    return 0;
}
