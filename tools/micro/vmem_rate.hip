// Vector-memory load cost on gfx950 for cache-resident data: cycles of the CU per wave64 load
// instruction, 16 wavefronts per CU issuing back to back, for the shapes the front-end kernels use:
//   dword  x 256 contiguous bytes per instruction            (front_2048*.h, front_pow2.h)
//   dword  x four 64-byte pieces one hop apart               (front_400.h)
//   dwordx2 / dwordx4 x 512 / 1024 contiguous bytes
// Footprint per wavefront FOOT bytes, re-read many times (L1 / L2 resident by size).
//   hipcc -O3 --offload-arch=gfx950 tools/micro/vmem_rate.hip -o tools/micro/vmem_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { if ((x) != hipSuccess) { printf("HIP error at %s:%d\n", __FILE__, __LINE__); return; } } while (0)

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

// SHAPE 0: dword contiguous, 1: dword in 4 pieces 640 B apart, 2: dwordx2, 4: dwordx4
template <int SHAPE>
__global__ __launch_bounds__(1024) void k(const float* __restrict__ src, float* out, int iters, int foot_floats) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // every wavefront of a workgroup walks its own window of the workgroup's slice, as consecutive
    // frames do; windows of neighbouring wavefronts overlap by 3/4
    const float* base = src + (size_t)blockIdx.x * (foot_floats * 4) + wave * (foot_floats / 4);
    const int lane_off = SHAPE == 0 ? lane : SHAPE == 1 ? (lane >> 4) * 160 + (lane & 15) : SHAPE == 2 ? 2 * lane : 4 * lane;
    const int step = SHAPE == 0 ? 64 : SHAPE == 1 ? 16 : SHAPE == 2 ? 128 : 256;
    const int per_iter = 8 * step;
    float acc = 0.0f;
    int off = 0;
    for (int it = 0; it < iters; ++it) {
        const float* p = base + off + lane_off;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            if (SHAPE <= 1) { float v = p[r * step]; acc += v; }
            if (SHAPE == 2) { v2f v = *reinterpret_cast<const v2f*>(p + r * step); acc += v.x + v.y; }
            if (SHAPE == 4) { v4f v = *reinterpret_cast<const v4f*>(p + r * step); acc += v.x + v.w; }
        }
        off += per_iter;
        if (off + per_iter + 1024 > foot_floats) off = 0;
    }
    if (acc == 12345.678f) out[0] = acc;
}

template <int SHAPE>
void run(const float* src, float* d, int foot_floats, const char* name) {
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int iters = 2000, blocks = prop.multiProcessorCount;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k<SHAPE>), dim3(blocks), dim3(1024), 0, 0, src, d, 10, foot_floats);
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k<SHAPE>), dim3(blocks), dim3(1024), 0, 0, src, d, iters, foot_floats);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double instr_per_cu = (double)iters * 8 * 16;
    const double bytes = (SHAPE == 4 ? 1024.0 : SHAPE == 2 ? 512.0 : 256.0);
    const double cyc = ms * 1e-3 * prop.clockRate * 1e3 / instr_per_cu;
    printf("%-28s footprint %7d B per wavefront: %.3f ms, %6.2f cycles of the CU per instruction, %5.1f B/cycle/CU (at %.0f MHz)\n",
           name, foot_floats * 4, ms, cyc, bytes / cyc, prop.clockRate / 1e3);
}

static void all() {
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const size_t max_foot = 1 << 16;   // floats per wavefront window
    const size_t n = (size_t)prop.multiProcessorCount * max_foot * 4 + (1 << 20);
    float *src, *d;
    CK(hipMalloc(&src, n * sizeof(float))); CK(hipMemset(src, 0, n * sizeof(float))); CK(hipMalloc(&d, 4));
    for (int foot : {1024, 2048, 4096, 16384}) {
        run<0>(src, d, foot, "dword, 256 B contiguous");
        run<1>(src, d, foot, "dword, 4 x 64 B pieces");
        run<2>(src, d, foot, "dwordx2, 512 B contiguous");
        run<4>(src, d, foot, "dwordx4, 1 KB contiguous");
    }
}

int main() { all(); return 0; }
