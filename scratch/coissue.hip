// Do MFMA and VALU instructions of DIFFERENT waves on one SIMD overlap?  512-thread workgroups (one per CU), waves 0-3 run a chain-free
// stream of 16x16x32 bf16 MFMAs, waves 4-7 a stream of fp32 FMAs (one wave of each kind per SIMD).  mode 0: both, 1: MFMA waves only,
// 2: VALU waves only, 3: every wave alternates 8 MFMAs / 32 FMAs (the lock-step shape of the conv + pool kernels).
//   hipcc --offload-arch=gfx950 -O3 scratch/coissue.hip -o scratch/kb_coissue && ./kb_coissue
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
__global__ __launch_bounds__(512, 2) void k(float* out, int iters, int mode) {
    const int wave = threadIdx.x >> 6;
    const bool mf = wave < 4;
    f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    float v[8]; for (int q = 0; q < 8; ++q) v[q] = threadIdx.x * 0.001f + q;
    const bf16x8 a = {1, 2, 3, 4, 5, 6, 7, 8}, b = {8, 7, 6, 5, 4, 3, 2, 1};
    if (mode == 3) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int q = 0; q < 8; ++q) acc[q & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[q & 3], 0, 0, 0);
#pragma unroll
            for (int q = 0; q < 32; ++q) v[q & 7] = __builtin_fmaf(v[q & 7], 1.0001f, 0.5f);
        }
    } else if (mf) {
        if (mode != 2)
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int q = 0; q < 16; ++q) acc[q & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[q & 3], 0, 0, 0);
            }
    } else {
        if (mode != 1)
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int q = 0; q < 64; ++q) v[q & 7] = __builtin_fmaf(v[q & 7], 1.0001f, 0.5f);
            }
    }
    float s = 0; for (int q = 0; q < 8; ++q) s += v[q]; for (int q = 0; q < 4; ++q) s += acc[q][0] + acc[q][3];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}
int main() {
    float* out; hipMalloc(&out, 256 * 512 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    for (int mode = 0; mode < 4; ++mode) {
        hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, out, 100, mode); hipDeviceSynchronize();
        hipEventRecord(e0); hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, out, iters, mode); hipEventRecord(e1); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        // per iteration: mode 0-2: 16 MFMAs (MFMA waves) / 64 FMAs (VALU waves); mode 3: 8 MFMAs + 32 FMAs per wave, 2 waves per SIMD
        printf("mode %d: %.3f ms  -> %.1f ns per iteration (%.1f cycles at 2.4 GHz)\n", mode, ms, ms * 1e6 / iters, ms * 1e6 / iters * 2.4);
    }
    return 0;
}
