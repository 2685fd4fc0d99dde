// Microbenchmark behind the rollout upload design: how fast do E uint8 frames get from pinned host memory
// into HBM, as (a) hipMemcpyAsync of the whole step / halves / quarters, (b) a kernel that reads the pinned buffer itself
// (zero-copy over PCIe) and writes HBM, (c) copies on one stream beside kernels on another.
//   hipcc --offload-arch=gfx950 -O3 scratch/h2d_bench.hip -o scratch/h2d_bench && scratch/h2d_bench
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

__global__ void pull_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, long long n16) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long stride = (long long)gridDim.x * blockDim.x;
    // 4 loads in flight per thread
    for (; i + 3 * stride < n16; i += 4 * stride) {
        uint4 a = src[i], b = src[i + stride];
        uint4 c = src[i + 2 * stride], d = src[i + 3 * stride];
        dst[i] = a; dst[i + stride] = b; dst[i + 2 * stride] = c; dst[i + 3 * stride] = d;
    }
    for (; i < n16; i += stride) dst[i] = src[i];
}
__global__ void spin_kernel(float* p, int iters) {
    float v = p[threadIdx.x];
    for (int k = 0; k < iters; ++k) v = v * 1.000001f + 1e-7f;
    p[threadIdx.x] = v;
}

int main() {
    const size_t full = (size_t)256 * 12288;
    void *h = nullptr, *hc = nullptr; char* d = nullptr; float* dsp = nullptr;
    CK(hipHostMalloc(&h, full));                                                      // default pinned
    CK(hipHostMalloc(&hc, full, hipHostMallocCoherent | hipHostMallocMapped));       // fine-grained
    memset(h, 1, full); memset(hc, 2, full);
    CK(hipMalloc((void**)&d, full * 4)); CK(hipMalloc((void**)&dsp, 4096));
    hipStream_t s0, s1; CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    const int iters = 300;
    for (int parts : {1, 2, 4, 8}) {
        const size_t sz = full / parts;
        for (int k = 0; k < 20; ++k) { CK(hipMemcpyAsync(d, h, sz, hipMemcpyHostToDevice, s0)); CK(hipStreamSynchronize(s0)); }
        double t0 = now();
        for (int k = 0; k < iters; ++k) { CK(hipMemcpyAsync(d, h, sz, hipMemcpyHostToDevice, s0)); CK(hipStreamSynchronize(s0)); }
        double dt = (now() - t0) / iters;
        printf("memcpyAsync+sync  %8zu B : %7.1f us  %6.1f GB/s\n", sz, dt * 1e6, sz / dt / 1e9);
        // back-to-back (no sync in between): the pipelined rate
        t0 = now();
        for (int k = 0; k < iters; ++k) CK(hipMemcpyAsync(d + (k & 3) * full, h, sz, hipMemcpyHostToDevice, s0));
        CK(hipStreamSynchronize(s0));
        dt = (now() - t0) / iters;
        printf("memcpyAsync queue %8zu B : %7.1f us  %6.1f GB/s\n", sz, dt * 1e6, sz / dt / 1e9);
    }
    for (int src_kind = 0; src_kind < 2; ++src_kind) {
        void* dev_view = nullptr;
        CK(hipHostGetDevicePointer(&dev_view, src_kind ? hc : h, 0));
        for (int grid : {64, 128, 256, 512, 1024}) {
            for (int parts : {1, 2}) {
                const long long n16 = (long long)(full / parts / 16);
                for (int k = 0; k < 10; ++k) hipLaunchKernelGGL(pull_kernel, dim3(grid), dim3(256), 0, s0, (const uint4*)dev_view, (uint4*)d, n16);
                CK(hipStreamSynchronize(s0));
                double t0 = now();
                for (int k = 0; k < iters; ++k) { hipLaunchKernelGGL(pull_kernel, dim3(grid), dim3(256), 0, s0, (const uint4*)dev_view, (uint4*)d, n16); CK(hipStreamSynchronize(s0)); }
                double dt = (now() - t0) / iters;
                printf("pull kernel %s grid %4d %8lld B : %7.1f us  %6.1f GB/s\n", src_kind ? "coherent" : "default ", grid, n16 * 16, dt * 1e6, n16 * 16 / dt / 1e9);
            }
        }
    }
    // copy on s0 beside a ~60 us kernel on s1
    {
        int spin = 20000;
        for (int k = 0; k < 5; ++k) { hipLaunchKernelGGL(spin_kernel, dim3(128), dim3(256), 0, s1, dsp, spin); }
        CK(hipStreamSynchronize(s1));
        double t0 = now();
        for (int k = 0; k < 50; ++k) { hipLaunchKernelGGL(spin_kernel, dim3(128), dim3(256), 0, s1, dsp, spin); CK(hipStreamSynchronize(s1)); }
        const double tk = (now() - t0) / 50;
        t0 = now();
        for (int k = 0; k < 50; ++k) {
            CK(hipMemcpyAsync(d, h, full / 2, hipMemcpyHostToDevice, s0));
            hipLaunchKernelGGL(spin_kernel, dim3(128), dim3(256), 0, s1, dsp, spin);
            CK(hipStreamSynchronize(s0)); CK(hipStreamSynchronize(s1));
        }
        const double tb = (now() - t0) / 50;
        printf("spin kernel alone %.1f us ; half copy + kernel on two streams %.1f us\n", tk * 1e6, tb * 1e6);
    }
    return 0;
}
