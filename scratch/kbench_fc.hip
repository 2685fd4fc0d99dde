// Micro-bench of the embedder.fc kernels (fc_bf16.hip) at the training minibatch size, outside the engine:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 [-DFC_TIMING] -I train-procgen-pytorch_amd/csrc scratch/kbench_fc.hip -o scratch/kb_fc ;  ./kb_fc <n>
#include "fc_bf16.hip"
#include <cstdio>
#include <vector>
int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 8192, reps = 20;
    unsigned short *x, *wp, *wt, *dx; float *dy, *y, *bias, *gw, *ws;
    hipMalloc(&x, (size_t)n * 2048 * 2); hipMalloc(&dx, (size_t)n * 2048 * 2); hipMalloc(&wp, 256 * 2048 * 2); hipMalloc(&wt, 256 * 2048 * 2);
    hipMalloc(&dy, (size_t)n * 256 * 4); hipMalloc(&y, (size_t)n * 256 * 4); hipMalloc(&bias, 1024); hipMalloc(&gw, 256 * 2048 * 4); hipMalloc(&ws, (size_t)8 << 22);
    std::vector<unsigned short> h((size_t)n * 2048);
    for (size_t k = 0; k < h.size(); ++k) h[k] = (unsigned short)(0x3c00 + ((k * 2654435761u >> 17) & 0x83ff));
    hipMemcpy(x, h.data(), h.size() * 2, hipMemcpyHostToDevice); hipMemcpy(wp, h.data(), 256 * 2048 * 2, hipMemcpyHostToDevice); hipMemcpy(wt, h.data() + 999, 256 * 2048 * 2, hipMemcpyHostToDevice);
    std::vector<float> f((size_t)n * 256, 0.01f); hipMemcpy(dy, f.data(), f.size() * 4, hipMemcpyHostToDevice); hipMemset(bias, 0, 1024); hipMemset(gw, 0, 256 * 2048 * 4);
    hipStream_t st; hipStreamCreate(&st);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto timeit = [&](const char* nm, auto&& fn) {
        fn(); hipStreamSynchronize(st);
        hipEventRecord(e0, st);
        for (int r = 0; r < reps; ++r) fn();
        hipEventRecord(e1, st); hipStreamSynchronize(st);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-10s n=%d: %.1f us/launch  (%s)\n", nm, n, ms * 1000 / reps, hipGetErrorString(hipGetLastError()));
    };
    timeit("forward", [&] { launch_fc_fwd_bf16(x, wp, bias, y, n, st); });
#ifdef FC_TIMING
    unsigned long long zero[8] = {0}; hipMemcpyToSymbol(HIP_SYMBOL(g_fc_timing), zero, sizeof zero);
#endif
    timeit("dgrad", [&] { launch_fc_dgrad_bf16(dy, wt, x, dx, n, st); });
#ifdef FC_TIMING
    unsigned long long t[8]; hipMemcpyFromSymbol(t, HIP_SYMBOL(g_fc_timing), sizeof t);
    const double wgs = (double)(2048 / (64 * FD_STEPS)) * (n / 128) * (reps + 1);
    const char* nm[5] = {"dy fragments", "staged slice -> LDS", "barrier", "next slice + MFMAs", "epilogue"};
    for (int q = 0; q < 5; ++q) printf("    %-22s %9.0f cycles per workgroup\n", nm[q], t[q] / wgs);
#endif
    timeit("wgrad", [&] { launch_fc_tn(dy, x, gw, ws, (size_t)8 << 20, 256, 2048, n, st); });
    return 0;
}
