// Micro-bench of the whole-backward residual kernels outside the engine (32 channels @16x16), optionally with the per-phase
// shader-clock breakdown of the wave-specialised kernel:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 [-DWG_TIMING] -I train-procgen-pytorch_amd/csrc scratch/kbench_rb.hip -o scratch/kb_rb
//   ./kb_rb <n> <s | p | 8>    (s = specialised, p = plain, 8 = the 8x8 blocks' kernel)
#include "resblock_bf16.hip"
#include <cstdio>
#include <cstring>
#include <vector>
int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 8192, reps = 10;
    const bool spec = argc > 2 ? argv[2][0] == 's' : true, m8 = argc > 2 && (argv[2][0] == '8' || argv[2][0] == 'q'), mq = argc > 2 && argv[2][0] == 'q';
    const size_t X = (size_t)n * (m8 ? 64 : 256) * 32;
    unsigned short *dy, *af, *xf, *dx, *banks; float* slabs;
    hipMalloc(&dy, X * 2 + 4096); hipMalloc(&af, X * 2 + 4096); hipMalloc(&xf, X * 2 + 4096); hipMalloc(&dx, X * 2 + 4096);
    hipMalloc(&banks, 2 * 32 * 304 * 2 + 4096); hipMalloc(&slabs, (size_t)2 * 1024 * 9248 * 4);
    std::vector<unsigned short> h(X);
    for (size_t k = 0; k < X; ++k) h[k] = (unsigned short)((k * 2654435761u >> 16) & 0xBFFF);      // mixed signs, finite
    hipMemcpy(dy, h.data(), X * 2, hipMemcpyHostToDevice);
    for (size_t k = 0; k < X; ++k) h[k] = (unsigned short)((k * 40503u + 12345u) >> 3 & 0xBFFF);
    hipMemcpy(af, h.data(), X * 2, hipMemcpyHostToDevice); hipMemcpy(xf, h.data() + 17, (X - 17) * 2, hipMemcpyHostToDevice);
    std::vector<unsigned short> hb(2 * 32 * 304, 0x3c00); hipMemcpy(banks, hb.data(), hb.size() * 2, hipMemcpyHostToDevice);
    hipStream_t st; hipStreamCreate(&st);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    RbFullArgs a{dy, af, xf, dx, nullptr, banks, banks + 32 * 304, slabs, slabs + (size_t)1024 * 9248, n};
    auto run = [&]() { if (mq) launch_rb_full32q(a, st); else if (m8) launch_rb_full32_t<RbFull32S>(a, st); else if (spec) launch_rb_full32s(a, st); else launch_rb_full32_t<RbFull32>(a, st); };
    run(); hipStreamSynchronize(st);
#ifdef WG_TIMING
    unsigned long long zero[8] = {0}; hipMemcpyToSymbol(HIP_SYMBOL(g_rb_timing), zero, sizeof zero);
#endif
    hipEventRecord(e0, st);
    for (int r = 0; r < reps; ++r) run();
    hipEventRecord(e1, st); hipStreamSynchronize(st);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const int grid = mq ? rb_full32q_grid(n) : m8 ? rb_full32_grid_t<RbFull32S>(n) : RB32_SPECIALISED && spec ? rb_full32s_grid(n) : rb_full32_grid_t<RbFull32>(n);
    { std::vector<unsigned short> hd(X); hipMemcpy(hd.data(), dx, X * 2, hipMemcpyDeviceToHost);
      std::vector<float> hs((size_t)2 * 1024 * 9248); hipMemcpy(hs.data(), slabs, hs.size() * 4, hipMemcpyDeviceToHost);
      unsigned long long h1 = 1469598103934665603ull, h2 = h1;
      for (size_t k = 0; k < X; ++k) h1 = (h1 ^ hd[k]) * 1099511628211ull;
      const int g = mq ? rb_full32q_grid(n) : m8 ? rb_full32_grid_t<RbFull32S>(n) : RB32_SPECIALISED && spec ? rb_full32s_grid(n) : rb_full32_grid_t<RbFull32>(n);
      for (int l = 0; l < 2; ++l) for (size_t k = 0; k < (size_t)g * 9248; ++k) { unsigned u; memcpy(&u, &hs[(size_t)l * 1024 * 9248 + k], 4); h2 = (h2 ^ u) * 1099511628211ull; }
      printf("hash dx %016llx slabs %016llx\n", h1, h2); }
    printf("%s n=%d grid=%d: %.1f us/launch  (%s)\n", mq ? "8x8 quad" : m8 ? "8x8" : spec ? "specialised" : "plain", n, grid, ms * 1000 / reps, hipGetErrorString(hipGetLastError()));
#ifdef WG_TIMING
    if (spec || mq) {
        unsigned long long t[8]; hipMemcpyFromSymbol(t, HIP_SYMBOL(g_rb_timing), sizeof t);
        const char* nm[4] = {"staging (+wait)", "phase-1 work", "wait at mid barrier", "phase-2 work + top wait"};
        for (int role = 0; role < 2; ++role)
            for (int k = 0; k < 4; ++k) printf("  %s %-24s %9.0f cycles/WG/launch (%.0f per item)\n", role ? "wgrad" : "conv ", nm[k], (double)t[role * 4 + k] / grid / reps,
                                               (double)t[role * 4 + k] / grid / reps / ((double)n * (mq ? 0.25 : 2) / grid));
    }
#endif
    return 0;
}
