// Micro-bench of the whole backward of the 16-channel residual blocks (resblock_bwd_full_bf16_kernel) outside the engine, with checksums of
// dx and of the two weight-gradient slab sets (A/B of kernel variants: dx must stay bit-identical, the slabs may move in the last bits):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 [-DRBFULL_WG_REUSE=0] -I train-procgen-pytorch_amd/csrc scratch/kbench_rb16.hip -o scratch/kb_rb16 ; ./kb_rb16 [n]
#include "resblock_bf16.hip"
#include <cstdio>
#include <cstring>
#include <vector>
int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 8192, reps = 20;
    const size_t X = (size_t)n * 32 * 32 * 16;
    unsigned short *dy, *af, *xf, *dx, *banks; float* slabs;
    const int grid = resblock_bwd_full_grid(n);
    hipMalloc(&dy, X * 2 + 4096); hipMalloc(&af, X * 2 + 4096); hipMalloc(&xf, X * 2 + 4096); hipMalloc(&dx, X * 2 + 4096);
    hipMalloc(&banks, 2 * 16 * 176 * 2 + 4096); hipMalloc(&slabs, (size_t)2 * grid * 2320 * 4);
    std::vector<unsigned short> h(X);
    auto bf = [](float f) { unsigned u; memcpy(&u, &f, 4); return (unsigned short)(u >> 16); };
    unsigned long long sd = 88172645463325252ull;
    auto rnd = [&]() { sd ^= sd << 13; sd ^= sd >> 7; sd ^= sd << 17; return (float)((sd >> 40) & 0xffff) / 32768.f - 1.f; };
    for (size_t k = 0; k < X; ++k) h[k] = bf(rnd());
    hipMemcpy(dy, h.data(), X * 2, hipMemcpyHostToDevice);
    for (size_t k = 0; k < X; ++k) h[k] = bf(rnd());
    hipMemcpy(af, h.data(), X * 2, hipMemcpyHostToDevice);
    for (size_t k = 0; k < X; ++k) h[k] = bf(rnd());
    hipMemcpy(xf, h.data(), X * 2, hipMemcpyHostToDevice);
    std::vector<unsigned short> hb(2 * 16 * 176);
    for (auto& v : hb) v = bf(0.1f * rnd());
    hipMemcpy(banks, hb.data(), hb.size() * 2, hipMemcpyHostToDevice);
    hipStream_t st; hipStreamCreate(&st);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&]() { launch_resblock_bwd_full_bf16(dy, af, xf, dx, nullptr, n, banks, banks + 16 * 176, slabs, slabs + (size_t)grid * 2320, st); };
    run(); hipStreamSynchronize(st);
#ifdef RBF_TIMING
    unsigned long long zero[8] = {0}; hipMemcpyToSymbol(HIP_SYMBOL(g_rbf_timing), zero, sizeof zero);
#endif
    hipEventRecord(e0, st);
    for (int r = 0; r < reps; ++r) run();
    hipEventRecord(e1, st); hipStreamSynchronize(st);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned short> hx(X); hipMemcpy(hx.data(), dx, X * 2, hipMemcpyDeviceToHost);
    std::vector<float> hs((size_t)2 * grid * 2320); hipMemcpy(hs.data(), slabs, hs.size() * 4, hipMemcpyDeviceToHost);
    unsigned long long cx = 1469598103934665603ull; for (auto v : hx) { cx ^= v; cx *= 1099511628211ull; }
    double s2 = 0, s1 = 0; for (size_t k = 0; k < (size_t)grid * 2320; ++k) { s2 += hs[k]; s1 += hs[(size_t)grid * 2320 + k]; }
    printf("n=%d grid=%d: %.1f us/launch  dx fnv %016llx  slab sums %.6e %.6e  (%s)\n", n, grid, ms * 1000 / reps, cx, s2, s1, hipGetErrorString(hipGetLastError()));
#ifdef RBF_TIMING
    if (RBFULL16_SPECIALISED) {
        unsigned long long t[8]; hipMemcpyFromSymbol(t, HIP_SYMBOL(g_rbf_timing), sizeof t);
        const char* nm[4] = {"staging (+wait)", "phase-1 work", "wait at mid barrier", "phase-2 work + top wait"};
        const double items = (double)n * RbFull16S::TPI / grid * reps;
        for (int role = 0; role < 2; ++role)
            for (int k = 0; k < 4; ++k) printf("  %s %-24s %8.0f cycles per item\n", role ? "wgrad" : "conv ", nm[k], (double)t[role * 4 + k] / grid / items);
    } else
    {   // slot k = cycles between mark k-1 and mark k of wave 0 (slot 0: weight-gradient phase of the previous item + loop back)
        unsigned long long t[8]; hipMemcpyFromSymbol(t, HIP_SYMBOL(g_rbf_timing), sizeof t);
        const char* nm[8] = {"wgrad phase (prev item)", "wait top barrier", "stage tiles to LDS", "wait barrier", "issue next loads", "phase A conv (da)", "wait mid barrier", "phase B conv (dx)"};
        const double items = (double)n * 4 / grid * reps;
        double tot = 0; for (int k = 0; k < 8; ++k) tot += (double)t[k];
        for (int k = 0; k < 8; ++k) printf("  %-26s %8.0f cycles per item (%4.1f %%)\n", nm[k], (double)t[k] / grid / items, 100.0 * t[k] / tot);
        printf("  total %.0f cycles per item per workgroup\n", tot / grid / items);
    }
#endif
    return 0;
}
