// Micro-bench of the fused block2.conv backward (max-pool backward gather + data gradient + weight gradient, conv_bf16.hip) outside
// the engine, optionally with the per-phase shader-clock breakdown of wave 0:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 [-DBF_TIMING] -I train-procgen-pytorch_amd/csrc scratch/kbench_cb.hip -o scratch/kb_cb
//   ./kb_cb <n>
#include "conv_bf16.hip"
#include <cstdio>
#include <vector>
int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 8192, reps = 10;
    const size_t P = (size_t)n * 16 * 16 * 32, X = (size_t)n * 32 * 32 * 16;
    unsigned short *dp, *xi, *out, *bank; uint8_t* arg; float* slabs;
    hipMalloc(&dp, P * 2 + 4096); hipMalloc(&arg, P + 4096); hipMalloc(&xi, X * 2 + 4096); hipMalloc(&out, X * 2 + 4096);
    hipMalloc(&bank, 16 * 304 * 2 + 4096); hipMalloc(&slabs, (size_t)1024 * (4608 + 32) * 4);
    std::vector<unsigned short> h(X);
    for (size_t k = 0; k < X; ++k) h[k] = (unsigned short)((k * 2654435761u >> 16) & 0xBFFF);
    hipMemcpy(dp, h.data(), P * 2, hipMemcpyHostToDevice); hipMemcpy(xi, h.data(), X * 2, hipMemcpyHostToDevice);
    std::vector<uint8_t> ha(P);
    for (size_t k = 0; k < P; ++k) ha[k] = (uint8_t)((k * 40503u >> 7) % 9);
    hipMemcpy(arg, ha.data(), P, hipMemcpyHostToDevice);
    std::vector<unsigned short> hb(16 * 304, 0x3c00); hipMemcpy(bank, hb.data(), hb.size() * 2, hipMemcpyHostToDevice);
    hipStream_t st; hipStreamCreate(&st);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    ConvArgs a{};
    a.in = dp; a.pool_arg = arg; a.out = out; a.n = n; a.bf16 = 1; a.wbank = bank; a.wg_in = xi; a.wg_partial = slabs;
    auto run = [&]() { launch_conv_dgrad_bf16(CS_16_32_32, a, st); };
    run(); hipStreamSynchronize(st);
#ifdef BF_TIMING
    unsigned long long zero[16] = {0}; hipMemcpyToSymbol(HIP_SYMBOL(g_bf_timing), zero, sizeof zero);
#endif
    hipEventRecord(e0, st);
    for (int r = 0; r < reps; ++r) run();
    hipEventRecord(e1, st); hipStreamSynchronize(st);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const int grid = conv_bwd_fused_grid(CS_16_32_32, n);
    printf("fused block2.conv backward n=%d grid=%d: %.1f us/launch  (%s)\n", n, grid, ms * 1000 / reps, hipGetErrorString(hipGetLastError()));
#ifdef BF_TIMING
    unsigned long long t[16]; hipMemcpyFromSymbol(t, HIP_SYMBOL(g_bf_timing), sizeof t);
    const char* nm[8] = {"top barrier (+ wait for loads)", "staging stores + barrier", "gather (+ next loads)", "barrier after gather", "wgrad MFMAs", "dgrad MFMAs", "tail: reduction + slab", "prologue"};
    for (int role = 0; role < 2; ++role)
        for (int k = 0; k < 8; ++k) printf("  wave %d %-32s %9.0f cycles/WG/launch (%.0f per item)\n", role * 4, nm[k], (double)t[role * 8 + k] / grid / reps, (double)t[role * 8 + k] / grid / reps / ((double)n * 4 / grid));
#endif
    return 0;
}
