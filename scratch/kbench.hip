// Kernel micro-bench outside the engine, optionally with a per-phase shader-clock breakdown (wave 0 of every workgroup):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 [-DWG_TIMING] -I train-procgen-pytorch_amd/csrc scratch/kbench.hip -o scratch/kbench
//   ./kbench <n> <kernel: w16 | w8 | c1f | c1w>
// Only conv_bf16.hip is compiled in; symbols it expects from the other translation units are not needed here.
#include "conv_bf16.hip"
#include <cstdio>
#include <cstring>
#include <vector>
int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 8192, reps = 10;
    const char* kn = argc > 2 ? argv[2] : "w16";
    const size_t big = (size_t)n * 64 * 64 * 16;             // largest tensor any of them touches (elements)
    unsigned short *in, *dout; float* slabs; uint8_t *frames, *arg; unsigned short* lut; float *w, *bias;
    hipMalloc(&in, big * 2 + 4096); hipMalloc(&dout, big * 2 + 4096); hipMalloc(&slabs, (size_t)1024 * 9248 * 4);
    hipMalloc(&frames, (size_t)n * 64 * 64 * 3 + 4096); hipMalloc(&arg, big / 4 + 4096); hipMalloc(&lut, 512); hipMalloc(&w, 16 * 27 * 4); hipMalloc(&bias, 64);
    std::vector<unsigned short> h(big / 4);
    for (size_t k = 0; k < h.size(); ++k) h[k] = (unsigned short)(0x3c00 + (k * 2654435761u >> 20) % 512);
    for (int q = 0; q < 4; ++q) { hipMemcpy(in + q * h.size(), h.data(), h.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dout + q * h.size(), h.data(), h.size() * 2, hipMemcpyHostToDevice); }
    std::vector<uint8_t> hf((size_t)n * 64 * 64 * 3); for (size_t k = 0; k < hf.size(); ++k) hf[k] = (uint8_t)(k * 2654435761u >> 13);
    hipMemcpy(frames, hf.data(), hf.size(), hipMemcpyHostToDevice);
    std::vector<uint8_t> ha(big / 4); for (size_t k = 0; k < ha.size(); ++k) ha[k] = (uint8_t)((k * 2654435761u >> 11) % 9);
    hipMemcpy(arg, ha.data(), ha.size(), hipMemcpyHostToDevice);
    std::vector<unsigned short> hl(256); for (int k = 0; k < 256; ++k) { float f = k / 255.f; unsigned u; memcpy(&u, &f, 4); hl[k] = (unsigned short)((u + 0x7fff + ((u >> 16) & 1)) >> 16); }
    hipMemcpy(lut, hl.data(), 512, hipMemcpyHostToDevice);
    std::vector<float> hw(16 * 27, 0.01f); hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice); hipMemset(bias, 0, 64);
    hipStream_t st; hipStreamCreate(&st);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    WgradArgs a{}; a.in = in; a.dout = dout; a.partial = slabs; a.n = n; a.relu_in = 1; a.bf16 = 1;
    ConvArgs c{}; c.in = frames; c.w = w; c.bias = bias; c.n = n; c.bf16 = 1; c.lut16 = lut;
    int32_t* idx; hipMalloc(&idx, (size_t)n * 4);
    { std::vector<int32_t> hi(n); for (int k = 0; k < n; ++k) hi[k] = (int)(((long long)k * 4099) % n); hipMemcpy(idx, hi.data(), (size_t)n * 4, hipMemcpyHostToDevice); }
    if (argc > 3) c.idx = idx;                        // any third argument: frames gathered through an index list, as a minibatch does
    int grid = 0;
    auto run = [&]() {
        if (!strcmp(kn, "w16")) { launch_conv_wgrad_bf16(CS_32_32_16, a, st); grid = wgrad_grid_bf16(CS_32_32_16, n); }
        else if (!strcmp(kn, "w8")) { launch_conv_wgrad_bf16(CS_32_32_8, a, st); grid = wgrad_grid_bf16(CS_32_32_8, n); }
        else if (!strcmp(kn, "c1f")) { launch_conv1_pool_fwd_bf16(c, lut, in, arg, st); grid = n * 8 > 1024 ? 1024 : n * 8; }
        else { WgradArgs b = a; b.in = frames; b.lut16 = lut; b.pool_arg = arg; if (argc > 3) b.idx = idx; launch_conv1_wgrad_bf16(b, lut, st); grid = c1_grid(n); }
    };
    run(); hipStreamSynchronize(st);
#ifdef WG_TIMING
    unsigned long long zero[8] = {0}; hipMemcpyToSymbol(HIP_SYMBOL(g_wg_timing), zero, sizeof zero);
#endif
    hipEventRecord(e0, st);
    for (int r = 0; r < reps; ++r) run();
    hipEventRecord(e1, st); hipStreamSynchronize(st);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%s n=%d grid=%d: %.1f us/launch  (%s)\n", kn, n, grid, ms * 1000 / reps, hipGetErrorString(hipGetLastError()));
#ifdef WG_TIMING
    unsigned long long t[8]; hipMemcpyFromSymbol(t, HIP_SYMBOL(g_wg_timing), sizeof t);
    unsigned long long tot = 0; for (int k = 0; k < 8; ++k) tot += t[k];
    for (int k = 0; k < 8; ++k) printf("  phase %d %10.0f cycles/WG/launch  %5.1f %%\n", k, (double)t[k] / grid / reps, 100.0 * t[k] / tot);
#endif
    return 0;
}
