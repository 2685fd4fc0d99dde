// Micro-bench of the residual-pair forward at 32 channels @16x16 (training: all four outputs stored): the LDS-bank kernel against the
// role-pipelined one, with checksums of all four outputs (they must agree bit for bit):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 [-DPAIR_HW=8] -I train-procgen-pytorch_amd/csrc scratch/kbench_pair32.hip -o scratch/kb_pair32 ; ./kb_pair32 [n]
#include "resblock_bf16.hip"
#include <cstdio>
#include <vector>
int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 8192, reps = 10;
#ifndef PAIR_HW
#define PAIR_HW 16
#endif
#if PAIR_HW == 16
    using C = RB_32_16; using R = RBR_32_16;
#else
    using C = RB_32_8P; using R = RBR_32_8;
#endif
    const size_t X = (size_t)n * PAIR_HW * PAIR_HW * 32;
    unsigned short *x, *o[4], *banks; float* bias;
    hipMalloc(&x, X * 2 + 4096); for (auto& q : o) hipMalloc(&q, X * 2 + 4096);
    hipMalloc(&banks, (size_t)4 * C::W_ELEMS * 2 + 4096); hipMalloc(&bias, 4 * 32 * 4);
    std::vector<unsigned short> h(X); unsigned r = 12345u;
    for (auto& v : h) { r = r * 1664525u + 1013904223u; v = (unsigned short)(0x3c00u + ((r >> 20) & 0x3ffu) + ((r >> 8) & 0x8000u)); }
    hipMemcpy(x, h.data(), X * 2, hipMemcpyHostToDevice);
    std::vector<unsigned short> hb((size_t)4 * C::W_ELEMS); for (auto& v : hb) { r = r * 1664525u + 1013904223u; v = (unsigned short)(0x3800u + ((r >> 20) & 0xffu) + ((r >> 8) & 0x8000u)); }
    hipMemcpy(banks, hb.data(), hb.size() * 2, hipMemcpyHostToDevice);
    std::vector<float> hbias(4 * 32); for (auto& v : hbias) { r = r * 1664525u + 1013904223u; v = ((r >> 8) & 0xffff) / 65536.f - 0.5f; }
    hipMemcpy(bias, hbias.data(), hbias.size() * 4, hipMemcpyHostToDevice);
    ResblockPairArgs a{}; a.x = x; a.n = n; a.a1_out = o[0]; a.y1_out = o[1]; a.a2_out = o[2]; a.y2_out = o[3];
    for (int k = 0; k < 4; ++k) { a.b[k] = bias + 32 * k; a.bank[k] = banks + (size_t)k * C::W_ELEMS; }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 2; ++mode) {
        auto run = [&]() { if (mode) launch_rbp32r<R>(a, 0); else launch_rbp_t<C>(a, 0); };
        for (auto& q : o) hipMemset(q, 0, X * 2);
        for (int k = 0; k < 3; ++k) run();
        hipEventRecord(e0, 0);
        for (int k = 0; k < reps; ++k) run();
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        unsigned long long cs[4];
        std::vector<unsigned short> out(X);
        for (int t = 0; t < 4; ++t) { hipMemcpy(out.data(), o[t], X * 2, hipMemcpyDeviceToHost); cs[t] = 0; for (size_t k = 0; k < X; ++k) cs[t] = cs[t] * 1099511628211ull + out[k]; }
        printf("%s n=%d: %.1f us per launch, checksums %016llx %016llx %016llx %016llx (%s)\n", mode ? "roles  " : "lds-bank", n, ms / reps * 1e3, cs[0], cs[1], cs[2], cs[3],
               hipGetErrorString(hipGetLastError()));
    }
    return 0;
}
