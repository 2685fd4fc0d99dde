// Micro-bench of the fused residual-pair forward (16 channels @32x32, training: all four outputs stored) outside the engine:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 [-DRB_SWZ16=0] -I train-procgen-pytorch_amd/csrc scratch/kbench_pair.hip -o scratch/kb_pair
//   ./kb_pair <n>
#include "resblock_bf16.hip"
#include <cstdio>
#include <vector>
int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 8192, reps = 10;
    using C = RbCfg<16, 32, 32, 1, 1024>;
    const size_t X = (size_t)n * 32 * 32 * 16;
    unsigned short *x, *o[4], *banks; float* bias;
    hipMalloc(&x, X * 2 + 4096); for (auto& q : o) hipMalloc(&q, X * 2 + 4096);
    hipMalloc(&banks, (size_t)4 * C::W_ELEMS * 2 + 4096); hipMalloc(&bias, 4 * 16 * 4);
    std::vector<unsigned short> h(X); unsigned r = 12345u;
    for (auto& v : h) { r = r * 1664525u + 1013904223u; v = (unsigned short)(0x3c00u + ((r >> 20) & 0x3ffu) + ((r >> 8) & 0x8000u)); }
    hipMemcpy(x, h.data(), X * 2, hipMemcpyHostToDevice);
    std::vector<unsigned short> hb((size_t)4 * C::W_ELEMS); for (auto& v : hb) { r = r * 1664525u + 1013904223u; v = (unsigned short)(0x3800u + ((r >> 20) & 0xffu) + ((r >> 8) & 0x8000u)); }
    hipMemcpy(banks, hb.data(), hb.size() * 2, hipMemcpyHostToDevice); hipMemset(bias, 0, 4 * 16 * 4);
    ResblockPairArgs a{}; a.x = x; a.n = n; a.a1_out = o[0]; a.y1_out = o[1]; a.a2_out = o[2]; a.y2_out = o[3];
    for (int k = 0; k < 4; ++k) { a.b[k] = bias + 16 * k; a.bank[k] = banks + (size_t)k * C::W_ELEMS; }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int k = 0; k < 3; ++k) launch_rbp_t<C>(a, 0);
    hipEventRecord(e0, 0);
    for (int k = 0; k < reps; ++k) launch_rbp_t<C>(a, 0);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned short> out(X); hipMemcpy(out.data(), o[3], X * 2, hipMemcpyDeviceToHost);
    unsigned long long cs = 0; for (size_t k = 0; k < X; ++k) cs = cs * 1099511628211ull + out[k];
    printf("pair16 n=%d swz=%d: %.1f us per launch, checksum %016llx\n", n, (int)RB_SWZ16, ms / reps * 1e3, cs);
    return 0;
}
