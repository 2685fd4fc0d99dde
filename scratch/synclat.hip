// How long after a kernel's last store does the host see completion?  (a) hipStreamSynchronize, (b) spinning on a flag the kernel
// writes to host-mapped memory.  hipcc --offload-arch=gfx950 -O3 scratch/synclat.hip -o scratch/kb_synclat
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void work(float* x, int iters, volatile unsigned* flag, unsigned ticket) {
    float v = x[threadIdx.x];
    for (int k = 0; k < iters; ++k) v = v * 1.0001f + 0.5f;
    x[threadIdx.x] = v;
    if (flag && threadIdx.x == 0) { __threadfence_system(); *flag = ticket; }
}
int main() {
    float* x; hipMalloc(&x, 4096);
    unsigned* flag; hipHostMalloc(&flag, 64, hipHostMallocMapped); *flag = 0;
    hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    using clk = std::chrono::steady_clock;
    for (int mode = 0; mode < 2; ++mode)
        for (int iters : {100, 20000}) {
            double tot = 0; const int reps = 2000;
            for (int r = 0; r < reps + 50; ++r) {
                auto t0 = clk::now();
                hipLaunchKernelGGL(work, dim3(1), dim3(64), 0, st, x, iters, mode ? flag : nullptr, (unsigned)(r + 1));
                if (mode) { while (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != (unsigned)(r + 1)) { } }
                else hipStreamSynchronize(st);
                auto t1 = clk::now();
                if (r >= 50) tot += std::chrono::duration<double, std::micro>(t1 - t0).count();
            }
            hipStreamSynchronize(st);
            printf("%s iters=%5d: %.2f us per launch+wait\n", mode ? "flag spin " : "stream sync", iters, tot / reps);
        }
    return 0;
}
