// What a host->device copy in front of a dependent kernel costs, closed loop as in the rollout chain (host waits for the kernel's ticket,
// writes fresh data, uploads, launches the consumer):   hipcc --offload-arch=gfx950 -O3 scratch/copy_handover.hip -o scratch/kb_handover
//   V0 copy and kernel on one stream        V1 copy on a copy stream, event, hipStreamWaitEvent       V2 copy + 4-byte flag copy on the copy
//   stream, the kernel (launched at once) polls the flag in device memory       V3 a kernel pulls the bytes       V4 no copy (hand-shake only)
// The consumer checks one word per 4 KB page against the step number (stale data = bit 31 of the ticket).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void consume(const unsigned* data, int pages, unsigned step, const unsigned* dflag, volatile unsigned* hflag, unsigned* spins_out) {
    __shared__ int bad;
    if (threadIdx.x == 0) {
        bad = 0;
        if (dflag) {
            unsigned n = 0;
            while (__hip_atomic_load(dflag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < step && n < 4000000u) { ++n; __builtin_amdgcn_s_sleep(1); }
            if (n >= 4000000u) bad = 2;
            *spins_out = n;
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
        }
    }
    __syncthreads();
    for (int p = threadIdx.x; p < pages; p += 256) if (data[(size_t)p * 1024 + 5] != step) atomicOr(&bad, 1);
    __syncthreads();
    if (threadIdx.x == 0) { __threadfence_system(); *hflag = step | (bad ? 0x80000000u : 0u) | (bad == 2 ? 0x40000000u : 0u); }
}
__global__ __launch_bounds__(256) void pull(const u32x4* src, u32x4* dst, long long n16) {
    const long long k0 = (long long)blockIdx.x * 1024 + threadIdx.x;
    u32x4 v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { long long k = k0 + j * 256; if (k >= n16) k = n16 - 1; v[j] = __builtin_nontemporal_load(src + k); }
#pragma unroll
    for (int j = 0; j < 4; ++j) { const long long k = k0 + j * 256; if (k < n16) dst[k] = v[j]; }
}
int main(int argc, char** argv) {
    using clk = std::chrono::steady_clock;
    for (size_t bytes : {(size_t)786432, (size_t)196608, (size_t)32768}) {
        const int pages = (int)(bytes / 4096);
        unsigned *h_src, *h_tick, *d_data, *d_flag, *d_spins; volatile unsigned* h_flag;
        hipHostMalloc((void**)&h_src, bytes); hipHostMalloc((void**)&h_tick, 64); hipHostMalloc((void**)&h_flag, 64, hipHostMallocCoherent | hipHostMallocMapped);
        hipMalloc((void**)&d_data, bytes); hipMalloc((void**)&d_flag, 64); hipMalloc((void**)&d_spins, 64); hipMemset(d_flag, 0, 64); hipMemset(d_data, 0, bytes);
        hipStream_t st, cs; hipStreamCreateWithFlags(&st, hipStreamNonBlocking); hipStreamCreateWithFlags(&cs, hipStreamNonBlocking);
        hipEvent_t ev; hipEventCreateWithFlags(&ev, hipEventDisableTiming);
        for (int v = 0; v < 5; ++v) {
            *h_flag = 0; hipMemset(d_flag, 0, 64); hipDeviceSynchronize();
            const int reps = 600; double tot = 0; int stale = 0, timeouts = 0; unsigned long long spins = 0;
            unsigned step0 = (unsigned)(v * 100000 + 1);
            for (int r = 0; r < reps + 30; ++r) {
                const unsigned step = step0 + r;
                auto t0 = clk::now();
                for (int p = 0; p < pages; ++p) h_src[(size_t)p * 1024 + 5] = step;       // "env.step": fresh frames
                if (v == 0) { hipMemcpyAsync(d_data, h_src, bytes, hipMemcpyHostToDevice, st); }
                else if (v == 1) { hipMemcpyAsync(d_data, h_src, bytes, hipMemcpyHostToDevice, cs); hipEventRecord(ev, cs); hipStreamWaitEvent(st, ev, 0); }
                else if (v == 2) { *h_tick = step; hipMemcpyAsync(d_data, h_src, bytes, hipMemcpyHostToDevice, cs); hipMemcpyAsync(d_flag, h_tick, 4, hipMemcpyHostToDevice, cs); }
                else if (v == 3) { hipLaunchKernelGGL(pull, dim3((unsigned)((bytes / 16 + 1023) / 1024)), dim3(256), 0, st, (const u32x4*)h_src, (u32x4*)d_data, (long long)(bytes / 16)); }
                else { hipMemcpyAsync(d_data, h_src, 0, hipMemcpyHostToDevice, st); }
                hipLaunchKernelGGL(consume, dim3(1), dim3(256), 0, st, d_data, v == 4 ? 0 : pages, step, v == 2 ? d_flag : nullptr, h_flag, d_spins);
                unsigned got; unsigned long long sp = 0;
                while (((got = *h_flag) & 0x3fffffffu) != step) { if (++sp > (1ull << 33)) { printf("host timeout\n"); return 1; } }
                auto t1 = clk::now();
                if (v == 2) hipStreamSynchronize(cs);          // (the flag copy's host word is reused next step)
                if (r >= 30) { tot += std::chrono::duration<double, std::micro>(t1 - t0).count(); stale += (got >> 31) & 1; timeouts += (got >> 30) & 1; }
            }
            hipDeviceSynchronize();
            printf("%7zu bytes  V%d: %.1f us per step   stale %d  timeouts %d\n", bytes, v, tot / reps, stale, timeouts);
        }
        hipDeviceSynchronize();
    }
    return 0;
}
