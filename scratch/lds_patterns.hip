// LDS access patterns of the conv / residual kernels at candidate pixel strides: cycles per wave-instruction (one wave per workgroup,
// 256 back-to-back operations of one pattern, clock64 around them).  hipcc --offload-arch=gfx950 -O3 scratch/lds_patterns.hip -o scratch/lds_pat
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef s16x4 __attribute__((address_space(3))) * lds_s16x4_ptr;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
__global__ void pat(int S, int pattern, int swz, long long* out) {
    extern __shared__ __attribute__((aligned(16))) unsigned short sm[];
    const int lane = threadIdx.x & 63, i = lane & 15, kq = lane >> 4, rq = (lane & 15) >> 2, cp = lane & 3;
    for (int e = threadIdx.x; e < 32768; e += blockDim.x) sm[e] = (unsigned short)e;
    __syncthreads();
    int off;                                       // element offset of this lane
    auto sw = [&](int pix, int chunk_elems) { return swz ? (chunk_elems ^ (((pix >> 3) & 1) * swz)) : chunk_elems; };
    if (pattern == 0) off = i * S + sw(i, kq * 8);                                   // b128 read: 16 pixels x 4 chunks (32-ch A operand)
    else if (pattern == 1 || pattern == 2) off = i * S + sw(i, kq * 4);            // b64 write / read: epilogue, mask, skip (channel quad kq)
    else if (pattern == 3) { const int pix = 16 * (kq >> 1) + 4 * (kq & 1) + rq; off = pix * S + sw(pix, 4 * cp); }   // transposing read (weight gradient)
    else { const int pix = lane >> 2; off = pix * S + sw(pix, (lane & 3) * 8); }     // b128 write: staging, 4 chunks per pixel
    const unsigned short* q[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) q[u] = sm + off + ((u + (threadIdx.x >> 6)) & 7) * 16 * S;      // 8 tiles with the same bank picture; addresses fixed before the loop
    unsigned t = 0;
    const u32x2 w2 = {1u, 2u}; const u32x4 w4 = {1u, 2u, 3u, 4u};
#pragma unroll 1
    for (int r = 0; r < 512; ++r) {
        if (pattern == 0) {
            u32x4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = *(const u32x4*)q[u];
#pragma unroll
            for (int u = 0; u < 8; ++u) t ^= v[u].x;
        } else if (pattern == 1) {
#pragma unroll
            for (int u = 0; u < 8; ++u) *(u32x2*)q[u] = w2;
        } else if (pattern == 2) {
            u32x2 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = *(const u32x2*)q[u];
#pragma unroll
            for (int u = 0; u < 8; ++u) t ^= v[u].x;
        } else if (pattern == 3) {
            s16x4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)q[u]);
#pragma unroll
            for (int u = 0; u < 8; ++u) t ^= (unsigned)v[u].x;
        } else {
#pragma unroll
            for (int u = 0; u < 8; ++u) *(u32x4*)q[u] = w4;
        }
        asm volatile("" ::: "memory");
    }
    if (t == 12345u) out[1000] = 1;
}
int main() {
    long long* d; hipMalloc(&d, 8192 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const char* nm[5] = {"b128 read (A operand)", "b64 write (epilogue)", "b64 read (mask / skip)", "tr_b64 read (wgrad)", "b128 write (staging)"};
    const int strides[] = {16, 32, 40, 48, 56, 64, 72};
    for (int p = 0; p < 5; ++p) {
        printf("%-24s", nm[p]);
        for (int S : strides) for (int swz : {0, 8, 16}) {
            if (swz && (S != 16 && S != 48 && S != 32)) continue;
            if (swz == 16 && S == 16) continue;
            hipLaunchKernelGGL(pat, dim3(256), dim3(256), 65536, 0, S, p, swz, d);
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(pat, dim3(256), dim3(256), 65536, 0, S, p, swz, d);
            hipEventRecord(e1, 0); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("  S%d%s:%5.1f", S, swz == 8 ? "^8" : swz == 16 ? "^16" : "", ms * 1e3);      // us for 4 waves x 4096 operations per CU
        }
        printf("\n");
    }
    return 0;
}
