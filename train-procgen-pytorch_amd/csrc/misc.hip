// HBM-bound scan / elementwise / reduction kernels of the PPO path and the generic MFMA GEMM used
// by the linear layers.  gfx950 only (64-lane waves).  Every reduction has a fixed summation
// order: results are bitwise reproducible run to run (no float atomics anywhere).
#include "common.h"
#include <hip/hip_ext.h>
#include <math.h>
#include <stdlib.h>

static thread_local const char* tl_launch_err = nullptr;
void mi_launch_fail(const char* msg) { if (!tl_launch_err) tl_launch_err = msg; }
const char* mi_launch_failed_take() { const char* m = tl_launch_err; tl_launch_err = nullptr; return m; }

#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

template <typename T>
__device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_down(v, o, 64));
    return v;
}
// sum over a 256-thread block, result valid in thread 0; sbuf >= 4 entries
template <typename T>
__device__ __forceinline__ T block_sum256(T v, T* sbuf) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) sbuf[w] = v;
    __syncthreads();
    return sbuf[0] + sbuf[1] + sbuf[2] + sbuf[3];
}

// ------------------------------------------------------------------------------------------ GEMM
// 64x64 output tile per workgroup, K tile 16, four waves as 2x2 of 32x32; K permuted so that lane
// quarter q owns k = 4q..4q+3 of the K tile (one ds_read_b128 per operand per 4 MFMAs).
// Used for: embedder.fc, the policy/value heads and the MLP embedder (forward, dgrad, wgrad) --
// nn.Linear in common/model.py:176,199 / :966-971 and common/policy.py:39-40,75,80.
__device__ __forceinline__ unsigned short f2bf_g(float x) { __bf16 h = (__bf16)x; return __builtin_bit_cast(unsigned short, h); }
__device__ __forceinline__ float gemm_ld(const float* p, long long o, int bf16) {
    return bf16 ? __uint_as_float(((unsigned)((const unsigned short*)p)[o]) << 16) : p[o];
}
// K tile 32; the next tile's global loads are issued into registers before the MFMAs of the current one.
__global__ __launch_bounds__(256) void gemm_kernel(GemmArgs g, int k_chunk, float* ws) {
    __shared__ __attribute__((aligned(16))) float As[64 * 36];
    __shared__ __attribute__((aligned(16))) float Bs[64 * 36];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 15, q = lane >> 4, wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
    const int kbeg = blockIdx.z * k_chunk, kend = min(g.K, kbeg + k_chunk);
    f32x4 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const bool a_kc = (g.sak == 1), b_kc = (g.sbk == 1);
    float ra[8], rb[8];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int l = tid + e * 256;
            int m, k;
            if (a_kc) { m = l >> 5; k = l & 31; } else { m = l & 63; k = l >> 6; }
            float v = 0.f;
            if (m0 + m < g.M && k0 + k < kend) v = gemm_ld(g.A, (long long)(m0 + m) * g.sam + (long long)(k0 + k) * g.sak, g.a_bf16);
            ra[e] = v;
            int n;
            if (b_kc) { n = l >> 5; k = l & 31; } else { n = l & 63; k = l >> 6; }
            v = 0.f;
            if (n0 + n < g.N && k0 + k < kend) v = gemm_ld(g.B, (long long)(k0 + k) * g.sbk + (long long)(n0 + n) * g.sbn, g.b_bf16);
            rb[e] = v;
        }
    };
    if (kbeg < kend) fetch(kbeg);
    for (int k0 = kbeg; k0 < kend; k0 += 32) {
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int l = tid + e * 256;
            int m, k;
            if (a_kc) { m = l >> 5; k = l & 31; } else { m = l & 63; k = l >> 6; }
            As[m * 36 + k] = g.relu_a ? fmaxf(ra[e], 0.f) : ra[e];
            int n;
            if (b_kc) { n = l >> 5; k = l & 31; } else { n = l & 63; k = l >> 6; }
            Bs[n * 36 + k] = g.relu_b ? fmaxf(rb[e], 0.f) : rb[e];
        }
        __syncthreads();
        if (k0 + 32 < kend) fetch(k0 + 32);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            f32x4 av[2], bv[2];
#pragma unroll
            for (int a = 0; a < 2; ++a) av[a] = *(const f32x4*)(As + (wm * 32 + a * 16 + i) * 36 + q * 4 + h * 16);
#pragma unroll
            for (int b = 0; b < 2; ++b) bv[b] = *(const f32x4*)(Bs + (wn * 32 + b * 16 + i) * 36 + q * 4 + h * 16);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b) acc[a][b] = MFMA16(av[a][e], bv[b][e], acc[a][b]);
        }
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + wm * 32 + a * 16 + q * 4 + r, n = n0 + wn * 32 + b * 16 + i;
                if (m < g.M && n < g.N) {
                    float v = acc[a][b][r];
                    if (ws) { ws[((long long)blockIdx.z * g.M + m) * g.N + n] = v; continue; }
                    const long long o = (long long)m * g.ldc + n;
                    if (g.bias) v += g.bias[n];
                    if (g.relu_out) v = fmaxf(v, 0.f);
                    if (g.mask) v = gemm_ld(g.mask, o, g.mask_bf16) > 0.f ? v : 0.f;
                    if (g.c_bf16) { ((unsigned short*)g.C)[o] = f2bf_g(v); continue; }
                    if (g.accumulate) v += g.C[o];
                    g.C[o] = v;
                }
            }
}

__global__ void gemm_splitk_reduce(const float* ws, int split, GemmArgs g) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (long long)g.M * g.N) return;
    float v = 0.f;
    {   // 8 loads in flight, adds in split order (a plain run-time loop waited for every load before issuing the next: 21 us for 64 splits)
        const long long mn = (long long)g.M * g.N;
        int z = 0;
        for (; z + 8 <= split; z += 8) {
            float t[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) t[q] = ws[(long long)(z + q) * mn + e];
#pragma unroll
            for (int q = 0; q < 8; ++q) v += t[q];
        }
        for (; z < split; ++z) v += ws[(long long)z * mn + e];
    }
    const int n = (int)(e % g.N);
    const long long o = (e / g.N) * g.ldc + n;
    if (g.bias) v += g.bias[n];
    if (g.relu_out) v = fmaxf(v, 0.f);
    if (g.mask) {
        const float mv = g.mask_bf16 ? __uint_as_float(((unsigned)((const unsigned short*)g.mask)[o]) << 16) : g.mask[o];
        v = mv > 0.f ? v : 0.f;
    }
    if (g.c_bf16) { ((unsigned short*)g.C)[o] = f2bf_g(v); return; }
    if (g.accumulate) v += g.C[o];
    g.C[o] = v;
}

// split-K slabs live in the caller's workspace (GemmArgs.ws, one per context: two contexts on two streams must not share it)
void launch_gemm(const GemmArgs& g, hipStream_t st) {
    if (g.M <= 0 || g.N <= 0) return;
    const int tm = (g.M + 63) / 64, tn = (g.N + 63) / 64;
    int split = 1;
    // few output tiles and a long K (weight gradients: K = batch; rollout-sized forward GEMMs: M = n_envs):
    // split K over workgroups, slabs summed in a fixed order by the reduce kernel (which carries the epilogue).
    if (g.K >= 512 && tm * tn < 192 && g.ws) {
        split = 512 / (tm * tn);
        if (split > g.K / 128) split = g.K / 128;
        if (split < 1) split = 1;
        while (split > 1 && (size_t)split * g.M * g.N > g.ws_floats) --split;
    }
    int k_chunk = ((g.K + split - 1) / split + 31) / 32 * 32;
    if (split == 1) {
        hipLaunchKernelGGL(gemm_kernel, dim3(tn, tm, 1), dim3(256), 0, st, g, k_chunk, (float*)nullptr);
    } else {
        split = (g.K + k_chunk - 1) / k_chunk;
        hipLaunchKernelGGL(gemm_kernel, dim3(tn, tm, split), dim3(256), 0, st, g, k_chunk, g.ws);
        const long long tot = (long long)g.M * g.N;
        hipLaunchKernelGGL(gemm_splitk_reduce, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st,
                           (const float*)g.ws, split, g);
    }
}

// ------------------------------------------------------------------------------------------ max pool 3/2/1
// nn.MaxPool2d(kernel_size=3, stride=2, padding=1) (common/model.py:158), NHWC.  The winner is the
// FIRST maximum in (ky,kx) scan order (strict >), which is where torch's CPU kernel routes the
// gradient; its window-relative position is kept as one byte per output for the backward pass.
template <int HW, int C>
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                          uint8_t* __restrict__ arg, int n) {
    constexpr int HO = HW / 2, C4 = C / 4;
    const unsigned e = blockIdx.x * 256u + threadIdx.x;          // one thread = 4 channels of one output pixel
    const unsigned tot = (unsigned)n * HO * HO * C4;
    if (e >= tot) return;
    const unsigned c4 = e % C4, ox = (e / C4) % HO, oy = (e / (C4 * HO)) % HO, img = e / (C4 * HO * HO);
    f32x4 best = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    unsigned bi[4] = {0, 0, 0, 0};
    bool first = true;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int y = 2 * (int)oy - 1 + ky, x = 2 * (int)ox - 1 + kx;
            if (y < 0 || y >= HW || x < 0 || x >= HW) continue;
            const f32x4 v = *(const f32x4*)(in + (((size_t)img * HW + y) * HW + x) * C + c4 * 4);
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (first || v[k] > best[k] || v[k] != v[k]) { best[k] = v[k]; bi[k] = ky * 3 + kx; }
            first = false;
        }
    *(f32x4*)(out + (size_t)e * 4) = best;
    *(uint32_t*)(arg + (size_t)e * 4) = bi[0] | (bi[1] << 8) | (bi[2] << 16) | (bi[3] << 24);
}

template <int HW, int C>
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const float* __restrict__ dout, const uint8_t* __restrict__ arg,
                                                          float* __restrict__ din, int n) {
    constexpr int HO = HW / 2, C4 = C / 4;
    const unsigned e = blockIdx.x * 256u + threadIdx.x;          // 4 channels of one INPUT pixel
    const unsigned tot = (unsigned)n * HW * HW * C4;
    if (e >= tot) return;
    const unsigned c4 = e % C4, x = (e / C4) % HW, y = (e / (C4 * HW)) % HW, img = e / (C4 * HW * HW);
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    // windows oy with 2*oy-1 <= y <= 2*oy+1
    for (unsigned oy = y / 2; oy <= (y + 1) / 2; ++oy) {
        if (oy >= HO) continue;
        for (unsigned ox = x / 2; ox <= (x + 1) / 2; ++ox) {
            if (ox >= HO) continue;
            const size_t o = ((((size_t)img * HO + oy) * HO + ox) * C4 + c4) * 4;
            const unsigned pos = (y - (2 * oy - 1)) * 3 + (x - (2 * ox - 1));
            const uint32_t a = *(const uint32_t*)(arg + o);
            const f32x4 d = *(const f32x4*)(dout + o);
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (((a >> (8 * k)) & 0xffu) == pos) s[k] += d[k];
        }
    }
    *(f32x4*)(din + (size_t)e * 4) = s;
}

__device__ __forceinline__ float bfw_lo(unsigned w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bfw_hi(unsigned w) { return __uint_as_float(w & 0xffff0000u); }
__device__ __forceinline__ unsigned short f2bf_m(float x) { __bf16 h = (__bf16)x; return __builtin_bit_cast(unsigned short, h); }

// bf16 storage: 8 channels (16 bytes) per thread; comparisons on the exact bf16 values
template <int HW, int C>
__global__ __launch_bounds__(256) void maxpool_fwd_bf16_kernel(const unsigned short* __restrict__ in, unsigned short* __restrict__ out,
                                                               uint8_t* __restrict__ arg, int n) {
    constexpr int HO = HW / 2, C8 = C / 8;
    const unsigned e = blockIdx.x * 256u + threadIdx.x;
    const unsigned tot = (unsigned)n * HO * HO * C8;
    if (e >= tot) return;
    const unsigned c8 = e % C8, ox = (e / C8) % HO, oy = (e / (C8 * HO)) % HO, img = e / (C8 * HO * HO);
    float best[8];
    unsigned bw[4] = {0, 0, 0, 0};       // winning bf16 bits, packed
    unsigned bi[8];
    bool first = true;
#pragma unroll
    for (int k = 0; k < 8; ++k) { best[k] = -INFINITY; bi[k] = 0; }
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int y = 2 * (int)oy - 1 + ky, x = 2 * (int)ox - 1 + kx;
            if (y < 0 || y >= HW || x < 0 || x >= HW) continue;
            const uint4 u = *(const uint4*)(in + (((size_t)img * HW + y) * HW + x) * C + c8 * 8);
            const unsigned w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float v = (k & 1) ? bfw_hi(w[k >> 1]) : bfw_lo(w[k >> 1]);
                if (first || v > best[k] || v != v) {
                    best[k] = v; bi[k] = ky * 3 + kx;
                    const unsigned bits = (k & 1) ? (w[k >> 1] >> 16) : (w[k >> 1] & 0xffffu);
                    bw[k >> 1] = (k & 1) ? ((bw[k >> 1] & 0x0000ffffu) | (bits << 16)) : ((bw[k >> 1] & 0xffff0000u) | bits);
                }
            }
            first = false;
        }
    *(uint4*)(out + (size_t)e * 8) = (uint4){bw[0], bw[1], bw[2], bw[3]};
    *(uint2*)(arg + (size_t)e * 8) = (uint2){bi[0] | (bi[1] << 8) | (bi[2] << 16) | (bi[3] << 24), bi[4] | (bi[5] << 8) | (bi[6] << 16) | (bi[7] << 24)};
}

template <int HW, int C>
__global__ __launch_bounds__(256) void maxpool_bwd_bf16_kernel(const unsigned short* __restrict__ dout, const uint8_t* __restrict__ arg,
                                                               unsigned short* __restrict__ din, int n) {
    constexpr int HO = HW / 2, C8 = C / 8;
    const unsigned e = blockIdx.x * 256u + threadIdx.x;
    const unsigned tot = (unsigned)n * HW * HW * C8;
    if (e >= tot) return;
    const unsigned c8 = e % C8, x = (e / C8) % HW, y = (e / (C8 * HW)) % HW, img = e / (C8 * HW * HW);
    float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (unsigned oy = y / 2; oy <= (y + 1) / 2; ++oy) {
        if (oy >= HO) continue;
        for (unsigned ox = x / 2; ox <= (x + 1) / 2; ++ox) {
            if (ox >= HO) continue;
            const size_t o = ((((size_t)img * HO + oy) * HO + ox) * C8 + c8) * 8;
            const unsigned pos = (y - (2 * oy - 1)) * 3 + (x - (2 * ox - 1));
            const uint2 a = *(const uint2*)(arg + o);
            const uint4 d = *(const uint4*)(dout + o);
            const unsigned w[4] = {d.x, d.y, d.z, d.w};
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const unsigned ak = ((k < 4 ? a.x : a.y) >> (8 * (k & 3))) & 0xffu;
                if (ak == pos) s[k] += (k & 1) ? bfw_hi(w[k >> 1]) : bfw_lo(w[k >> 1]);
            }
        }
    }
    uint4 r;
    r.x = f2bf_m(s[0]) | ((unsigned)f2bf_m(s[1]) << 16); r.y = f2bf_m(s[2]) | ((unsigned)f2bf_m(s[3]) << 16);
    r.z = f2bf_m(s[4]) | ((unsigned)f2bf_m(s[5]) << 16); r.w = f2bf_m(s[6]) | ((unsigned)f2bf_m(s[7]) << 16);
    *(uint4*)(din + (size_t)e * 8) = r;
}

template <int HW, int C>
static void pool_fwd_bf_t(const void* in, void* out, uint8_t* arg, int n, hipStream_t st) {
    const unsigned tot = (unsigned)n * (HW / 2) * (HW / 2) * (C / 8);
    hipLaunchKernelGGL((maxpool_fwd_bf16_kernel<HW, C>), dim3((tot + 255) / 256), dim3(256), 0, st, (const unsigned short*)in, (unsigned short*)out, arg, n);
}
template <int HW, int C>
static void pool_bwd_bf_t(const void* dout, const uint8_t* arg, void* din, int n, hipStream_t st) {
    const unsigned tot = (unsigned)n * HW * HW * (C / 8);
    hipLaunchKernelGGL((maxpool_bwd_bf16_kernel<HW, C>), dim3((tot + 255) / 256), dim3(256), 0, st, (const unsigned short*)dout, arg, (unsigned short*)din, n);
}
void launch_maxpool_fwd_bf16(const void* in, void* out, uint8_t* arg, int n, int hw, int c, hipStream_t st) {
    if (n <= 0) return;
    if (hw == 64 && c == 16) pool_fwd_bf_t<64, 16>(in, out, arg, n, st);
    else if (hw == 32 && c == 32) pool_fwd_bf_t<32, 32>(in, out, arg, n, st);
    else if (hw == 16 && c == 32) pool_fwd_bf_t<16, 32>(in, out, arg, n, st);
    else mi_launch_fail("max pool (bf16): unsupported image size / channel count");
}
void launch_maxpool_bwd_bf16(const void* dout, const uint8_t* arg, void* din, int n, int hw, int c, hipStream_t st) {
    if (n <= 0) return;
    if (hw == 64 && c == 16) pool_bwd_bf_t<64, 16>(dout, arg, din, n, st);
    else if (hw == 32 && c == 32) pool_bwd_bf_t<32, 32>(dout, arg, din, n, st);
    else if (hw == 16 && c == 32) pool_bwd_bf_t<16, 32>(dout, arg, din, n, st);
    else mi_launch_fail("max pool backward (bf16): unsupported image size / channel count");
}

template <int HW, int C>
static void pool_fwd_t(const float* in, float* out, uint8_t* arg, int n, hipStream_t st) {
    const unsigned tot = (unsigned)n * (HW / 2) * (HW / 2) * (C / 4);
    hipLaunchKernelGGL((maxpool_fwd_kernel<HW, C>), dim3((tot + 255) / 256), dim3(256), 0, st, in, out, arg, n);
}
template <int HW, int C>
static void pool_bwd_t(const float* dout, const uint8_t* arg, float* din, int n, hipStream_t st) {
    const unsigned tot = (unsigned)n * HW * HW * (C / 4);
    hipLaunchKernelGGL((maxpool_bwd_kernel<HW, C>), dim3((tot + 255) / 256), dim3(256), 0, st, dout, arg, din, n);
}
void launch_maxpool_fwd(const float* in, float* out, uint8_t* arg, int n, int hw, int c, hipStream_t st) {
    if (n <= 0) return;
    if (hw == 64 && c == 16) pool_fwd_t<64, 16>(in, out, arg, n, st);
    else if (hw == 32 && c == 32) pool_fwd_t<32, 32>(in, out, arg, n, st);
    else if (hw == 16 && c == 32) pool_fwd_t<16, 32>(in, out, arg, n, st);
    else mi_launch_fail("max pool: unsupported image size / channel count");
}
void launch_maxpool_bwd(const float* dout, const uint8_t* arg, float* din, int n, int hw, int c, hipStream_t st) {
    if (n <= 0) return;
    if (hw == 64 && c == 16) pool_bwd_t<64, 16>(dout, arg, din, n, st);
    else if (hw == 32 && c == 32) pool_bwd_t<32, 32>(dout, arg, din, n, st);
    else if (hw == 16 && c == 32) pool_bwd_t<16, 32>(dout, arg, din, n, st);
    else mi_launch_fail("max pool backward: unsupported image size / channel count");
}

// ------------------------------------------------------------------------------------------ slab / column reductions
__global__ __launch_bounds__(1024) void reduce_slabs_kernel(const float* __restrict__ partial, int nslab, int slab_len,
                                                            float* dst_w, int n_w, float* dst_b, int n_b) {
    __shared__ float sh[32][33];
    const int ex = threadIdx.x & 31, g = threadIdx.x >> 5;        // 32 elements x 32 slab groups
    const int e = blockIdx.x * 32 + ex;
    float s = 0.f;
    if (e < slab_len)
        for (int b = g; b < nslab; b += 32) s += partial[(long long)b * slab_len + e];
    sh[g][ex] = s;
    __syncthreads();
    if (g == 0 && e < slab_len) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 32; ++k) t += sh[k][ex];
        if (e < n_w) dst_w[e] += t;
        else if (e - n_w < n_b) dst_b[e - n_w] += t;
    }
}
// All conv layers of one backward pass in ONE launch (blockIdx.y = layer): each layer's workgroups left their slabs in
// its own region of the workspace.  Same per-element summation order as reduce_slabs_kernel.
__global__ __launch_bounds__(1024) void reduce_all_slabs_kernel(const float* __restrict__ ws, float* __restrict__ grads, const SlabDesc* __restrict__ desc) {
    __shared__ f32x4 sh[32][33];
    const SlabDesc d = desc[blockIdx.y];
    const int ex = threadIdx.x & 31, g = threadIdx.x >> 5;
    const int e = (blockIdx.x * 32 + ex) * 4;                     // 4 consecutive elements per thread (slab lengths, offsets and the
    if (blockIdx.x * 128 >= d.slab_len) return;                   // weight / bias boundary are multiples of 4): 512-byte rows per half-wave
    const float* partial = ws + d.src_off;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (e < d.slab_len) {
        int b = g;
        for (; b + 96 < d.nslab; b += 128) {                      // 4 loads in flight; adds stay in slab order
            const f32x4 v0 = *(const f32x4*)(partial + (long long)b * d.slab_len + e), v1 = *(const f32x4*)(partial + (long long)(b + 32) * d.slab_len + e);
            const f32x4 v2 = *(const f32x4*)(partial + (long long)(b + 64) * d.slab_len + e), v3 = *(const f32x4*)(partial + (long long)(b + 96) * d.slab_len + e);
            s += v0; s += v1; s += v2; s += v3;
        }
        for (; b < d.nslab; b += 32) s += *(const f32x4*)(partial + (long long)b * d.slab_len + e);
    }
    sh[g][ex] = s;
    __syncthreads();
    if (g == 0 && e < d.slab_len) {
        f32x4 t = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 8                                                   // (all 32 reads in flight spilled 8 registers to scratch memory)
        for (int k = 0; k < 32; ++k) t += sh[k][ex];
        float* dst = (e < d.n_w) ? grads + d.w_off + e : grads + d.b_off + e - d.n_w;
        dst[0] += t.x; dst[1] += t.y; dst[2] += t.z; dst[3] += t.w;
    }
}
void launch_reduce_all_slabs(const float* ws, float* grads, const SlabDesc* d_desc, int n_desc, int max_slab_len, hipStream_t st) {
    if (n_desc <= 0) return;
    hipLaunchKernelGGL(reduce_all_slabs_kernel, dim3((max_slab_len + 127) / 128, n_desc), dim3(1024), 0, st, ws, grads, d_desc);
}
void launch_reduce_slabs(const float* partial, int nslab, int slab_len, float* dst_w, int n_w, float* dst_b, int n_b, hipStream_t st) {
    if (nslab <= 0) return;
    hipLaunchKernelGGL(reduce_slabs_kernel, dim3((slab_len + 31) / 32), dim3(1024), 0, st, partial, nslab, slab_len, dst_w, n_w, dst_b, n_b);
}

// db[n] += sum_m dY[m][n]   (bias gradients of the linear layers); 64 row groups x fixed order
__global__ void colsum_partial_kernel(const float* dY, int M, int N, int ld, float* part) {
    const int n = blockIdx.x * 64 + (threadIdx.x & 63), rg = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int groups = gridDim.y * 4;
    float s = 0.f;
    if (n < N) {
        int m = rg;
        for (; m + 7 * groups < M; m += 8 * groups) {        // 8 loads in flight; the adds keep row order
            float v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = dY[(long long)(m + q * groups) * ld + n];
#pragma unroll
            for (int q = 0; q < 8; ++q) s += v[q];
        }
        for (; m < M; m += groups) s += dY[(long long)m * ld + n];
        part[(long long)rg * N + n] = s;
    }
}
void launch_colsum_acc(const float* dY, int M, int N, int ld, float* db, float* col_ws, hipStream_t st) {
    if (M <= 0) return;
    const int gy = (M >= 4096 && N <= 1024) ? 64 : 16;         // workspace: 64 x 4096 floats (engine.hip)
    hipLaunchKernelGGL(colsum_partial_kernel, dim3((N + 63) / 64, gy), dim3(256), 0, st, dY, M, N, ld, col_ws);
    launch_reduce_slabs(col_ws, gy * 4, N, db, N, nullptr, 0, st);
}

// ------------------------------------------------------------------------------------------ heads forward (training-sized batches)
// hout[s][o] = feat[s] . Wh[o] + bh[o] for the (A+1) <= 16 head outputs, H == 256 (policy.py:74-80).  The generic GEMM ran this
// 8192 x 16 x 256 product as 16-wide tiles of a 64-wide kernel plus a split-K pass (19 us); here 16 rows and the head matrix go
// through LDS and thread (row, output) owns one dot product, summed in the order of heads_sample_kernel -- the rollout's logits and
// the update's logits of one observation under one set of weights are the same bits.
__global__ __launch_bounds__(256) void heads_fwd_kernel(const float* __restrict__ feat, const float* __restrict__ Wh, const float* __restrict__ bh,
                                                        float* __restrict__ hout, int n, int O) {
    __shared__ __attribute__((aligned(16))) float s_f[16 * 260];
    __shared__ __attribute__((aligned(16))) float s_w[16 * 260];
    const int tid = threadIdx.x, e0 = blockIdx.x * 16, el = tid >> 4, o = tid & 15;
    f32x4 rf[4], rw[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int k = tid + j * 256, r = k >> 6, kk = (k & 63) * 4;
        const int rr = e0 + r < n ? e0 + r : n - 1;
        rf[j] = *(const f32x4*)(feat + (long long)rr * 256 + kk);
        rw[j] = *(const f32x4*)(Wh + (long long)(r < O ? r : O - 1) * 256 + kk);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) { const int k = tid + j * 256; *(f32x4*)(s_f + (k >> 6) * 260 + (k & 63) * 4) = rf[j]; *(f32x4*)(s_w + (k >> 6) * 260 + (k & 63) * 4) = rw[j]; }
    __syncthreads();
    const float* w = s_w + o * 260;
    const float* f = s_f + el * 260;
    float acc = 0.f;
#pragma unroll 8
    for (int k = 0; k < 256; k += 4) {
        const f32x4 fv = *(const f32x4*)(f + k), ww = *(const f32x4*)(w + k);
        acc += fv.x * ww.x + fv.y * ww.y + fv.z * ww.z + fv.w * ww.w;
    }
    if (o < O && e0 + el < n) hout[(long long)(e0 + el) * O + o] = acc + bh[o];
}
void launch_heads_fwd(const float* feat, const float* Wh, const float* bh, float* hout, int n, int O, hipStream_t st) {
    if (n <= 0) return;
    hipLaunchKernelGGL(heads_fwd_kernel, dim3((n + 15) / 16), dim3(256), 0, st, feat, Wh, bh, hout, n, O);
}

// ------------------------------------------------------------------------------------------ heads backward in one launch
// The policy / value heads are one 256 x (A+1) linear layer (policy.py:74-80): their data gradient, weight gradient and bias gradient
// used to be two GEMM launches (each too small to fill a matrix-core tile grid), a split-K reduce and a two-stage column sum -- five
// launches of 5-14 us for 67 MFLOP.  One thread per feature column k: for every row of the workgroup's chunk it forms
// dfeat[s][k] = (feat > 0) * sum_o dY[s][o] W[o][k] (an fmaf chain in output order) and accumulates gW[o][k] += dY[s][o] feat[s][k];
// threads k < O also sum dY[s][k] (the bias gradient).  One slab per workgroup, added in fixed order by reduce_slabs_kernel.
// dY rows are read through the constant address space (uniform address -> scalar loads: 16 values per row for the whole workgroup).
// H <= 256, O <= 16.
typedef const __attribute__((address_space(4))) float* hb_const_f32p;
__global__ __launch_bounds__(256) void heads_bwd_kernel(const float* __restrict__ dY, const float* __restrict__ feat, const float* __restrict__ Wh,
                                                        int relu_mask, float* __restrict__ dfeat, float* __restrict__ slab, int n, int H, int O) {
    const int k = threadIdx.x;
    const int chunk = (n + gridDim.x - 1) / gridDim.x, r0 = blockIdx.x * chunk, r1 = min(n, r0 + chunk);
    float w[16], gw[16], gb = 0.f;
#pragma unroll
    for (int o = 0; o < 16; ++o) { w[o] = (o < O && k < H) ? Wh[(long long)o * H + k] : 0.f; gw[o] = 0.f; }
    const int kk = k < H ? k : H - 1;
    int s = r0;
    constexpr int RF = 8;                                  // rows in flight (a workgroup's rows are a chain of load round trips otherwise)
    for (; s + RF <= r1; s += RF) {
        float f[RF], d[RF];
#pragma unroll
        for (int q = 0; q < RF; ++q) f[q] = feat[(long long)(s + q) * H + kk];
#pragma unroll
        for (int q = 0; q < RF; ++q) {
            hb_const_f32p dy = (hb_const_f32p)(dY + (long long)(s + q) * O);
            float acc = 0.f;
#pragma unroll
            for (int o = 0; o < 16; ++o) if (o < O) { const float v = dy[o]; acc = fmaf(v, w[o], acc); gw[o] = fmaf(v, f[q], gw[o]); }
            d[q] = (relu_mask && !(f[q] > 0.f)) ? 0.f : acc;
            if (k < O) gb += dY[(long long)(s + q) * O + k];
        }
        if (k < H) {
#pragma unroll
            for (int q = 0; q < RF; ++q) dfeat[(long long)(s + q) * H + k] = d[q];
        }
    }
    for (; s < r1; ++s) {
        const float f = feat[(long long)s * H + kk];
        hb_const_f32p dy = (hb_const_f32p)(dY + (long long)s * O);
        float acc = 0.f;
#pragma unroll
        for (int o = 0; o < 16; ++o) if (o < O) { const float v = dy[o]; acc = fmaf(v, w[o], acc); gw[o] = fmaf(v, f, gw[o]); }
        if (k < H) dfeat[(long long)s * H + k] = (relu_mask && !(f > 0.f)) ? 0.f : acc;
        if (k < O) gb += dY[(long long)s * O + k];
    }
    float* sl = slab + (long long)blockIdx.x * (O * H + O);
    if (k < H) {
#pragma unroll
        for (int o = 0; o < 16; ++o) if (o < O) sl[(long long)o * H + k] = gw[o];
    }
    if (k < O) sl[(long long)O * H + k] = gb;
}
// gW[O][H] += dY^T feat ; gb[O] += colsum(dY) ; dfeat = (dY W) * (feat > 0 if relu_mask).  ws: >= 512 * (O*H + O) floats.
static int heads_bwd_grid(int n) { const int g = (n + 15) / 16; return g > 512 ? 512 : g; }   // 16 rows per workgroup at the training sizes: two steps of 8 rows, two workgroups per CU
// the slab sum of launch_heads_bwd(..., with_reduce = false), on a stream of the caller's choice (nothing but the optimizer needs gW / gb)
void launch_heads_bwd_reduce(const float* ws, float* gW, float* gb, int n, int H, int O, hipStream_t st) {
    if (n <= 0) return;
    launch_reduce_slabs(ws, heads_bwd_grid(n), O * H + O, gW, O * H, gb, O, st);
}
void launch_heads_bwd(const float* dY, const float* feat, const float* Wh, int relu_mask, float* dfeat, float* gW, float* gb, float* ws,
                      int n, int H, int O, hipStream_t st, bool with_reduce, hipEvent_t done_ev) {
    if (n <= 0) return;
    // done_ev: completion of THIS launch as an event (hipExtLaunchKernelGGL: the dispatch packet's own completion signal) -- a separate
    // hipEventRecord behind the kernel is one more packet on the queue and a 7-9 us bubble in front of the next kernel
    if (done_ev) hipExtLaunchKernelGGL(heads_bwd_kernel, dim3(heads_bwd_grid(n)), dim3(256), 0, st, nullptr, done_ev, 0, dY, feat, Wh, relu_mask, dfeat, ws, n, H, O);
    else
    hipLaunchKernelGGL(heads_bwd_kernel, dim3(heads_bwd_grid(n)), dim3(256), 0, st, dY, feat, Wh, relu_mask, dfeat, ws, n, H, O);
    if (with_reduce) launch_heads_bwd_reduce(ws, gW, gb, n, H, O, st);
}

__global__ void gather_rows_kernel(const float* src, const int32_t* idx, long long base, float* dst, int n, int d) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (long long)n * d) return;
    const int s = e / d, c = e % d;
    const long long row = idx ? (long long)idx[s] : base + s;
    dst[e] = src[row * d + c];
}
void launch_gather_rows(const float* src, const int32_t* idx, long long base, float* dst, int n, int d, hipStream_t st) {
    if (n <= 0) return;
    hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)(((long long)n * d + 255) / 256)), dim3(256), 0, st, src, idx, base, dst, n, d);
}

__global__ void fill_kernel(float* p, long long n, float v) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < n) p[e] = v;
}
void launch_fill(float* p, long long n, float v, hipStream_t st) {
    if (n <= 0) return;
    hipLaunchKernelGGL(fill_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, p, n, v);
}

// ------------------------------------------------------------------------------------------ PPO loss
// agents/ppo.py:131-169 + cross_batch_entropy (common/misc_util.py:42-51), forward statistics and
// the analytic gradient wrt (logits, value).  One thread per sample; A <= 16.
#define MAXA 16
__device__ __forceinline__ float philox_uniform(unsigned long long seed, unsigned long long ctr);
struct SampleTerms {
    float lp[MAXA], p[MAXA];
    float H, ratio, adv, surr1, surr2, v, oldv, ret, vclip, vs1, vs2;
    int act;
};
// (All loops over the actions run to the compile-time MAXA under `k < A`: with a run-time bound the per-thread arrays are indexed
// dynamically and live in scratch memory -- 180 B per thread, the loss kernel then took 35 us for 8192 samples.)
__device__ __forceinline__ void log_softmax_twice(const float* z, int A, float* lp, float* p) {
    float mx = z[0];
#pragma unroll
    for (int a = 1; a < MAXA; ++a) if (a < A) mx = fmaxf(mx, z[a]);
    float s = 0.f;
#pragma unroll
    for (int a = 0; a < MAXA; ++a) if (a < A) s += expf(z[a] - mx);
    const float lse = mx + logf(s);
    float s2 = 0.f;
#pragma unroll
    for (int a = 0; a < MAXA; ++a) if (a < A) { lp[a] = z[a] - lse; s2 += expf(lp[a]); }
    const float lse2 = logf(s2);                 // Categorical(logits=log_probs) normalises again (policy.py:86-87)
    float s3 = 0.f;
#pragma unroll
    for (int a = 0; a < MAXA; ++a) if (a < A) { lp[a] -= lse2; p[a] = expf(lp[a]); s3 += p[a]; }
#pragma unroll
    for (int a = 0; a < MAXA; ++a) if (a < A) p[a] /= s3;       // Categorical.probs = softmax(logits)
}
__device__ __forceinline__ void sample_terms(const LossArgs& a, int s, SampleTerms& t) {
    const float* h = a.hout + (long long)s * (a.A + 1);
    float z[MAXA];
#pragma unroll
    for (int k = 0; k < MAXA; ++k) z[k] = (k < a.A) ? h[k] : 0.f;
    log_softmax_twice(z, a.A, t.lp, t.p);
    const int gi = a.idx[s];
    t.act = a.act[gi];
    t.adv = a.adv[gi]; t.ret = a.ret[gi]; t.oldv = a.old_value[gi];
    t.v = h[a.A];
    float H = 0.f, lp_act = 0.f;
#pragma unroll
    for (int k = 0; k < MAXA; ++k) if (k < a.A) { H -= t.p[k] * t.lp[k]; lp_act = (k == t.act) ? t.lp[k] : lp_act; }
    t.H = H;
    t.ratio = expf(lp_act - a.old_logp[gi]);
    t.surr1 = t.ratio * t.adv;
    t.surr2 = fminf(fmaxf(t.ratio, 1.f - a.hp.eps_clip), 1.f + a.hp.eps_clip) * t.adv;
    t.vclip = t.oldv + fminf(fmaxf(t.v - t.oldv, -a.hp.eps_clip), a.hp.eps_clip);
    t.vs1 = (t.v - t.ret) * (t.v - t.ret);
    t.vs2 = (t.vclip - t.ret) * (t.vclip - t.ret);
}

// samples per workgroup of the loss kernels: one wave -- 128 workgroups for an 8192-sample minibatch (the exp / log heavy per-sample
// work ran on 32 CUs with four-wave workgroups: 35 us)
constexpr int LOSS_BLK = 64;
int loss_blocks(int n) { return (n + LOSS_BLK - 1) / LOSS_BLK; }

__device__ __forceinline__ void loss_bwd_sample(const LossArgs& a, int s, const SampleTerms& t) {
    const float ib = a.inv_n_global;
    // d pi_loss / d logp_act  (torch.min routes to surr1 on <=; a tie is the unclipped regime where both
    // branches carry adv/2 each; clamp passes gradient inside [1-eps, 1+eps])
    const float g_lp = -ib * ((t.surr1 <= t.surr2) ? t.adv : 0.f) * t.ratio;
    const float cH = (-a.hp.entropy_coef * a.hp.entropy_mult + a.hp.x_entropy_coef) * ib;
    float plq = 0.f;
    float lq[MAXA];
    const bool xe = a.hp.x_entropy_coef != 0.f;
    if (xe) {
#pragma unroll
        for (int k = 0; k < MAXA; ++k) if (k < a.A) { lq[k] = logf(a.stats[8 + k]); plq += t.p[k] * lq[k]; }
    }
    float* d = a.dY + (long long)s * (a.A + 1);
#pragma unroll
    for (int k = 0; k < MAXA; ++k) if (k < a.A) {
        float gk = g_lp * ((k == t.act ? 1.f : 0.f) - t.p[k]);
        gk += cH * (-t.p[k] * (t.lp[k] + t.H));
        if (xe) gk += a.hp.x_entropy_coef * ib * t.p[k] * (lq[k] - plq);
        d[k] = gk;
    }
    const float dvl = t.v - t.oldv;
    const float inr = (dvl >= -a.hp.eps_clip && dvl <= a.hp.eps_clip) ? 1.f : 0.f;
    float gv;
    if (t.vs1 > t.vs2) gv = 2.f * (t.v - t.ret);
    else if (t.vs2 > t.vs1) gv = 2.f * (t.vclip - t.ret) * inr;
    else gv = (t.v - t.ret) + (t.vclip - t.ret) * inr;
    d[a.A] = a.hp.value_coef * 0.5f * ib * gv;
}
__global__ __launch_bounds__(LOSS_BLK) void loss_bwd_kernel(LossArgs a) {
    const int s = blockIdx.x * LOSS_BLK + threadIdx.x;
    if (s >= a.n) return;
    SampleTerms t;
    sample_terms(a, s, t);
    loss_bwd_sample(a, s, t);
}
// one LOSS_BLK-sample block of the loss: sums over the samples s < s_end of this block -> partial row `row` (per value: wave sums,
// then the waves in order, one barrier per block).  BWD: the loss has no
// batch-level term (x_entropy_coef == 0), so the gradient of the sample goes out in the same pass.
template <bool BWD>
__device__ __forceinline__ void loss_fwd_block(const LossArgs& a, int s, int s_end, int row) {
    constexpr int NW = LOSS_BLK / 64;
    __shared__ float sw[NW][8 + MAXA];
    SampleTerms t;
    float pi = 0.f, vm = 0.f, H = 0.f;
    float pa[MAXA];
#pragma unroll
    for (int k = 0; k < MAXA; ++k) pa[k] = 0.f;
    if (s < s_end) {
        sample_terms(a, s, t);
        pi = fminf(t.surr1, t.surr2);
        vm = fmaxf(t.vs1, t.vs2);
        H = t.H;
#pragma unroll
        for (int k = 0; k < MAXA; ++k) if (k < a.A) pa[k] = t.p[k];
        if (BWD) loss_bwd_sample(a, s, t);
    }
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    float r = wave_sum(pi); if (lane == 0) sw[w][0] = r;
    r = wave_sum(vm);       if (lane == 0) sw[w][1] = r;
    r = wave_sum(H);        if (lane == 0) sw[w][2] = r;
#pragma unroll
    for (int k = 0; k < MAXA; ++k) if (k < a.A) { r = wave_sum(pa[k]); if (lane == 0) sw[w][8 + k] = r; }
    __syncthreads();
    const int k = threadIdx.x;
    if (k < 3 || (k >= 8 && k < 8 + a.A)) {
        float t = sw[0][k];
#pragma unroll
        for (int w2 = 1; w2 < NW; ++w2) t += sw[w2][k];
        a.partial[(long long)row * (8 + a.A) + k] = t;
    }
}
__global__ __launch_bounds__(LOSS_BLK) void loss_fwd_kernel(LossArgs a) {
    loss_fwd_block<false>(a, blockIdx.x * LOSS_BLK + threadIdx.x, a.n, blockIdx.x);
}
// segment k owns ceil(len_k / LOSS_BLK) consecutive blocks (no block straddles two minibatches)
__device__ __forceinline__ void seg_of_block(const SegTab& st, int b, int& k, int& first) {
    first = 0;
    for (k = 0; k < st.n_seg - 1; ++k) {
        const int nb = (st.start[k + 1] - st.start[k] + LOSS_BLK - 1) / LOSS_BLK;
        if (b < first + nb) break;
        first += nb;
    }
}
template <bool BWD>
__global__ __launch_bounds__(LOSS_BLK) void loss_fwd_seg_kernel(LossArgs a, SegTab st) {
    int k, first;
    seg_of_block(st, blockIdx.x, k, first);
    loss_fwd_block<BWD>(a, st.start[k] + (blockIdx.x - first) * LOSS_BLK + threadIdx.x, st.start[k + 1], blockIdx.x);
}
int loss_blocks_seg(const SegTab& st) {
    int nb = 0;
    for (int k = 0; k < st.n_seg; ++k) nb += (st.start[k + 1] - st.start[k] + LOSS_BLK - 1) / LOSS_BLK;
    return nb;
}
void launch_loss_fwd_seg(const LossArgs& a, const SegTab& st, bool with_bwd, hipStream_t stream) {
    const int nb = loss_blocks_seg(st);
    if (nb <= 0) return;
    if (with_bwd) hipLaunchKernelGGL(loss_fwd_seg_kernel<true>, dim3(nb), dim3(LOSS_BLK), 0, stream, a, st);
    else hipLaunchKernelGGL(loss_fwd_seg_kernel<false>, dim3(nb), dim3(LOSS_BLK), 0, stream, a, st);
}
void launch_loss_fwd(const LossArgs& a, hipStream_t st) {
    if (a.n <= 0) return;
    hipLaunchKernelGGL(loss_fwd_kernel, dim3(loss_blocks(a.n)), dim3(LOSS_BLK), 0, st, a);
}

// phase 1: block partials -> this rank's contribution to the GLOBAL-minibatch means (x inv_n_global)
// phase 2: (after the optional cross-rank sum of stats[0..2], stats[8..8+A)) derived terms + log record
__device__ __forceinline__ void loss_finalize_body(const LossArgs& a, int nblk, int phase, const float* fs_ptr, float* log_slot) {
    const int k = threadIdx.x;
    if (phase & 1) {
        // column c of the partial rows, summed by blockDim.x / 32 row groups (rows g, g + G, ...) and then over the groups in order
        // (one thread per column walking all rows serially took 29 us for the 128 rows of an 8192-sample minibatch)
        __shared__ float sp[8][32];
        const int c = k & 31, g = k >> 5, G = blockDim.x >> 5;
        const bool col = c < 3 || (c >= 8 && c < 8 + a.A);
        float s = 0.f;
        if (col && g < 8) for (int b = g; b < nblk; b += G) s += a.partial[(long long)b * (8 + a.A) + c];
        if (g < 8) sp[g][c] = s;
        __syncthreads();
        if (g == 0 && col) {
            float t = sp[0][c];
            for (int j = 1; j < (G < 8 ? G : 8); ++j) t += sp[j][c];
            t *= a.inv_n_global;
            if (c == 0) t = -t;            // pi_loss = -mean(min(surr1,surr2))
            if (c == 1) t = 0.5f * t;      // value_loss = 0.5*mean(max(..))
            a.stats[c] = t;
        }
        __syncthreads();
    }
    if ((phase & 2) && k == 0) {
        float marg = 0.f;
        for (int j = 0; j < a.A; ++j) { const float qj = a.stats[8 + j]; marg -= qj * logf(qj); }
        const float ent = a.stats[2], xent = marg - ent;
        const float fs = fs_ptr ? fs_ptr[0] : 0.f;
        const float total = a.stats[0] + a.hp.value_coef * a.stats[1] - a.hp.entropy_coef * ent * a.hp.entropy_mult
                            - a.hp.x_entropy_coef * xent + a.hp.fs_coef * fs;
        a.stats[3] = xent; a.stats[4] = total; a.stats[5] = fs; a.stats[6] = marg;
        if (log_slot) {
            log_slot[0] = a.stats[0]; log_slot[1] = a.stats[1]; log_slot[2] = ent; log_slot[3] = xent;
            log_slot[4] = total; log_slot[5] = fs; log_slot[6] = marg; log_slot[7] = 0.f;
        }
    }
}
__global__ void loss_finalize_kernel(LossArgs a, int nblk, int phase, const float* fs_ptr, float* log_slot) {
    loss_finalize_body(a, nblk, phase, fs_ptr, log_slot);
}
__global__ void loss_finalize_seg_kernel(LossArgs a, SegTab st, int phase, float* stats_base, const double* fs_parts, int fs_d, float* fs_out,
                                         float* log_base) {
    const int k = blockIdx.x;
    int first = 0;
    for (int j = 0; j < k; ++j) first += (st.start[j + 1] - st.start[j] + LOSS_BLK - 1) / LOSS_BLK;
    a.partial += (long long)first * (8 + a.A);
    a.stats = stats_base + 32 * k;
    if (fs_parts && threadIdx.x == 0) {                 // second half of the feature-sparsity metric: mean over the d columns
        double tot = 0.0;
        for (int j = 0; j < FS_PARTS; ++j) tot += fs_parts[k * FS_PARTS + j];
        fs_out[k] = (float)(tot / fs_d);
    }
    loss_finalize_body(a, (st.start[k + 1] - st.start[k] + LOSS_BLK - 1) / LOSS_BLK, phase, fs_parts ? fs_out + k : nullptr, log_base ? log_base + 8 * k : nullptr);
}
// phase 2 of n_rec records at once (after the cross-rank sum of the statistics ring): record k from stats_base + 32 k, fs_base[k]
__global__ void loss_finalize_records_kernel(LossArgs a, float* stats_base, const float* fs_base, float* log_base) {
    const int k = blockIdx.x;
    a.stats = stats_base + 32 * k;
    loss_finalize_body(a, 0, 2, fs_base ? fs_base + k : nullptr, log_base + 8 * k);
}
void launch_loss_finalize_records(const LossArgs& a, int n_rec, float* stats_base, const float* fs_base, float* log_base, hipStream_t stream) {
    if (n_rec <= 0) return;
    hipLaunchKernelGGL(loss_finalize_records_kernel, dim3(n_rec), dim3(64), 0, stream, a, stats_base, fs_base, log_base);
}
void launch_loss_finalize_seg(const LossArgs& a, const SegTab& st, int phase, float* stats_base, const double* fs_parts, int fs_d, float* fs_out,
                              float* log_base, hipStream_t stream) {
    hipLaunchKernelGGL(loss_finalize_seg_kernel, dim3(st.n_seg), dim3(256), 0, stream, a, st, phase, stats_base, fs_parts, fs_d, fs_out, log_base);
}
void launch_loss_finalize(const LossArgs& a, int nblk, int phase, const float* fs_ptr, float* log_slot, hipStream_t st) {
    hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(256), 0, st, a, nblk, phase, fs_ptr, log_slot);
}

void launch_loss_bwd(const LossArgs& a, hipStream_t st) {
    if (a.n <= 0) return;
    hipLaunchKernelGGL(loss_bwd_kernel, dim3(loss_blocks(a.n)), dim3(LOSS_BLK), 0, st, a);
}

// feature-sparsity metric (common/model.py:207): mean_j max_b tanh(|100*relu(h_bj)|)
// = mean_j tanh(100 * max_b relu(h_bj)) (tanh monotone).  flat_pre is block3's output BEFORE the ReLU.
// FS_GROUPS row groups x d columns of partial maxima; a thread owns 8 consecutive columns (16-byte bf16 / 2 x 16-byte
// fp32 loads), a workgroup walks whole rows.
constexpr int FS_GROUPS = 128;
__global__ __launch_bounds__(256) void colmax_partial_kernel(const void* x, int bf16, int n, int d, float* part) {
    for (int j = threadIdx.x * 8; j < d; j += 256 * 8) {
        float m[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int b = blockIdx.x; b < n; b += FS_GROUPS) {
            const long long o = (long long)b * d + j;
            if (bf16) {
                const uint4 u = *(const uint4*)((const unsigned short*)x + o);
                const unsigned w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
                for (int k = 0; k < 8; ++k) m[k] = fmaxf(m[k], (k & 1) ? __uint_as_float(w[k >> 1] & 0xffff0000u) : __uint_as_float(w[k >> 1] << 16));
            } else {
                const f32x4 lo = *(const f32x4*)((const float*)x + o), hi = *(const f32x4*)((const float*)x + o + 4);
                m[0] = fmaxf(m[0], lo.x); m[1] = fmaxf(m[1], lo.y); m[2] = fmaxf(m[2], lo.z); m[3] = fmaxf(m[3], lo.w);
                m[4] = fmaxf(m[4], hi.x); m[5] = fmaxf(m[5], hi.y); m[6] = fmaxf(m[6], hi.z); m[7] = fmaxf(m[7], hi.w);
            }
        }
        float* p = part + (long long)blockIdx.x * d + j;
        *(f32x4*)p = (f32x4){m[0], m[1], m[2], m[3]};
        *(f32x4*)(p + 4) = (f32x4){m[4], m[5], m[6], m[7]};
    }
}
__global__ __launch_bounds__(1024) void fs_finalize_kernel(const float* part, int groups, int d, float* fs_out) {
    __shared__ double sb[16];
    double s = 0.0;
    for (int j = threadIdx.x; j < d; j += 1024) {
        float m = 0.f;
#pragma unroll 16
        for (int g = 0; g < groups; ++g) m = fmaxf(m, part[(long long)g * d + j]);
        s += (double)tanhf(fabsf(m * 100.f));
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) sb[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) { double tot = 0.0; for (int k = 0; k < 16; ++k) tot += sb[k]; fs_out[0] = (float)(tot / d); }
}
// segment-aware two-stage version (mi_minibatch_multi and the single-minibatch path of modes 0 / 2): G row groups per segment with
// G x n_seg ~ 512 workgroups (all CUs busy, 4 rows in flight per thread), then FS_PARTS column blocks per segment whose fp64
// partial sums the loss finalisation adds up.
__device__ __forceinline__ void fs_row_max(const void* x, int bf16, long long o, float (&m)[8]) {
    if (bf16) {
        const uint4 u = *(const uint4*)((const unsigned short*)x + o);
        const unsigned w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
        for (int q = 0; q < 8; ++q) m[q] = fmaxf(m[q], (q & 1) ? __uint_as_float(w[q >> 1] & 0xffff0000u) : __uint_as_float(w[q >> 1] << 16));
    } else {
        const f32x4 lo = *(const f32x4*)((const float*)x + o), hi = *(const f32x4*)((const float*)x + o + 4);
        m[0] = fmaxf(m[0], lo.x); m[1] = fmaxf(m[1], lo.y); m[2] = fmaxf(m[2], lo.z); m[3] = fmaxf(m[3], lo.w);
        m[4] = fmaxf(m[4], hi.x); m[5] = fmaxf(m[5], hi.y); m[6] = fmaxf(m[6], hi.z); m[7] = fmaxf(m[7], hi.w);
    }
}
template <bool BF16>
__global__ __launch_bounds__(256) void colmax_partial_seg_kernel(const void* x, SegTab st, int d, int G, float* part) {
    const int k = blockIdx.y, r0 = st.start[k], r1 = st.start[k + 1];
    for (int j = threadIdx.x * 8; j < d; j += 256 * 8) {
        float m[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        int b = r0 + blockIdx.x;
        // 8 rows in flight (the storage type is a template parameter: no branch between the loads): a workgroup's 32 rows are a chain
        // of load round trips otherwise (14.8 us for 32 MB with 4 in flight)
        for (; b + 7 * G < r1; b += 8 * G) {
            if constexpr (BF16) {
                uint4 u[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) u[q] = *(const uint4*)((const unsigned short*)x + (long long)(b + q * G) * d + j);
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const unsigned w[4] = {u[q].x, u[q].y, u[q].z, u[q].w};
#pragma unroll
                    for (int e = 0; e < 8; ++e) m[e] = fmaxf(m[e], (e & 1) ? __uint_as_float(w[e >> 1] & 0xffff0000u) : __uint_as_float(w[e >> 1] << 16));
                }
            } else {
#pragma unroll
                for (int q = 0; q < 8; ++q) fs_row_max(x, 0, (long long)(b + q * G) * d + j, m);
            }
        }
        for (; b < r1; b += G) fs_row_max(x, BF16 ? 1 : 0, (long long)b * d + j, m);
        float* p = part + ((long long)k * G + blockIdx.x) * d + j;
        *(f32x4*)p = (f32x4){m[0], m[1], m[2], m[3]};
        *(f32x4*)(p + 4) = (f32x4){m[4], m[5], m[6], m[7]};
    }
}
__global__ __launch_bounds__(256) void fs_parts_seg_kernel(const float* part, int G, int d, double* fs_parts) {
    __shared__ double sb[4];
    const int k = blockIdx.y, cols = d / FS_PARTS;
    part += (long long)k * G * d;
    double s = 0.0;
    for (int j = blockIdx.x * cols + threadIdx.x; j < (blockIdx.x + 1) * cols; j += 256) {
        float m = 0.f;
#pragma unroll 16
        for (int g = 0; g < G; ++g) m = fmaxf(m, part[(long long)g * d + j]);
        s += (double)tanhf(fabsf(m * 100.f));
    }
    const double t = block_sum256(s, sb);
    if (threadIdx.x == 0) fs_parts[k * FS_PARTS + blockIdx.x] = t;
}
#ifndef FS_TOTAL_GROUPS
#define FS_TOTAL_GROUPS 256      // row groups over all segments: stage 1 (column maxima) gets faster, stage 2 (max over groups) slower with more
#endif
int fs_groups_per_segment(int n_seg) { int g = FS_TOTAL_GROUPS / (n_seg < 1 ? 1 : n_seg); int p = 32; while (p * 2 <= g) p *= 2; return p; }
void launch_fs_metric_seg(const void* flat_pre, int bf16, const SegTab& st, int d, float* colmax_scratch, double* fs_parts, hipStream_t stream) {
    if (st.n_seg <= 0) return;
    if (d % (8 * FS_PARTS)) { mi_launch_fail("feature-sparsity metric: the feature count must be a multiple of 64"); return; }
    const int G = fs_groups_per_segment(st.n_seg);
    if (bf16) hipLaunchKernelGGL(colmax_partial_seg_kernel<true>, dim3(G, st.n_seg), dim3(256), 0, stream, flat_pre, st, d, G, colmax_scratch);
    else hipLaunchKernelGGL(colmax_partial_seg_kernel<false>, dim3(G, st.n_seg), dim3(256), 0, stream, flat_pre, st, d, G, colmax_scratch);
    hipLaunchKernelGGL(fs_parts_seg_kernel, dim3(FS_PARTS, st.n_seg), dim3(256), 0, stream, (const float*)colmax_scratch, G, d, fs_parts);
}
int fs_metric_groups() { return FS_GROUPS; }                    // row groups launch_fs_metric leaves in its scratch
void launch_fs_metric(const void* flat_pre, int bf16, int n, int d, float* colmax_scratch, float* fs_out, hipStream_t st) {
    if (n <= 0) return;
    if (d % 8) { mi_launch_fail("feature-sparsity metric: the feature count must be a multiple of 8"); return; }      // the flattened IMPALA feature map (2048)
    hipLaunchKernelGGL(colmax_partial_kernel, dim3(FS_GROUPS), dim3(256), 0, st, flat_pre, bf16, n, d, colmax_scratch);
    hipLaunchKernelGGL(fs_finalize_kernel, dim3(1), dim3(1024), 0, st, (const float*)colmax_scratch, FS_GROUPS, d, fs_out);
}

// Gradient of the feature-sparsity term fs_coef * mean_j max_b tanh(|100 h_bj|) (agents/ppo.py:148-169, common/model.py:207) with
// respect to the pre-fc features: for every column j the term depends on ONE row, the first one (in minibatch order, torch.max's
// choice) that attains the column maximum m_j; d/dh = fs_coef * 100 * (1 - tanh(100 m_j)^2) / d there (h > 0: |.| is the identity; a
// column whose maximum is 0 has gradient 0, abs'(0) = 0).  part: the G x d partial column maxima the metric kernels left behind.
__global__ __launch_bounds__(256) void fs_colmax_final_kernel(const float* part, int G, int d, float* colmax, int* arg) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= d) return;
    float m = 0.f;
    for (int g = 0; g < G; ++g) m = fmaxf(m, part[(long long)g * d + j]);
    colmax[j] = m; arg[j] = 0x7fffffff;
}
__global__ __launch_bounds__(256) void fs_argfirst_kernel(const void* x, int bf16, int n, int d, const float* colmax, int* arg) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= d) return;
    const float m = colmax[j];
    if (!(m > 0.f)) return;
    int first = 0x7fffffff;
    for (int b = blockIdx.y; b < n; b += gridDim.y) {
        const long long o = (long long)b * d + j;
        const float v = bf16 ? __uint_as_float(((unsigned)((const unsigned short*)x)[o]) << 16) : ((const float*)x)[o];
        if (v == m && b < first) first = b;
    }
    if (first != 0x7fffffff) atomicMin(arg + j, first);
}
__global__ __launch_bounds__(256) void fs_apply_kernel(void* G, int bf16, int d, const float* colmax, const int* arg, float scale) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= d) return;
    const int b = arg[j];
    if (b == 0x7fffffff) return;
    const float t = tanhf(fabsf(colmax[j] * 100.f)), g = scale * (1.f - t * t);
    const long long o = (long long)b * d + j;
    if (bf16) {
        unsigned short* p = (unsigned short*)G + o;
        *p = f2bf_g(__uint_as_float(((unsigned)*p) << 16) + g);
    } else ((float*)G)[o] += g;
}
// Multi-rank: the column maxima are those of the GLOBAL minibatch.  Every rank packs its own candidate per column
// into one 64-bit key -- (bits of the local maximum) << 32 | (0xffffffff - global position of its first row attaining it) -- so that ONE
// max-all-reduce yields the global maximum (non-negative floats order like their bit patterns) and, among equal maxima, the row that
// comes first in the global minibatch (torch.max's choice in the single-process reference).  Key 0 = no positive value in the column.
__global__ __launch_bounds__(256) void fs_keys_kernel(const float* colmax, const int* arg, const int32_t* gpos, int d, long long* keys, long long* keys_local) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= d) return;
    const int b = arg[j];
    long long k = 0;
    if (b != 0x7fffffff && colmax[j] > 0.f)
        k = ((long long)__float_as_uint(colmax[j]) << 32) | (long long)(0xffffffffu - (unsigned)gpos[b]);
    keys[j] = k; keys_local[j] = k;
}
// after the all-reduce: the winner of column j adds the gradient at its row; every rank takes the global maximum for the metric
__global__ __launch_bounds__(256) void fs_apply_keys_kernel(void* G, int bf16, int d, const long long* keys, const long long* keys_local, const int* arg, float scale) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= d) return;
    const long long k = keys[j];
    if (k == 0 || k != keys_local[j]) return;
    const float m = __uint_as_float((unsigned)((unsigned long long)k >> 32));
    const float t = tanhf(fabsf(m * 100.f)), g = scale * (1.f - t * t);
    const long long o = (long long)arg[j] * d + j;
    if (bf16) {
        unsigned short* p = (unsigned short*)G + o;
        *p = f2bf_g(__uint_as_float(((unsigned)*p) << 16) + g);
    } else ((float*)G)[o] += g;
}
__global__ __launch_bounds__(1024) void fs_from_keys_kernel(const long long* keys, int d, float* fs_out) {      // mean_j tanh(|100 m_j|) of the global maxima
    __shared__ double sb[16];
    double s = 0.0;
    for (int j = threadIdx.x; j < d; j += 1024) s += (double)tanhf(fabsf(__uint_as_float((unsigned)((unsigned long long)keys[j] >> 32)) * 100.f));
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) sb[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) { double tot = 0.0; for (int k = 0; k < 16; ++k) tot += sb[k]; fs_out[0] = (float)(tot / d); }
}
// part: [G][d] partial column maxima of this rank's rows; gpos: global minibatch position of every local row (ascending)
void launch_fs_keys(const void* x, int bf16, int n, int d, const float* part, int G, const int32_t* gpos, float* colmax, int* arg,
                    long long* keys, long long* keys_local, hipStream_t st) {
    hipLaunchKernelGGL(fs_colmax_final_kernel, dim3((d + 255) / 256), dim3(256), 0, st, part, G, d, colmax, arg);
    if (n > 0) hipLaunchKernelGGL(fs_argfirst_kernel, dim3((d + 255) / 256, n < 64 ? n : 64), dim3(256), 0, st, x, bf16, n, d, (const float*)colmax, arg);
    hipLaunchKernelGGL(fs_keys_kernel, dim3((d + 255) / 256), dim3(256), 0, st, (const float*)colmax, (const int*)arg, gpos, d, keys, keys_local);
}
void launch_fs_apply_keys(void* Gd, int bf16, int d, const long long* keys, const long long* keys_local, const int* arg, float fs_coef, hipStream_t st) {
    hipLaunchKernelGGL(fs_apply_keys_kernel, dim3((d + 255) / 256), dim3(256), 0, st, Gd, bf16, d, keys, keys_local, arg, fs_coef * 100.f / (float)d);
}
void launch_fs_from_keys(const long long* keys, int d, float* fs_out, hipStream_t st) {
    hipLaunchKernelGGL(fs_from_keys_kernel, dim3(1), dim3(1024), 0, st, keys, d, fs_out);
}
// x: block3's output before the ReLU [n][d] (fp32 / bf16); part: [G][d] partial column maxima of relu(x); Gd: d loss / d x [n][d], updated in place
void launch_fs_grad(const void* x, int bf16, int n, int d, const float* part, int G, void* Gd, float fs_coef, float* colmax, int* arg, hipStream_t st) {
    if (n <= 0) return;
    hipLaunchKernelGGL(fs_colmax_final_kernel, dim3((d + 255) / 256), dim3(256), 0, st, part, G, d, colmax, arg);
    hipLaunchKernelGGL(fs_argfirst_kernel, dim3((d + 255) / 256, n < 64 ? n : 64), dim3(256), 0, st, x, bf16, n, d, (const float*)colmax, arg);
    hipLaunchKernelGGL(fs_apply_kernel, dim3((d + 255) / 256), dim3(256), 0, st, Gd, bf16, d, (const float*)colmax, (const int*)arg, fs_coef * 100.f / (float)d);
}

// ------------------------------------------------------------------------------------------ GAE scan
// Storage.compute_estimates (common/storage.py:56-77).  One thread per env (coalesced over E),
// serial over T; fp contraction off so every operation rounds exactly like the reference's
// separate fp32 tensor ops (bit-exact advantages/returns).
__global__ void gae_kernel(const float* rew, const float* done, const float* value, float* adv, float* ret, int T, int E,
                           float gamma, float gl, int use_gae) {
#pragma clang fp contract(off)
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    float A = 0.f;
    for (int i = T - 1; i >= 0; --i) {
        const long long o = (long long)i * E + e;
        float a_i = 0.f;
        if (use_gae) {
            const float nd = 1.f - done[o];
            const float delta = (rew[o] + (gamma * value[o + E]) * nd) - value[o];
            A = ((gl * A) * nd) + delta;
            a_i = A;
        }
        adv[o] = a_i;
        ret[o] = a_i + value[o];           // use_gae=False: return = adv(=0) + V (the reference's overwrite, storage.py:77)
    }
}
void launch_gae(const float* rew, const float* done, const float* value, float* adv, float* ret, int T, int E,
                float gamma, float lmbda, int use_gae, hipStream_t st) {
    const float gl = (float)((double)gamma * (double)lmbda);
    hipLaunchKernelGGL(gae_kernel, dim3((E + 63) / 64), dim3(64), 0, st, rew, done, value, adv, ret, T, E, gamma, gl, use_gae);
}

// advantage normalisation (common/storage.py:78-79): (A - mean) / (std_unbiased + 1e-8).
// stats3 = {count, mean, M2} in fp64 so that ranks can merge them (Chan) before apply.
__global__ __launch_bounds__(1024) void advnorm_stats_kernel(const float* adv, int n, double* stats3) {
    __shared__ double sb[16];
    __shared__ double s_mean;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    double s = 0.0;
    for (int k = tid; k < n; k += 1024) s += (double)adv[k];
    s = wave_sum(s);
    if (lane == 0) sb[w] = s;
    __syncthreads();
    if (tid == 0) { double t = 0.0; for (int k = 0; k < 16; ++k) t += sb[k]; s_mean = n > 0 ? t / n : 0.0; }
    __syncthreads();
    const double mean = s_mean;
    double m2 = 0.0;
    for (int k = tid; k < n; k += 1024) { const double d = (double)adv[k] - mean; m2 += d * d; }
    m2 = wave_sum(m2);
    __syncthreads();
    if (lane == 0) sb[w] = m2;
    __syncthreads();
    if (tid == 0) { double t = 0.0; for (int k = 0; k < 16; ++k) t += sb[k]; stats3[0] = (double)n; stats3[1] = mean; stats3[2] = t; }
}
__global__ void advnorm_apply_kernel(float* adv, int n, const double* stats3) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const float mean = (float)stats3[1];
    const float sd = (float)sqrt(stats3[2] / (stats3[0] - 1.0));
    adv[k] = (adv[k] - mean) / (sd + 1e-8f);
}
// Chan et al. merge of R ranks' {count, mean, M2} triples (fp64) into stats3 -- the device side of mi355/dist.py::merge_adv_stats
__global__ void advnorm_merge_kernel(const double* all, int R, double* stats3) {
    if (threadIdx.x || blockIdx.x) return;
    double n = 0.0, mean = 0.0, m2 = 0.0;
    for (int r = 0; r < R; ++r) {
        const double c = all[3 * r], mu = all[3 * r + 1], s = all[3 * r + 2];
        if (c == 0.0) continue;
        const double tot = n + c, d = mu - mean;
        mean = mean + d * c / tot;
        m2 = m2 + s + d * d * n * c / tot;
        n = tot;
    }
    stats3[0] = n; stats3[1] = mean; stats3[2] = m2;
}
void launch_advnorm_merge(const double* all, int R, double* stats3, hipStream_t st) {
    hipLaunchKernelGGL(advnorm_merge_kernel, dim3(1), dim3(64), 0, st, all, R, stats3);
}
void launch_advnorm_stats(const float* adv, int n, double* stats3, hipStream_t st) {
    hipLaunchKernelGGL(advnorm_stats_kernel, dim3(1), dim3(1024), 0, st, adv, n, stats3);
}
void launch_advnorm_apply(float* adv, int n, const double* stats3, hipStream_t st) {
    if (n <= 0) return;
    hipLaunchKernelGGL(advnorm_apply_kernel, dim3((n + 255) / 256), dim3(256), 0, st, adv, n, stats3);
}

// ------------------------------------------------------------------------------------------ rollout head: sample
// agents/ppo.py:77-79: dist.sample(), dist.log_prob(act).  Inverse-CDF over the A probabilities with a
// uniform from a caller-supplied array (tests) or Philox4x32-10 keyed by (seed, counter + env).
__global__ void sample_kernel(const float* hout, int n, int A, const float* u, unsigned long long seed,
                              unsigned long long ctr, int32_t* act, float* logp, float* value) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    const float* h = hout + (long long)e * (A + 1);
    float z[MAXA], lp[MAXA], p[MAXA];
    for (int k = 0; k < A; ++k) z[k] = h[k];
    log_softmax_twice(z, A, lp, p);
    const float uu = u ? u[e] : philox_uniform(seed, ctr + e);
    float cdf = 0.f;
    int a_sel = 0;
    for (int k = 0; k < A; ++k) { cdf += expf(lp[k]); if (cdf <= uu) a_sel = k + 1; }
    if (a_sel > A - 1) a_sel = A - 1;
    if (act) act[e] = a_sel;
    if (logp) logp[e] = lp[a_sel];
    if (value) value[e] = h[A];
}
__global__ void logp_all_kernel(const float* hout, int n, int A, float* lp_out, float* value_out) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    const float* h = hout + (long long)e * (A + 1);
    float z[MAXA], lp[MAXA], p[MAXA];
    for (int k = 0; k < A; ++k) z[k] = h[k];
    log_softmax_twice(z, A, lp, p);
    if (lp_out) for (int k = 0; k < A; ++k) lp_out[(long long)e * A + k] = lp[k];
    if (value_out) value_out[e] = h[A];
}
void launch_logp_all(const float* hout, int n, int A, float* lp_out, float* value_out, hipStream_t st) {
    if (n <= 0) return;
    hipLaunchKernelGGL(logp_all_kernel, dim3((n + 63) / 64), dim3(64), 0, st, hout, n, A, lp_out, value_out);
}
__device__ __forceinline__ void philox_round(uint32_t* c, uint32_t* k) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
    const uint32_t hi0 = __umulhi(M0, c[0]), lo0 = M0 * c[0];
    const uint32_t hi1 = __umulhi(M1, c[2]), lo1 = M1 * c[2];
    const uint32_t n0 = hi1 ^ c[1] ^ k[0], n2 = hi0 ^ c[3] ^ k[1];
    c[0] = n0; c[1] = lo1; c[2] = n2; c[3] = lo0;
    k[0] += 0x9E3779B9u; k[1] += 0xBB67AE85u;
}
// Philox4x32-10 (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3", SC'11): ten rounds on a 128-bit counter under a
// 64-bit key, the key bumped by the Weyl constants between rounds (the bump after the last round is unused).  Known-answer vectors
// of the Random123 distribution are checked through mi_debug_philox (tests/test_gpu_engine.py) and in oracle/philox.py.
__device__ __forceinline__ void philox4x32_10(uint32_t* c, uint32_t* k) {
#pragma unroll
    for (int r = 0; r < 10; ++r) philox_round(c, k);
}
__device__ __forceinline__ float philox_uniform(unsigned long long seed, unsigned long long ctr) {
    uint32_t c[4] = {(uint32_t)ctr, (uint32_t)(ctr >> 32), 0u, 0u};
    uint32_t k[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    philox4x32_10(c, k);
    return (float)(c[0] >> 8) * (1.0f / 16777216.0f);     // [0,1), 24 bits
}
// test hook: in6[i] = {c0,c1,c2,c3,k0,k1} -> out4[i] = the four output words; u_out[i] = the sampler's uniform for
// (seed = k0 | k1 << 32, counter = c0 | c1 << 32), i.e. exactly what sample_kernel / heads_sample_kernel draw
__global__ void philox_debug_kernel(const uint32_t* in6, int n, uint32_t* out4, float* u_out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t c[4] = {in6[6 * i], in6[6 * i + 1], in6[6 * i + 2], in6[6 * i + 3]};
    uint32_t k[2] = {in6[6 * i + 4], in6[6 * i + 5]};
    philox4x32_10(c, k);
    for (int j = 0; j < 4; ++j) out4[4 * i + j] = c[j];
    u_out[i] = philox_uniform((unsigned long long)in6[6 * i + 4] | ((unsigned long long)in6[6 * i + 5] << 32),
                              (unsigned long long)in6[6 * i] | ((unsigned long long)in6[6 * i + 1] << 32));
}
void launch_philox_debug(const uint32_t* in6, int n, uint32_t* out4, float* u_out, hipStream_t st) {
    if (n <= 0) return;
    hipLaunchKernelGGL(philox_debug_kernel, dim3((n + 255) / 256), dim3(256), 0, st, in6, n, out4, u_out);
}
// Rollout head, fused: policy/value heads (common/policy.py:74-80) + log-softmax + sample + log_prob
// (agents/ppo.py:77-79); results also packed [n][3] = {act, logp, value} for ONE read-back.
// 16 envs per workgroup: feature rows AND the (A+1) x H head matrix go through LDS, thread (env, output) owns one dot
// product and then one ACTION of the normalisation: the exp / log terms are computed one per lane, every sum is still
// taken in action order 0..A-1 (each lane re-adds the terms from LDS), so the numbers are those of the sequential
// log_softmax_twice + CDF walk of sample_kernel.
__global__ __launch_bounds__(256) void heads_sample_kernel(const float* __restrict__ feat, const float* __restrict__ Wh,
                                                           const float* __restrict__ bh, int n, int H, int A, const float* u,
                                                           unsigned long long seed, unsigned long long ctr, int32_t* act,
                                                           float* logp, float* value, float* pack, float* hout,
                                                           const float* rd, float* rew_dst, float* done_dst,
                                                           unsigned* done_ctr, unsigned* host_flag, unsigned ticket) {
    __shared__ __attribute__((aligned(16))) float s_f[16 * 260];
    __shared__ __attribute__((aligned(16))) float s_w[17 * 260];
    __shared__ float s_z[16 * 17];
    const int tid = threadIdx.x, e0 = blockIdx.x * 16;
    const int el = tid >> 4, o = tid & 15, e = e0 + el;
    float rwd = 0.f, dn = 0.f;
    if (rd && o == 0 && e < n) { rwd = rd[e]; dn = rd[n + e]; }          // (host-visible staging) issued first: latency hidden by the dots
    if (H == 256) {
        // the IMPALA width: all 9 loads of a thread are issued before the first LDS store (the general loop below divides by a
        // run-time H and waits for every load before the next: 9 memory latencies in a row, half of this kernel's time)
        f32x4 rf[4], rw[5];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k = tid + j * 256, r = k >> 6, kk = (k & 63) * 4;
            rf[j] = (e0 + r < n) ? *(const f32x4*)(feat + (long long)(e0 + r) * 256 + kk) : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const int k = tid + j * 256, r = k >> 6, kk = (k & 63) * 4;
            rw[j] = (r <= A) ? *(const f32x4*)(Wh + (long long)r * 256 + kk) : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) { const int k = tid + j * 256; *(f32x4*)(s_f + (k >> 6) * 260 + (k & 63) * 4) = rf[j]; }
#pragma unroll
        for (int j = 0; j < 5; ++j) { const int k = tid + j * 256; if ((k >> 6) <= A) *(f32x4*)(s_w + (k >> 6) * 260 + (k & 63) * 4) = rw[j]; }
    } else if ((H & 3) == 0) {
        const int H4 = H >> 2;
        for (int k = tid; k < 16 * H4; k += 256) {
            const int r = k / H4, kk = (k % H4) * 4;
            *(f32x4*)(s_f + r * 260 + kk) = (e0 + r < n) ? *(const f32x4*)(feat + (long long)(e0 + r) * H + kk) : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        for (int k = tid; k < (A + 1) * H4; k += 256) {
            const int r = k / H4, kk = (k % H4) * 4;
            *(f32x4*)(s_w + r * 260 + kk) = *(const f32x4*)(Wh + (long long)r * H + kk);
        }
    } else {
        for (int k = tid; k < 16 * H; k += 256) { const int r = k / H, kk = k % H; s_f[r * 260 + kk] = (e0 + r < n) ? feat[(long long)(e0 + r) * H + kk] : 0.f; }
        for (int k = tid; k < (A + 1) * H; k += 256) { const int r = k / H, kk = k % H; s_w[r * 260 + kk] = Wh[(long long)r * H + kk]; }
    }
    __syncthreads();
    for (int oo = o; oo <= A; oo += 16) {                 // A+1 <= 17 outputs: output 16 (if any) is taken by o == 0
        const float* w = s_w + oo * 260;
        const float* f = s_f + el * 260;
        float acc = 0.f;
        int k = 0;
        if (H == 256) {                                   // same order of operations, LDS reads of 8 steps in flight
#pragma unroll 8
            for (; k < 256; k += 4) {
                const f32x4 fv = *(const f32x4*)(f + k), ww = *(const f32x4*)(w + k);
                acc += fv.x * ww.x + fv.y * ww.y + fv.z * ww.z + fv.w * ww.w;
            }
        }
        for (; k + 4 <= H; k += 4) {
            const f32x4 fv = *(const f32x4*)(f + k), ww = *(const f32x4*)(w + k);
            acc += fv.x * ww.x + fv.y * ww.y + fv.z * ww.z + fv.w * ww.w;
        }
        for (; k < H; ++k) acc += f[k] * w[k];
        s_z[el * 17 + oo] = acc + bh[oo];
    }
    __syncthreads();
    // The 16 lanes of an env sit in one wave (lane = 16*(el & 3) + o).  A lane publishes its term in the env's 64-byte LDS row and
    // reads the whole row back with four 16-byte reads (same wave: LDS executes a wave's operations in order, no barrier), then adds
    // the A terms in action order from registers.  (A run-time loop of 15 ds_bpermute + add, each waiting for the previous, was ~40 %
    // of this kernel: six such loops.)
    __shared__ __attribute__((aligned(16))) float s_t[16 * 16];
    const bool lane_on = o < A;
    auto row = [&](float term, float (&v)[16]) {
        __builtin_amdgcn_wave_barrier();
        s_t[el * 16 + o] = term;
        __builtin_amdgcn_wave_barrier();
        const f32x4 a0 = *(const f32x4*)(s_t + el * 16), a1 = *(const f32x4*)(s_t + el * 16 + 4);
        const f32x4 a2 = *(const f32x4*)(s_t + el * 16 + 8), a3 = *(const f32x4*)(s_t + el * 16 + 12);
        v[0] = a0.x; v[1] = a0.y; v[2] = a0.z; v[3] = a0.w; v[4] = a1.x; v[5] = a1.y; v[6] = a1.z; v[7] = a1.w;
        v[8] = a2.x; v[9] = a2.y; v[10] = a2.z; v[11] = a2.w; v[12] = a3.x; v[13] = a3.y; v[14] = a3.z; v[15] = a3.w;
    };
    auto group_sum = [&](float term) {                    // sum over the env's A lanes, in action order
        float v[16];
        row(term, v);
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) if (k < A) t += v[k];
        return t;
    };
    const float z = lane_on ? s_z[el * 17 + o] : 0.f;
    float mx;
    {
        float v[16];
        row(z, v);
        mx = v[0];
#pragma unroll
        for (int k = 1; k < 16; ++k) if (k < A) mx = fmaxf(mx, v[k]);
    }
    const float s1 = group_sum(lane_on ? expf(z - mx) : 0.f);
    float lp = z - (mx + logf(s1));
    const float s2 = group_sum(lane_on ? expf(lp) : 0.f);
    lp -= logf(s2);                                        // Categorical(logits=log_probs) normalises again (policy.py:86-87)
    const float pr = lane_on ? expf(lp) : 0.f;
    float cdf = 0.f;
    {
        float v[16];
        row(pr, v);
#pragma unroll
        for (int k = 0; k < 16; ++k) if (k < A) cdf += (k <= o) ? v[k] : 0.f;      // prefix in action order
    }
    const float uu = (e < n) ? (u ? u[e] : philox_uniform(seed, ctr + e)) : 0.f;
    const float mark = (lane_on && cdf <= uu) ? (float)(o + 1) : 0.f;
    float sel = 0.f;
    {
        float v[16];
        row(mark, v);
#pragma unroll
        for (int k = 0; k < 16; ++k) if (k < A) sel = fmaxf(sel, v[k]);
    }
    int a_sel = (int)sel;
    if (a_sel > A - 1) a_sel = A - 1;
    float lp_pick;
    {
        float v[16];
        row(lp, v);
        lp_pick = v[0];
#pragma unroll
        for (int k = 1; k < 16; ++k) lp_pick = (k == a_sel) ? v[k] : lp_pick;
    }
    if (o == 0 && e < n) {
        const float lp_sel = lp_pick, val = s_z[el * 17 + A];
        if (rd) { rew_dst[e] = rwd; done_dst[e] = dn; }        // previous step's reward / done into the (T,E) arrays
        if (act) act[e] = a_sel;
        if (logp) logp[e] = lp_sel;
        if (value) value[e] = val;
        if (pack) { pack[e * 3] = (float)a_sel; pack[e * 3 + 1] = lp_sel; pack[e * 3 + 2] = val; }
        if (hout) for (int k = 0; k <= A; ++k) hout[(long long)e * (A + 1) + k] = s_z[el * 17 + k];
    }
    if (host_flag) {
        // completion ticket in host-visible memory, written by the LAST workgroup after every workgroup's results are visible system-
        // wide: the host spins on it instead of waiting for the stream (hipStreamSynchronize returns ~5 us later: scratch/synclat.hip)
        __threadfence_system();
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned old = atomicAdd(done_ctr, 1u);
            if (old == gridDim.x - 1) {
                *done_ctr = 0;
                __threadfence_system();
                __hip_atomic_store(host_flag, ticket, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    }
}
void launch_heads_sample(const float* feat, const float* Wh, const float* bh, int n, int H, int A, const float* u,
                         unsigned long long seed, unsigned long long ctr, int32_t* act, float* logp, float* value, float* pack,
                         float* hout, const float* rd, float* rew_dst, float* done_dst, hipStream_t st,
                         unsigned* done_ctr, unsigned* host_flag, unsigned ticket) {
    if (n <= 0) return;
    hipLaunchKernelGGL(heads_sample_kernel, dim3((n + 15) / 16), dim3(256), 0, st, feat, Wh, bh, n, H, A, u, seed, ctr, act, logp, value, pack, hout,
                       rd, rew_dst, done_dst, done_ctr, host_flag, ticket);
}
void launch_sample(const float* hout, int n, int A, const float* u, unsigned long long seed, unsigned long long ctr,
                   int32_t* act, float* logp, float* value, hipStream_t st) {
    if (n <= 0) return;
    hipLaunchKernelGGL(sample_kernel, dim3((n + 63) / 64), dim3(64), 0, st, hout, n, A, u, seed, ctr, act, logp, value);
}

// ------------------------------------------------------------------------------------------ clip + Adam
// torch.nn.utils.clip_grad_norm_(params, c) + optim.Adam(eps=1e-5).step() + zero_grad() (agents/ppo.py:174-176)
// over the flat parameter / gradient / moment buffers.
__global__ __launch_bounds__(1024) void sumsq_kernel(const float* g, long long n, double* out) {
    __shared__ double sb[16];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    double s = 0.0;
    for (long long k = tid; k < n; k += 1024) { const double x = (double)g[k]; s += x * x; }
    s = wave_sum(s);
    if (lane == 0) sb[w] = s;
    __syncthreads();
    if (tid == 0) { double t = 0.0; for (int k = 0; k < 16; ++k) t += sb[k]; out[0] = t; }
}
// 128 fixed chunks -> partials -> fixed-order final sum (deterministic, and 100x the bandwidth of one workgroup)
__global__ __launch_bounds__(256) void sumsq_partial_kernel(const float* g, long long n, double* part) {
    __shared__ double sb[4];
    const long long chunk = (n + gridDim.x - 1) / gridDim.x, beg = (long long)blockIdx.x * chunk;
    const long long end = beg + chunk < n ? beg + chunk : n;
    double s = 0.0;
    long long k = beg + threadIdx.x;
    for (; k + 3 * 256 < end; k += 4 * 256) {                  // 4 loads in flight, adds in index order
        const double x0 = (double)g[k], x1 = (double)g[k + 256], x2 = (double)g[k + 512], x3 = (double)g[k + 768];
        s += x0 * x0; s += x1 * x1; s += x2 * x2; s += x3 * x3;
    }
    for (; k < end; k += 256) { const double x = (double)g[k]; s += x * x; }
    const double t = block_sum256(s, sb);
    if (threadIdx.x == 0) part[blockIdx.x] = t;
}
__global__ void sumsq_final_kernel(const double* part, int n, double* out) {      // one wave; fixed (tree) order in fp64
    double t = 0.0;
    for (int k = threadIdx.x; k < n; k += 64) t += part[k];
    t = wave_sum(t);
    if (threadIdx.x == 0) out[0] = t;
}
void launch_sumsq(const float* g, long long n, double* out, double* part /* >= 128 doubles, or null */, hipStream_t st) {
    if (!part) { hipLaunchKernelGGL(sumsq_kernel, dim3(1), dim3(1024), 0, st, g, n, out); return; }
    hipLaunchKernelGGL(sumsq_partial_kernel, dim3(128), dim3(256), 0, st, g, n, part);
    hipLaunchKernelGGL(sumsq_final_kernel, dim3(1), dim3(64), 0, st, (const double*)part, 128, out);
}
// npart > 0: `sumsq` holds the npart fixed-chunk partial sums of launch_sumsq_partials and every wave adds them up itself, in exactly
// sumsq_final_kernel's order (lane k takes k, k + 64, ...; shuffle tree) -- one launch less per optimizer step, same bits
__global__ void adam_kernel(float* p, float* g, float* m, float* v, long long n, const double* sumsq, int npart, float max_norm, float lr_unused,
                            float beta1, float beta2, float eps, float step_size, float bc2_sqrt, float* gnorm_out) {
#pragma clang fp contract(off)
    const long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    double total;
    if (npart > 0) {
        double t = 0.0;
        for (int q = threadIdx.x & 63; q < npart; q += 64) t += sumsq[q];
        t = wave_sum(t);
        total = __shfl(t, 0, 64);
    } else total = sumsq[0];
    const float norm = (float)sqrt(total);
    float coef = max_norm / (norm + 1e-6f);
    coef = coef > 1.f ? 1.f : coef;
    if (k == 0 && gnorm_out) gnorm_out[0] = norm;
    if (k >= n) return;
    const float gk = g[k] * coef;
    float mk = m[k], vk = v[k];
    mk = mk + (gk - mk) * (1.f - beta1);                 // exp_avg.lerp_(grad, 1 - beta1)
    vk = vk * beta2 + ((1.f - beta2) * gk) * gk;         // exp_avg_sq.mul_(b2).addcmul_(g, g, value=1-b2)
    const float denom = sqrtf(vk) / bc2_sqrt + eps;
    p[k] = p[k] + (-step_size) * (mk / denom);           // param.addcdiv_(exp_avg, denom, value=-step_size)
    m[k] = mk; v[k] = vk;
    g[k] = 0.f;                                          // optimizer.zero_grad()
}
void launch_adam(float* p, float* g, float* m, float* v, long long n, const double* sumsq, int npart, float max_norm, float lr,
                 float beta1, float beta2, float eps, float step_size, float bc2_sqrt, float* gnorm_out, hipStream_t st) {
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, p, g, m, v, n, sumsq, npart, max_norm, lr,
                       beta1, beta2, eps, step_size, bc2_sqrt, gnorm_out);
}
void launch_sumsq_partials(const float* g, long long n, double* part /* 128 doubles */, hipStream_t st) {
    hipLaunchKernelGGL(sumsq_partial_kernel, dim3(128), dim3(256), 0, st, g, n, part);
}

// A minibatch's indices, pulled from the pinned host ring by a kernel on the compute stream: hipMemcpyAsync in that place (32 KB between
// one minibatch's optimizer step and the next one's first conv) left the chip idle for ~23 us per minibatch -- the compute queue hands
// over to the copy engine and back (rocprofv3 kernel trace: the only gap of the update phase); moving the copy to a stream of its own
// only traded it for the event wait.  Eight workgroups reading 16 bytes per lane over PCIe take ~4 us, in stream order, no hand-over.
__global__ __launch_bounds__(256) void pull_i32_kernel(const int32_t* __restrict__ host_src, int32_t* __restrict__ dst, int n) {
    const int k = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (k + 4 <= n) *(int4*)(dst + k) = *(const int4*)(host_src + k);
    else for (int j = k; j < n; ++j) dst[j] = host_src[j];
}
// The same for a group's frames (E/G x 12 KB per policy step): workgroups of 256 lanes, four 16-byte PCIe reads in flight per lane
typedef unsigned pull_u32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void pull_bytes_kernel(const pull_u32x4* __restrict__ host_src, pull_u32x4* __restrict__ dst, long long n16) {
    const long long k0 = (long long)blockIdx.x * 1024 + threadIdx.x;
    pull_u32x4 v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { long long k = k0 + j * 256; if (k >= n16) k = n16 - 1; v[j] = __builtin_nontemporal_load(host_src + k); }
#pragma unroll
    for (int j = 0; j < 4; ++j) { const long long k = k0 + j * 256; if (k < n16) dst[k] = v[j]; }
}
void launch_pull_bytes(const void* host_src, void* dst, size_t bytes /* multiple of 16 */, hipStream_t st) {
    const long long n16 = (long long)(bytes / 16);
    if (n16 <= 0) return;
    hipLaunchKernelGGL(pull_bytes_kernel, dim3((unsigned)((n16 + 1023) / 1024)), dim3(256), 0, st, (const pull_u32x4*)host_src, (pull_u32x4*)dst, n16);
}
void launch_pull_i32(const int32_t* host_src, int32_t* dst, int n, hipStream_t st) {
    if (n <= 0) return;
    hipLaunchKernelGGL(pull_i32_kernel, dim3((n + 1023) / 1024), dim3(256), 0, st, host_src, dst, n);
}

// ------------------------------------------------------------------------------------------ GRU cell (rollout only)
// nn.GRU single step (common/model.py:219-225): gi = W_ih x + b_ih, gh = W_hh (h*mask) + b_hh come from the GEMM;
// r = s(gi_r+gh_r), z = s(gi_z+gh_z), n = tanh(gi_n + r*gh_n), h' = (1-z)*n + z*h.
__global__ void mask_rows_kernel(const float* h, const float* done, float* out, int n, int H) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n * H) return;
    out[e] = h[e] * (1.f - done[e / H]);
}
void launch_mask_rows(const float* h, const float* done, float* out, int n, int H, hipStream_t st) {
    if (n <= 0) return;
    hipLaunchKernelGGL(mask_rows_kernel, dim3((n * H + 255) / 256), dim3(256), 0, st, h, done, out, n, H);
}
__global__ void gru_gates_kernel(const float* gi, const float* gh, const float* hm, float* h_out, float* feat_out, int n, int H) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n * H) return;
    const int row = e / H, j = e % H;
    const float* a = gi + (long long)row * 3 * H;
    const float* b = gh + (long long)row * 3 * H;
    const float r = 1.f / (1.f + expf(-(a[j] + b[j])));
    const float z = 1.f / (1.f + expf(-(a[H + j] + b[H + j])));
    const float nn = tanhf(a[2 * H + j] + r * b[2 * H + j]);
    const float hn = (1.f - z) * nn + z * hm[e];
    h_out[e] = hn;
    feat_out[e] = hn;
}
// d value / d (GRU input pre-activations): value = w_v . h' + b_v with h' = (1 - z) n + z h (gates as in gru_gates_kernel), so with
// g = w_v[j]:  dn = g (1 - z), dz = g (h - n);  d pre_n = dn (1 - n^2), d pre_z = dz z (1 - z), d pre_r = d pre_n gh_n r (1 - r).
// dgates[row] = {d pre_r, d pre_z, d pre_n} (3H): the gradient wrt gi = W_ih x + b_ih; dx = dgates W_ih follows as a GEMM.
__global__ void gru_value_bwd_kernel(const float* gi, const float* gh, const float* hm, const float* wv, float* dgates, int n, int H) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n * H) return;
    const int row = e / H, j = e % H;
    const float* a = gi + (long long)row * 3 * H;
    const float* b = gh + (long long)row * 3 * H;
    const float r = 1.f / (1.f + expf(-(a[j] + b[j])));
    const float z = 1.f / (1.f + expf(-(a[H + j] + b[H + j])));
    const float nn = tanhf(a[2 * H + j] + r * b[2 * H + j]);
    const float g = wv[j];
    const float dpn = g * (1.f - z) * (1.f - nn * nn);
    const float dpz = g * (hm[e] - nn) * z * (1.f - z);
    const float dpr = dpn * b[2 * H + j] * r * (1.f - r);
    float* d = dgates + (long long)row * 3 * H;
    d[j] = dpr; d[H + j] = dpz; d[2 * H + j] = dpn;
}
void launch_gru_value_bwd(const float* gi, const float* gh, const float* hm, const float* wv, float* dgates, int n, int H, hipStream_t st) {
    if (n <= 0) return;
    hipLaunchKernelGGL(gru_value_bwd_kernel, dim3((n * H + 255) / 256), dim3(256), 0, st, gi, gh, hm, wv, dgates, n, H);
}
void launch_gru_gates(const float* gi, const float* gh, const float* hm, float* h_out, float* feat_out, int n, int H, hipStream_t st) {
    if (n <= 0) return;
    hipLaunchKernelGGL(gru_gates_kernel, dim3((n * H + 255) / 256), dim3(256), 0, st, gi, gh, hm, h_out, feat_out, n, H);
}

// ------------------------------------------------------------------------------------------ value saliency (agents/ppo.py:83-94)
// value.backward() seeds dY = e_value for every env; the backward pass then runs down to the network input.  The last step,
// the input gradient of block1.conv (16 -> 3 channels @64x64), exists for this path only: one thread per pixel, direct form.
__global__ void value_seed_kernel(float* dY, int n, int A) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n * (A + 1)) return;
    dY[e] = (e % (A + 1) == A) ? 1.f : 0.f;
}
void launch_value_seed(float* dY, int n, int A, hipStream_t st) {
    if (n <= 0) return;
    hipLaunchKernelGGL(value_seed_kernel, dim3((n * (A + 1) + 255) / 256), dim3(256), 0, st, dY, n, A);
}
template <bool BF>
__global__ __launch_bounds__(256) void conv1_input_grad_kernel(const void* dC, const float* __restrict__ W, float* __restrict__ dX, int n) {
    __shared__ float sw[16 * 9 * 3];                       // device layout [co][tap][ci]
    for (int k = threadIdx.x; k < 432; k += 256) sw[k] = W[k];
    __syncthreads();
    const long long p = (long long)blockIdx.x * 256 + threadIdx.x;
    if (p >= (long long)n * 4096) return;
    const int x = (int)(p & 63), y = (int)((p >> 6) & 63);
    const long long img = p >> 12;
    float acc[3] = {0.f, 0.f, 0.f};
    for (int ky = 0; ky < 3; ++ky) {
        const int yy = y + 1 - ky;                          // forward: out[yy][xx] read in[yy - 1 + ky][xx - 1 + kx]
        if (yy < 0 || yy >= 64) continue;
        for (int kx = 0; kx < 3; ++kx) {
            const int xx = x + 1 - kx;
            if (xx < 0 || xx >= 64) continue;
            const long long o = ((img * 64 + yy) * 64 + xx) * 16;
            for (int co = 0; co < 16; ++co) {
                const float g = BF ? __uint_as_float(((unsigned)((const unsigned short*)dC)[o + co]) << 16) : ((const float*)dC)[o + co];
                const float* w = sw + (co * 9 + ky * 3 + kx) * 3;
                acc[0] += g * w[0]; acc[1] += g * w[1]; acc[2] += g * w[2];
            }
        }
    }
    dX[p * 3] = acc[0]; dX[p * 3 + 1] = acc[1]; dX[p * 3 + 2] = acc[2];
}
void launch_conv1_input_grad(const void* dC, int bf16, const float* W, float* dX, int n, hipStream_t st) {
    if (n <= 0) return;
    const unsigned grid = (unsigned)(((long long)n * 4096 + 255) / 256);
    if (bf16) hipLaunchKernelGGL(conv1_input_grad_kernel<true>, dim3(grid), dim3(256), 0, st, dC, W, dX, n);
    else hipLaunchKernelGGL(conv1_input_grad_kernel<false>, dim3(grid), dim3(256), 0, st, dC, W, dX, n);
}
