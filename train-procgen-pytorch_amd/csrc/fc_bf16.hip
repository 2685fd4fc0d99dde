// embedder.fc (2048 -> 256, common/model.py:176,199-200) on the bf16 matrix cores for the bf16 mode: forward,
// data gradient and weight gradient.  The activations (block3 output, bf16 NHWC = the flattened 2048 features) are
// read as stored; the fp32 gradient of the 256 features is rounded to bf16 while staged; fc.weight is kept as two
// packed bf16 images ([256][2048] for forward, [2048][256] for dgrad) refreshed after every optimizer step.
//   NT kernel : C[M][N] = A[M][K] * Bp[N][K]^T         (forward: +bias, ReLU, fp32 out; dgrad: ReLU mask, bf16 out)
//   TN kernel : gW[M][N] += A[K][M]^T * relu(B[K][N])   (weight gradient; both operands K(=batch)-major in memory ->
//               ds_read_b64_tr_b16 fragments; split over the batch, slabs summed in fixed order)
#include "common.h"

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef s16x4 __attribute__((address_space(3))) * lds_s16x4_ptr;
#define MFMA_BF16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16((a), (b), (c), 0, 0, 0)

__device__ __forceinline__ unsigned short fc_f2bf(float x) { __bf16 h = (__bf16)x; return __builtin_bit_cast(unsigned short, h); }
__device__ __forceinline__ unsigned fc_pack2(float lo, float hi) { return mi_pk_bf16(lo, hi); }
__device__ __forceinline__ unsigned fc_relu2(unsigned w) { const unsigned neg = (w >> 15) & 0x00010001u; return w & ~(neg * 0xFFFFu); }

__global__ void fc_pack_kernel(const float* __restrict__ w, unsigned short* __restrict__ wp, unsigned short* __restrict__ wt, int N, int K) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= N * K) return;
    const unsigned short h = fc_f2bf(w[e]);
    wp[e] = h;                                   // [n][k]
    wt[(long long)(e % K) * N + e / K] = h;      // [k][n]
}
void launch_fc_pack(const float* w, unsigned short* wp, unsigned short* wt, int N, int K, hipStream_t st) {
    hipLaunchKernelGGL(fc_pack_kernel, dim3((N * K + 255) / 256), dim3(256), 0, st, w, wp, wt, N, K);
}

// ------------------------------------------------------------------------------------------ NT
struct FcNtArgs {
    const void* A; const unsigned short* Bp; void* C;
    int M, N, K;
    const float* bias; const unsigned short* mask;      // mask: same shape as C (bf16), ReLU mask source
    int a_f32, relu_a, relu_out, c_bf16;
};
constexpr int NT_LD = 80;                                // LDS row stride (bf16 elements): 10 x 16-B slots, conflict free
// TM = rows of C per workgroup (128, or 64 when the grid would otherwise leave CUs with a single workgroup); a wave owns
// TM/4 rows x 64 columns.  The MFMA takes the B-tile rows (n) as its A operand and the A-tile rows (m) as B, so a lane's 4
// accumulator registers are 4 CONSECUTIVE n of one row m: bias / mask / output move as 16- or 8-byte words.
template <bool A_F32, int TM>
__global__ __launch_bounds__(256) void fc_nt_kernel(FcNtArgs g) {
    constexpr int MA = TM / 64;                          // 16-row tiles per wave
    __shared__ __attribute__((aligned(16))) unsigned short As[TM * NT_LD];
    __shared__ __attribute__((aligned(16))) unsigned short Bs[64 * NT_LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i = lane & 15, kq = lane >> 4;
    const int m0 = blockIdx.y * TM, n0 = blockIdx.x * 64;
    f32x4 acc[MA][4];
#pragma unroll
    for (int a = 0; a < MA; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
    uint4 ra[TM / 32], rb[2];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int e = 0; e < TM / 32; ++e) {              // A tile: TM rows x 8 chunks of 8 elements
            const int l = tid + e * 256, r = l >> 3, c8 = l & 7;
            uint4 v = {0u, 0u, 0u, 0u};
            if (m0 + r < g.M) {
                if constexpr (A_F32) {
                    const float* p = (const float*)g.A + (long long)(m0 + r) * g.K + k0 + c8 * 8;
                    const f32x4 lo = *(const f32x4*)p, hi = *(const f32x4*)(p + 4);
                    v = (uint4){fc_pack2(lo.x, lo.y), fc_pack2(lo.z, lo.w), fc_pack2(hi.x, hi.y), fc_pack2(hi.z, hi.w)};
                } else {
                    v = *(const uint4*)((const unsigned short*)g.A + (long long)(m0 + r) * g.K + k0 + c8 * 8);
                }
            }
            ra[e] = v;
        }
#pragma unroll
        for (int e = 0; e < 2; ++e) {                    // B tile: 64 rows x 8 chunks
            const int l = tid + e * 256, r = l >> 3, c8 = l & 7;
            uint4 v = {0u, 0u, 0u, 0u};
            if (n0 + r < g.N) v = *(const uint4*)(g.Bp + (long long)(n0 + r) * g.K + k0 + c8 * 8);
            rb[e] = v;
        }
    };
    fetch(0);
    for (int k0 = 0; k0 < g.K; k0 += 64) {
        __syncthreads();
#pragma unroll
        for (int e = 0; e < TM / 32; ++e) {
            const int l = tid + e * 256;
            uint4 v = ra[e];
            if (g.relu_a) { v.x = fc_relu2(v.x); v.y = fc_relu2(v.y); v.z = fc_relu2(v.z); v.w = fc_relu2(v.w); }
            *(uint4*)(As + (l >> 3) * NT_LD + (l & 7) * 8) = v;
        }
#pragma unroll
        for (int e = 0; e < 2; ++e) { const int l = tid + e * 256; *(uint4*)(Bs + (l >> 3) * NT_LD + (l & 7) * 8) = rb[e]; }
        __syncthreads();
        if (k0 + 64 < g.K) fetch(k0 + 64);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 av[MA], bv[4];
#pragma unroll
            for (int a = 0; a < MA; ++a) av[a] = *(const bf16x8*)(As + (wave * (TM / 4) + a * 16 + i) * NT_LD + ks * 32 + kq * 8);
#pragma unroll
            for (int b = 0; b < 4; ++b) bv[b] = *(const bf16x8*)(Bs + (b * 16 + i) * NT_LD + ks * 32 + kq * 8);
#pragma unroll
            for (int a = 0; a < MA; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) acc[a][b] = MFMA_BF16(bv[b], av[a], acc[a][b]);
        }
    }
#pragma unroll
    for (int a = 0; a < MA; ++a) {
        const int m = m0 + wave * (TM / 4) + a * 16 + i;
        if (m >= g.M) continue;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int n = n0 + b * 16 + kq * 4;          // n .. n+3 (N is a multiple of 64 for both uses)
            if (n >= g.N) continue;
            const long long o = (long long)m * g.N + n;
            float v[4] = {acc[a][b][0], acc[a][b][1], acc[a][b][2], acc[a][b][3]};
            if (g.bias) { const f32x4 bb = *(const f32x4*)(g.bias + n); v[0] += bb.x; v[1] += bb.y; v[2] += bb.z; v[3] += bb.w; }
            if (g.relu_out) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
            }
            if (g.mask) {                                 // mask > 0 (bf16 bits: not negative, not zero)
                const uint2 mk = *(const uint2*)(g.mask + o);
                const unsigned h[4] = {mk.x & 0xffffu, mk.x >> 16, mk.y & 0xffffu, mk.y >> 16};
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = ((h[r] & 0x8000u) || h[r] == 0) ? 0.f : v[r];
            }
            if (g.c_bf16) *(uint2*)((unsigned short*)g.C + o) = (uint2){fc_pack2(v[0], v[1]), fc_pack2(v[2], v[3])};
            else *(f32x4*)((float*)g.C + o) = (f32x4){v[0], v[1], v[2], v[3]};
        }
    }
}
void launch_fc_nt(const FcNtArgs& g, hipStream_t st) {
    if (g.M <= 0) return;
    const int tn = (g.N + 63) / 64;
    const bool small_grid = (long long)tn * ((g.M + 127) / 128) < 512;         // fewer than 2 workgroups per CU with 128-row tiles
    if (small_grid) {
        dim3 grid(tn, (g.M + 63) / 64);
        if (g.a_f32) hipLaunchKernelGGL((fc_nt_kernel<true, 64>), grid, dim3(256), 0, st, g);
        else hipLaunchKernelGGL((fc_nt_kernel<false, 64>), grid, dim3(256), 0, st, g);
    } else {
        dim3 grid(tn, (g.M + 127) / 128);
        if (g.a_f32) hipLaunchKernelGGL((fc_nt_kernel<true, 128>), grid, dim3(256), 0, st, g);
        else hipLaunchKernelGGL((fc_nt_kernel<false, 128>), grid, dim3(256), 0, st, g);
    }
}

// ------------------------------------------------------------------------------------------ TN (weight gradient)
// gW[m][n] (+)= sum_k A[k][m] * relu(B[k][n]);  A fp32 [K][M] (rounded to bf16), B bf16 [K][N]; tile 64(M) x 128(N),
// K tile 64; the 8 rows a half-wave's transpose read touches are 8 consecutive k (same permutation as the conv wgrad).
constexpr int TN_LDA = 80, TN_LDB = 144;                 // 16 x odd elements: conflict-free transpose reads
__global__ __launch_bounds__(256) void fc_tn_kernel(const float* __restrict__ A, const unsigned short* __restrict__ B, float* __restrict__ ws,
                                                    int M, int N, int K, int k_chunk) {
    __shared__ __attribute__((aligned(16))) unsigned short As[64 * TN_LDA];
    __shared__ __attribute__((aligned(16))) unsigned short Bs[64 * TN_LDB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i = lane & 15, kq = lane >> 4, rq = (lane & 15) >> 2, cp = lane & 3;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 128;
    const int kbeg = blockIdx.z * k_chunk, kend = min(K, kbeg + k_chunk);
    f32x4 acc[2][4];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
    uint4 ra[2], rb[4];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int e = 0; e < 2; ++e) {                    // A tile: 64 k-rows x 64 m (fp32 -> bf16): 8 chunks of 8 per row
            const int l = tid + e * 256, r = l >> 3, c8 = l & 7;
            uint4 v = {0u, 0u, 0u, 0u};
            if (k0 + r < kend) {
                const float* p = A + (long long)(k0 + r) * M + m0 + c8 * 8;
                const f32x4 lo = *(const f32x4*)p, hi = *(const f32x4*)(p + 4);
                v = (uint4){fc_pack2(lo.x, lo.y), fc_pack2(lo.z, lo.w), fc_pack2(hi.x, hi.y), fc_pack2(hi.z, hi.w)};
            }
            ra[e] = v;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {                    // B tile: 64 k-rows x 128 n: 16 chunks per row
            const int l = tid + e * 256, r = l >> 4, c8 = l & 15;
            uint4 v = {0u, 0u, 0u, 0u};
            if (k0 + r < kend) v = *(const uint4*)(B + (long long)(k0 + r) * N + n0 + c8 * 8);
            rb[e] = v;
        }
    };
    if (kbeg < kend) fetch(kbeg);
    for (int k0 = kbeg; k0 < kend; k0 += 64) {
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 2; ++e) { const int l = tid + e * 256; *(uint4*)(As + (l >> 3) * TN_LDA + (l & 7) * 8) = ra[e]; }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int l = tid + e * 256;
            uint4 v = rb[e];
            v.x = fc_relu2(v.x); v.y = fc_relu2(v.y); v.z = fc_relu2(v.z); v.w = fc_relu2(v.w);
            *(uint4*)(Bs + (l >> 4) * TN_LDB + (l & 15) * 8) = v;
        }
        __syncthreads();
        if (k0 + 64 < kend) fetch(k0 + 64);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            int row[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) row[h] = 32 * ks + 16 * (kq >> 1) + 8 * h + 4 * (kq & 1) + rq;
            bf16x8 av[2], bv[4];
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(As + row[0] * TN_LDA + wm * 32 + a * 16 + 4 * cp));
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(As + row[1] * TN_LDA + wm * 32 + a * 16 + 4 * cp));
                av[a] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
            }
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(Bs + row[0] * TN_LDB + wn * 64 + b * 16 + 4 * cp));
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(Bs + row[1] * TN_LDB + wn * 64 + b * 16 + 4 * cp));
                bv[b] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
            }
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) acc[a][b] = MFMA_BF16(av[a], bv[b], acc[a][b]);
        }
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + wm * 32 + a * 16 + kq * 4 + r, n = n0 + wn * 64 + b * 16 + i;
                ws[((long long)blockIdx.z * M + m) * N + n] = acc[a][b][r];
            }
}
__global__ void fc_tn_reduce_kernel(const float* __restrict__ ws, int split, long long total, float* __restrict__ gW) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    float s = 0.f;
    int z = 0;
    for (; z + 4 <= split; z += 4) {                           // 4 loads in flight, adds in split order
        const float t0 = ws[(long long)z * total + e], t1 = ws[(long long)(z + 1) * total + e];
        const float t2 = ws[(long long)(z + 2) * total + e], t3 = ws[(long long)(z + 3) * total + e];
        s += t0; s += t1; s += t2; s += t3;
    }
    for (; z < split; ++z) s += ws[(long long)z * total + e];
    gW[e] += s;
}
// A: fp32 [K][M], B: bf16 [K][N], gW fp32 [M][N] accumulated; ws: >= split*M*N floats.  M % 64 == 0, N % 128 == 0.
void launch_fc_tn(const float* A, const unsigned short* B, float* gW, float* ws, size_t ws_floats, int M, int N, int K, hipStream_t st) {
    if (K <= 0) return;
    int split = 8;
    while (split > 1 && ((size_t)split * M * N > ws_floats || K / split < 64)) split >>= 1;
    const int k_chunk = ((K + split - 1) / split + 63) / 64 * 64;
    split = (K + k_chunk - 1) / k_chunk;
    hipLaunchKernelGGL(fc_tn_kernel, dim3(N / 128, M / 64, split), dim3(256), 0, st, A, B, ws, M, N, K, k_chunk);
    const long long total = (long long)M * N;
    hipLaunchKernelGGL(fc_tn_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, (const float*)ws, split, total, gW);
}

// embedder.fc: y[n][256] = relu(relu(x)[n][2048] * W^T + b)
void launch_fc_fwd_bf16(const void* x_bf16, const unsigned short* wp, const float* bias, float* y, int n, hipStream_t st) {
    FcNtArgs g{};
    g.A = x_bf16; g.Bp = wp; g.C = y; g.M = n; g.N = 256; g.K = 2048; g.bias = bias; g.mask = nullptr;
    g.a_f32 = 0; g.relu_a = 1; g.relu_out = 1; g.c_bf16 = 0;
    launch_fc_nt(g, st);
}
// dx[n][2048] (bf16) = (dy[n][256] * W) * (x > 0)
void launch_fc_dgrad_bf16(const float* dy, const unsigned short* wt, const void* mask_bf16, void* dx_bf16, int n, hipStream_t st) {
    FcNtArgs g{};
    g.A = dy; g.Bp = wt; g.C = dx_bf16; g.M = n; g.N = 2048; g.K = 256; g.bias = nullptr; g.mask = (const unsigned short*)mask_bf16;
    g.a_f32 = 1; g.relu_a = 0; g.relu_out = 0; g.c_bf16 = 1;
    launch_fc_nt(g, st);
}

// ------------------------------------------------------------------------------------------ small-batch forward (rollout)
// n = E (256) rows: latency-bound, so the work is spread over (n/16) x (256/16) workgroups.  A workgroup owns a 16 x 16
// output tile; its 4 waves split K = 2048 (16 MFMA steps each), every lane loads exactly its own operand fragments
// straight from global memory / L2 (all 32 loads in flight before the first MFMA), the 4 partial tiles are summed through
// LDS in fixed order, + bias, ReLU.
__global__ __launch_bounds__(256) void fc_small_bf16_kernel(const unsigned short* __restrict__ X, const unsigned short* __restrict__ Wp,
                                                            const float* __restrict__ bias, float* __restrict__ feat, int n) {
    __shared__ float red[4][256];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i = lane & 15, kq = lane >> 4;
    const int e0 = blockIdx.y * 16, o0 = blockIdx.x * 16;
    const int env = e0 + i;
    const unsigned short* xa = X + (long long)(env < n ? env : n - 1) * 2048 + wave * 512 + kq * 8;
    const unsigned short* wb = Wp + (long long)(o0 + i) * 2048 + wave * 512 + kq * 8;
    uint4 ra[16], rb[16];
#pragma unroll
    for (int s = 0; s < 16; ++s) { ra[s] = *(const uint4*)(xa + s * 32); rb[s] = *(const uint4*)(wb + s * 32); }
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        const uint4 a = {fc_relu2(ra[s].x), fc_relu2(ra[s].y), fc_relu2(ra[s].z), fc_relu2(ra[s].w)};
        acc = MFMA_BF16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, rb[s]), acc);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) red[wave][(kq * 4 + r) * 16 + i] = acc[r];
    __syncthreads();
    const int row = tid >> 4, col = tid & 15;
    if (e0 + row < n) {
        const float v = ((red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid])) + bias[o0 + col];
        feat[(long long)(e0 + row) * 256 + o0 + col] = v > 0.f ? v : 0.f;
    }
}
void launch_fc_fwd_small_bf16(const void* X, const unsigned short* Wp, const float* bias, float* feat, int n, hipStream_t st) {
    if (n <= 0) return;
    hipLaunchKernelGGL(fc_small_bf16_kernel, dim3(16, (n + 15) / 16), dim3(256), 0, st, (const unsigned short*)X, Wp, bias, feat, n);
}
