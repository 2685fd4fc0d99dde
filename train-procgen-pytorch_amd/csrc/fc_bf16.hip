// embedder.fc (2048 -> 256, common/model.py:176,199-200) on the bf16 matrix cores for the bf16 mode: forward,
// data gradient and weight gradient.  The activations (block3 output, bf16 NHWC = the flattened 2048 features) are
// read as stored; the fp32 gradient of the 256 features is rounded to bf16 while staged; fc.weight is kept as two
// packed bf16 images ([256][2048] for forward, [2048][256] for dgrad) refreshed after every optimizer step.
//   NT kernel : C[M][N] = A[M][K] * Bp[N][K]^T         (forward: +bias, ReLU, fp32 out; dgrad: ReLU mask, bf16 out)
//   TN kernel : gW[M][N] += A[K][M]^T * relu(B[K][N])   (weight gradient; both operands K(=batch)-major in memory ->
//               ds_read_b64_tr_b16 fragments; split over the batch, slabs summed in fixed order)
#include "common.h"
#include <mutex>

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

// Workgroup -> (column block, row block) for a 1-D grid of ncb x nrb workgroups, XCD-aware: workgroups are dealt to the 8 XCDs
// round-robin by their linear id and every XCD has its own L2, so the ncb workgroups that read the SAME rows get ids 8 apart --
// same XCD, dispatched close together -- and the shared operand comes from HBM once instead of once per XCD (the forward kernel's x
// was fetched 4 times, the data gradient's fp32 dy 8 times: both ran at HBM speed on traffic they did not need).
__device__ __forceinline__ void fc_block_map(int ncb, int nrb, int& cb, int& rb) {
    const int L = blockIdx.x, per = 8 * ncb, full = (nrb / 8) * per;
    if (L < full) { const int g = L / per, w = L % per; cb = w / 8; rb = g * 8 + w % 8; }
    else { const int t = L - full; cb = t % ncb; rb = (nrb / 8) * 8 + t / ncb; }          // the last, partial group of row blocks
}
// ------------------------------------------------------------------------------------------ NT
struct FcNtArgs {
    const void* A; const unsigned short* Bp; void* C;
    int M, N, K;
    const float* bias; const unsigned short* mask;      // mask: same shape as C (bf16), ReLU mask source
    int a_f32, relu_a, relu_out, c_bf16;
};
constexpr int NT_BK = 128, NT_LD = NT_BK + 16;           // K tile; LDS row stride (bf16 elements): 16-B slots x odd, conflict free
// TM = rows of C per workgroup (128, or 64 when the grid would otherwise leave CUs with a single workgroup); a wave owns
// TM/4 rows x 64 columns.  The MFMA takes the B-tile rows (n) as its A operand and the A-tile rows (m) as B, so a lane's 4
// accumulator registers are 4 CONSECUTIVE n of one row m: bias / mask / output move as 16- or 8-byte words.
// A K step is one memory round trip (the next step's tiles are fetched into registers while this one is multiplied), so the tile is
// 128 deep: K = 2048 is 16 round trips.  The fetch is unconditional from clamped rows and keeps fp32 operands RAW -- converting
// (or zeroing) right behind the load made every step wait for its own loads; both happen when the registers are stored to LDS.
template <bool A_F32, int TM>
__global__ __launch_bounds__(256) void fc_nt_kernel(FcNtArgs g) {
    constexpr int MA = TM / 64;                          // 16-row tiles per wave
    constexpr int CH = NT_BK / 8;                        // 16-byte bf16 chunks per tile row
    constexpr int EA = TM * CH / 256, EB = 64 * CH / 256;
    extern __shared__ __attribute__((aligned(16))) unsigned short nt_smem[];
    unsigned short* As = nt_smem;
    unsigned short* Bs = nt_smem + TM * NT_LD;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i = lane & 15, kq = lane >> 4;
    const int m0 = blockIdx.y * TM, n0 = blockIdx.x * 64;
    f32x4 acc[MA][4];
#pragma unroll
    for (int a = 0; a < MA; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
    uint4 ra[A_F32 ? 2 * EA : EA], rb[EB];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int e = 0; e < EA; ++e) {                   // A tile: TM rows x CH chunks of 8 elements
            const int l = tid + e * 256, r = l / CH, c8 = l % CH, row = m0 + r < g.M ? m0 + r : g.M - 1;
            if constexpr (A_F32) {
                const uint4* p = (const uint4*)((const float*)g.A + (long long)row * g.K + k0 + c8 * 8);
                ra[2 * e] = p[0]; ra[2 * e + 1] = p[1];
            } else ra[e] = *(const uint4*)((const unsigned short*)g.A + (long long)row * g.K + k0 + c8 * 8);
        }
#pragma unroll
        for (int e = 0; e < EB; ++e) {                   // B tile: 64 rows x CH chunks
            const int l = tid + e * 256, r = l / CH, c8 = l % CH, row = n0 + r < g.N ? n0 + r : g.N - 1;
            rb[e] = *(const uint4*)(g.Bp + (long long)row * g.K + k0 + c8 * 8);
        }
    };
    fetch(0);
    for (int k0 = 0; k0 < g.K; k0 += NT_BK) {
        __syncthreads();
#pragma unroll
        for (int e = 0; e < EA; ++e) {
            const int l = tid + e * 256, r = l / CH, c8 = l % CH;
            uint4 v;
            if constexpr (A_F32) {
                const uint4 lo = ra[2 * e], hi = ra[2 * e + 1];
                v = (uint4){fc_pack2(__uint_as_float(lo.x), __uint_as_float(lo.y)), fc_pack2(__uint_as_float(lo.z), __uint_as_float(lo.w)),
                            fc_pack2(__uint_as_float(hi.x), __uint_as_float(hi.y)), fc_pack2(__uint_as_float(hi.z), __uint_as_float(hi.w))};
            } else v = ra[e];
            if (g.relu_a) { v.x = fc_relu2(v.x); v.y = fc_relu2(v.y); v.z = fc_relu2(v.z); v.w = fc_relu2(v.w); }
            if (m0 + r >= g.M) v = (uint4){0u, 0u, 0u, 0u};
            *(uint4*)(As + r * NT_LD + c8 * 8) = v;
        }
#pragma unroll
        for (int e = 0; e < EB; ++e) {
            const int l = tid + e * 256, r = l / CH, c8 = l % CH;
            *(uint4*)(Bs + r * NT_LD + c8 * 8) = n0 + r < g.N ? rb[e] : (uint4){0u, 0u, 0u, 0u};
        }
        __syncthreads();
        fetch(k0 + NT_BK < g.K ? k0 + NT_BK : k0);        // unconditional (the last tile is simply fetched again)
#pragma unroll
        for (int ks = 0; ks < NT_BK / 32; ++ks) {
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
template <bool A_F32, int TM>
static void launch_fc_nt_t(const FcNtArgs& g, hipStream_t st) {
    constexpr size_t LDS = (size_t)(TM + 64) * NT_LD * 2;
    static std::once_flag attr;          // (launchers run on up to 4 group worker threads)
    std::call_once(attr, [] { hipFuncSetAttribute((const void*)fc_nt_kernel<A_F32, TM>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS); });
    hipLaunchKernelGGL((fc_nt_kernel<A_F32, TM>), dim3((g.N + 63) / 64, (g.M + TM - 1) / TM), dim3(256), LDS, st, g);
}
void launch_fc_nt(const FcNtArgs& g, hipStream_t st) {          // K % 128 == 0 (2048 forward, 256 data gradient)
    if (g.M <= 0) return;
    const int tn = (g.N + 63) / 64;
    const bool small_grid = (long long)tn * ((g.M + 127) / 128) < 512;         // fewer than 2 workgroups per CU with 128-row tiles
    if (small_grid) { if (g.a_f32) launch_fc_nt_t<true, 64>(g, st); else launch_fc_nt_t<false, 64>(g, st); }
    else { if (g.a_f32) launch_fc_nt_t<true, 128>(g, st); else launch_fc_nt_t<false, 128>(g, st); }
}

// ------------------------------------------------------------------------------------------ TN (weight gradient)
// gW[m][n] (+)= sum_k A[k][m] * relu(B[k][n]);  A fp32 [K][M] (rounded to bf16), B bf16 [K][N]; tile 64(M) x 128(N),
// K tile 64; the 8 rows a half-wave's transpose read touches are 8 consecutive k (same permutation as the conv wgrad).
constexpr int TN_BK = 128, TN_LDA = 80, TN_LDB = 144;    // K (= batch) tile; LDS row strides: 16 x odd elements, conflict-free transpose reads
// (same staging rules as the NT kernel: 128-deep K tile = half the memory round trips, unconditional clamped loads, fp32 operand kept
// raw in registers and rounded / zeroed when stored)
__global__ __launch_bounds__(256) void fc_tn_kernel(const float* __restrict__ A, const unsigned short* __restrict__ B, float* __restrict__ ws,
                                                    int M, int N, int K, int k_chunk) {
    extern __shared__ __attribute__((aligned(16))) unsigned short tn_smem[];
    unsigned short* As = tn_smem;
    unsigned short* Bs = tn_smem + TN_BK * TN_LDA;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i = lane & 15, kq = lane >> 4, rq = (lane & 15) >> 2, cp = lane & 3;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 128;
    const int kbeg = blockIdx.z * k_chunk, kend = min(K, kbeg + k_chunk);
    f32x4 acc[2][4];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
    constexpr int EA = TN_BK * 8 / 256, EB = TN_BK * 16 / 256;
    uint4 ra[2 * EA], rb[EB];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int e = 0; e < EA; ++e) {                   // A tile: TN_BK k-rows x 64 m (fp32, raw): 8 chunks of 8 per row
            const int l = tid + e * 256, r = l >> 3, c8 = l & 7, row = k0 + r < kend ? k0 + r : kend - 1;
            const uint4* p = (const uint4*)(A + (long long)row * M + m0 + c8 * 8);
            ra[2 * e] = p[0]; ra[2 * e + 1] = p[1];
        }
#pragma unroll
        for (int e = 0; e < EB; ++e) {                   // B tile: TN_BK k-rows x 128 n: 16 chunks per row
            const int l = tid + e * 256, r = l >> 4, c8 = l & 15, row = k0 + r < kend ? k0 + r : kend - 1;
            rb[e] = *(const uint4*)(B + (long long)row * N + n0 + c8 * 8);
        }
    };
    if (kbeg < kend) fetch(kbeg);
    for (int k0 = kbeg; k0 < kend; k0 += TN_BK) {
        __syncthreads();
#pragma unroll
        for (int e = 0; e < EA; ++e) {
            const int l = tid + e * 256, r = l >> 3;
            const uint4 lo = ra[2 * e], hi = ra[2 * e + 1];
            uint4 v = {fc_pack2(__uint_as_float(lo.x), __uint_as_float(lo.y)), fc_pack2(__uint_as_float(lo.z), __uint_as_float(lo.w)),
                       fc_pack2(__uint_as_float(hi.x), __uint_as_float(hi.y)), fc_pack2(__uint_as_float(hi.z), __uint_as_float(hi.w))};
            if (k0 + r >= kend) v = (uint4){0u, 0u, 0u, 0u};
            *(uint4*)(As + r * TN_LDA + (l & 7) * 8) = v;
        }
#pragma unroll
        for (int e = 0; e < EB; ++e) {
            const int l = tid + e * 256, r = l >> 4;
            uint4 v = rb[e];
            v.x = fc_relu2(v.x); v.y = fc_relu2(v.y); v.z = fc_relu2(v.z); v.w = fc_relu2(v.w);
            if (k0 + r >= kend) v = (uint4){0u, 0u, 0u, 0u};
            *(uint4*)(Bs + r * TN_LDB + (l & 15) * 8) = v;
        }
        __syncthreads();
        fetch(k0 + TN_BK < kend ? k0 + TN_BK : k0);       // unconditional (the last tile is simply fetched again)
#pragma unroll
        for (int ks = 0; ks < TN_BK / 32; ++ks) {
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
    while (split > 1 && ((size_t)split * M * N > ws_floats || K / split < TN_BK)) split >>= 1;
    const int k_chunk = ((K + split - 1) / split + TN_BK - 1) / TN_BK * TN_BK;
    split = (K + k_chunk - 1) / k_chunk;
    constexpr size_t LDS = (size_t)TN_BK * (TN_LDA + TN_LDB) * 2;
    static std::once_flag attr;          // (launchers run on up to 4 group worker threads)
    std::call_once(attr, [] { hipFuncSetAttribute((const void*)fc_tn_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS); });
    hipLaunchKernelGGL(fc_tn_kernel, dim3(N / 128, M / 64, split), dim3(256), LDS, st, A, B, ws, M, N, K, k_chunk);
    const long long total = (long long)M * N;
    hipLaunchKernelGGL(fc_tn_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, (const float*)ws, split, total, gW);
}

// ------------------------------------------------------------------------------------------ forward, dedicated kernel
// y[n][256] = relu(relu(x)[n][2048] x W^T + b).  K = 2048 is the long dimension: 16 steps of 128.  A workgroup owns 64 rows x 64 columns
// (wave w: rows 16w .. 16w+15); the x operand never touches LDS -- every lane loads its own MFMA fragments (16 bytes of one row) straight
// from global memory one step ahead, into one of two register sets -- and the 64 x 128 slice of the packed weight goes through a
// double-buffered LDS tile (one barrier per step).  512 workgroups at n = 8192: two or more per CU overlap each other's round trips
// (the tiled NT kernel: one round trip per step and nothing to overlap it, 33 us).  n % 64 == 0; other sizes take the NT kernel.
constexpr int FF_BK = 128, FF_LD = FF_BK + 16;
constexpr size_t FF_LDS = (size_t)2 * 64 * FF_LD * 2;
__global__ __launch_bounds__(256) void fc_fwd_bf16_kernel(const unsigned short* __restrict__ x, const unsigned short* __restrict__ wp,
                                                          const float* __restrict__ bias, float* __restrict__ y, int nrb) {
    typedef unsigned ff_u32x4 __attribute__((ext_vector_type(4)));
    extern __shared__ __attribute__((aligned(16))) unsigned short ff_smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i = lane & 15, kq = lane >> 4;
    int cb, rbk; fc_block_map(4, nrb, cb, rbk);
    const int m0 = rbk * 64 + wave * 16, n0 = cb * 64;
    const unsigned short* xrow = x + (long long)(m0 + i) * 2048 + kq * 8;
    struct XF { ff_u32x4 f[4]; };                           // the lane's x fragments of one K step (4 MFMA k-steps of 32)
    auto fetch_x = [&](XF& X, int k0) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) X.f[ks] = *(const ff_u32x4*)(xrow + k0 + ks * 32);
    };
    ff_u32x4 rb[4];                                         // staging: 64 weight rows x 16 chunks of the next slice
    auto fetch_w = [&](int k0) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { const int l = tid + e * 256; rb[e] = *(const ff_u32x4*)(wp + (long long)(n0 + (l >> 4)) * 2048 + k0 + (l & 15) * 8); }
    };
    f32x4 acc[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[b] = (f32x4){0.f, 0.f, 0.f, 0.f};
    XF X0, X1;
    fetch_w(0); fetch_x(X0, 0);
    auto step = [&](int st, XF& Xcur, XF& Xnext) {
        unsigned short* Ws = ff_smem + (st & 1) * 64 * FF_LD;
#pragma unroll
        for (int e = 0; e < 4; ++e) { const int l = tid + e * 256; *(ff_u32x4*)(Ws + (l >> 4) * FF_LD + (l & 15) * 8) = rb[e]; }
        __syncthreads();                                    // (also: every wave is done with the other buffer, written next step)
        const int kn = (st + 1 < 2048 / FF_BK ? st + 1 : st) * FF_BK;      // unconditional (the last step is simply fetched again)
        fetch_w(kn); fetch_x(Xnext, kn);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const ff_u32x4 r = Xcur.f[ks];
            const bf16x8 xv = __builtin_bit_cast(bf16x8, (ff_u32x4){fc_relu2(r.x), fc_relu2(r.y), fc_relu2(r.z), fc_relu2(r.w)});
#pragma unroll
            for (int b = 0; b < 4; ++b) acc[b] = MFMA_BF16(*(const bf16x8*)(Ws + (b * 16 + i) * FF_LD + ks * 32 + kq * 8), xv, acc[b]);
        }
    };
#pragma unroll 1
    for (int st = 0; st < 2048 / FF_BK; st += 2) { step(st, X0, X1); step(st + 1, X1, X0); }
#pragma unroll
    for (int b = 0; b < 4; ++b) {                           // lane: row m0 + i, columns n0 + 16 b + 4 kq .. +3
        const int n = n0 + b * 16 + kq * 4;
        const f32x4 bb = *(const f32x4*)(bias + n);
        f32x4 v = acc[b] + bb;
        v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
        *(f32x4*)(y + (long long)(m0 + i) * 256 + n) = v;
    }
}
// embedder.fc: y[n][256] = relu(relu(x)[n][2048] * W^T + b)
void launch_fc_fwd_bf16(const void* x_bf16, const unsigned short* wp, const float* bias, float* y, int n, hipStream_t st) {
    if (n > 0 && n % 64 == 0 && bias) {
        static std::once_flag attr;          // (launchers run on up to 4 group worker threads)
        std::call_once(attr, [] { hipFuncSetAttribute((const void*)fc_fwd_bf16_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)FF_LDS); });
        hipLaunchKernelGGL(fc_fwd_bf16_kernel, dim3(4 * (n / 64)), dim3(256), FF_LDS, st, (const unsigned short*)x_bf16, wp, bias, y, n / 64);
        return;
    }
    FcNtArgs g{};
    g.A = x_bf16; g.Bp = wp; g.C = y; g.M = n; g.N = 256; g.K = 2048; g.bias = bias; g.mask = nullptr;
    g.a_f32 = 0; g.relu_a = 1; g.relu_out = 1; g.c_bf16 = 0;
    launch_fc_nt(g, st);
}
// ------------------------------------------------------------------------------------------ data gradient, dedicated kernel
// dx[n][2048] (bf16) = (dy[n][256] x W) * (x > 0), W as the packed image wt [2048][256].  K = 256 is short and N = 2048 long, so the
// tile-per-workgroup NT kernel re-read the fp32 dy rows once per 64-column tile (32 x 8 MB through L2: it ran at L2 bandwidth,
// 37.8 us per 8192 rows).  Here a workgroup (4 waves x 32 rows = 128 rows) keeps its dy rows in REGISTERS as MFMA fragments for the
// whole K (2 row tiles x 8 K steps, rounded to bf16 once) and walks 256 columns in 4 steps of 64: per step the 64 x 256 slice of wt
// goes through a double-buffered LDS tile (one barrier per step), the mask words of the NEXT step are already in flight, and the lane
// layout (wt rows as the MFMA A operand) leaves 4 consecutive columns of one row per lane: 8-byte mask loads and stores.
// n % 128 == 0 (the training minibatch sizes); other sizes take the NT kernel.
constexpr int FD_LD = 256 + 16;                           // LDS row stride of the wt slice (bf16 elements)
constexpr size_t FD_LDS = (size_t)2 * 64 * FD_LD * 2;
constexpr int FD_STEPS = 4;                              // 64-column steps per workgroup: 2048 / (64 * 4) = 8 column ranges x n / 128 row blocks = 2 workgroups per CU at n = 8192
#ifdef FC_TIMING
__device__ unsigned long long g_fc_timing[8];
#endif
__global__ __launch_bounds__(256) void fc_dgrad_bf16_kernel(const float* __restrict__ dy, const unsigned short* __restrict__ wt,
                                                            const unsigned short* __restrict__ mask, unsigned short* __restrict__ dx, int nrb) {
    extern __shared__ __attribute__((aligned(16))) unsigned short fd_smem[];
#ifdef FC_TIMING
    long long tacc_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast_ = clock64();
#define FDCK(q) do { if (threadIdx.x == 0) { const long long now_ = clock64(); tacc_[q] += now_ - tlast_; tlast_ = now_; } } while (0)
#else
#define FDCK(q) do { } while (0)
#endif
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i = lane & 15, kq = lane >> 4;
    int cb, rbk; fc_block_map(2048 / (64 * FD_STEPS), nrb, cb, rbk);
    const int m0 = rbk * 128 + wave * 32, n0 = cb * (64 * FD_STEPS);
    // dy fragments: row tile a (16 rows), K step ks: lane (i, kq) holds dy[m0 + 16a + i][32 ks + 8 kq .. +7]
    bf16x8 av[2][8];
#pragma unroll
    for (int a = 0; a < 2; ++a) {                          // (one row tile at a time: 64 staging registers, not 128)
        uint4 raw[8][2];
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            const uint4* p = (const uint4*)(dy + (long long)(m0 + a * 16 + i) * 256 + ks * 32 + kq * 8);
            raw[ks][0] = p[0]; raw[ks][1] = p[1];
        }
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            const uint4 lo = raw[ks][0], hi = raw[ks][1];
            av[a][ks] = __builtin_bit_cast(bf16x8, (uint4){fc_pack2(__uint_as_float(lo.x), __uint_as_float(lo.y)), fc_pack2(__uint_as_float(lo.z), __uint_as_float(lo.w)),
                                                           fc_pack2(__uint_as_float(hi.x), __uint_as_float(hi.y)), fc_pack2(__uint_as_float(hi.z), __uint_as_float(hi.w))});
        }
        asm volatile("" ::: "memory");
    }
    // (ext-vector type: a straight global -> register -> LDS copy of the HIP uint4 STRUCT compiles to memcpy through a private array,
    //  which lands in scratch memory)
    typedef unsigned fd_u32x4 __attribute__((ext_vector_type(4)));
    fd_u32x4 rb[8];                                         // staging: 64 rows x 32 chunks of the next wt slice
    auto fetch_b = [&](int st) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { const int l = tid + e * 256; rb[e] = *(const fd_u32x4*)(wt + (long long)(n0 + st * 64 + (l >> 5)) * 256 + (l & 31) * 8); }
    };
    uint2 mk[2][4];                                         // mask words of the step in progress (fetched behind the previous step's epilogue)
    auto fetch_mask = [&](int st) {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) mk[a][b] = *(const uint2*)(mask + (long long)(m0 + a * 16 + i) * 2048 + n0 + st * 64 + b * 16 + kq * 4);
    };
    FDCK(0);                                                // dy fragments loaded and rounded
    fetch_b(0); fetch_mask(0);
#pragma unroll 2
    for (int st = 0; st < FD_STEPS; ++st) {
        unsigned short* Bs = fd_smem + (st & 1) * 64 * FD_LD;
#pragma unroll
        for (int e = 0; e < 8; ++e) { const int l = tid + e * 256; *(fd_u32x4*)(Bs + (l >> 5) * FD_LD + (l & 31) * 8) = rb[e]; }
        FDCK(1);                                            // wait for the staged slice + LDS stores
        __syncthreads();
        FDCK(2);                                    // (also: every wave is done with the other buffer, written next step)
        const int nx = st + 1 < FD_STEPS ? st + 1 : st;            // unconditional (the last slice / mask is simply fetched again)
        fetch_b(nx);
        f32x4 acc[2][4];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            bf16x8 bv[4];
#pragma unroll
            for (int b = 0; b < 4; ++b) bv[b] = *(const bf16x8*)(Bs + (b * 16 + i) * FD_LD + ks * 32 + kq * 8);
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) acc[a][b] = MFMA_BF16(bv[b], av[a][ks], acc[a][b]);
        }
        FDCK(3);                                            // next slice requested + MFMAs
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const uint2 w = mk[a][b];
                const unsigned h[4] = {w.x & 0xffffu, w.x >> 16, w.y & 0xffffu, w.y >> 16};
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = ((h[r] & 0x8000u) || h[r] == 0) ? 0.f : acc[a][b][r];      // mask > 0 (bf16 bits: not negative, not zero)
                *(uint2*)(dx + (long long)(m0 + a * 16 + i) * 2048 + n0 + st * 64 + b * 16 + kq * 4) = (uint2){fc_pack2(v[0], v[1]), fc_pack2(v[2], v[3])};
            }
        fetch_mask(nx);
        FDCK(4);                                            // epilogue (wait for the mask words, stores) + next mask requested
    }
#ifdef FC_TIMING
    if (threadIdx.x == 0) for (int q = 0; q < 8; ++q) atomicAdd(&g_fc_timing[q], (unsigned long long)tacc_[q]);
#endif
}
// dx[n][2048] (bf16) = (dy[n][256] * W) * (x > 0)
void launch_fc_dgrad_bf16(const float* dy, const unsigned short* wt, const void* mask_bf16, void* dx_bf16, int n, hipStream_t st) {
    if (n > 0 && n % 128 == 0 && mask_bf16) {
        static std::once_flag attr;          // (launchers run on up to 4 group worker threads)
        std::call_once(attr, [] { hipFuncSetAttribute((const void*)fc_dgrad_bf16_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)FD_LDS); });
        hipLaunchKernelGGL(fc_dgrad_bf16_kernel, dim3(2048 / (64 * FD_STEPS) * (n / 128)), dim3(256), FD_LDS, st, dy, wt, (const unsigned short*)mask_bf16, (unsigned short*)dx_bf16, n / 128);
        return;
    }
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
    typedef unsigned fs_u32x4 __attribute__((ext_vector_type(4)));
    fs_u32x4 ra[16], rb[16];
#pragma unroll
    for (int s = 0; s < 16; ++s) { ra[s] = *(const fs_u32x4*)(xa + s * 32); rb[s] = *(const fs_u32x4*)(wb + s * 32); }
    __builtin_amdgcn_sched_barrier(0);                     // (the scheduler otherwise sinks the loads between the MFMAs: 4 round trips instead of 1)
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        const fs_u32x4 a = {fc_relu2(ra[s].x), fc_relu2(ra[s].y), fc_relu2(ra[s].z), fc_relu2(ra[s].w)};
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
