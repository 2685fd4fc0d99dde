// Rollout-sized forward of IMPALA blocks 2 and 3 in ONE launch (bf16 mode): block2.conv + max pool + res1 + res2, block3.conv + max
// pool + res1 + res2 -- common/model.py:149-161 twice -- for n <= 256 images, one 512-thread workgroup per image
// (20.5 us at n = 256 / 18.5 us at n = 64 against 28.8 / 33 us for the four launches: profiles/r02_rollout_*).
//
// Why: a policy step (agents/ppo.py:72-81) runs the network on n_envs frames, one per env: every launch of the step is latency-
// bound (4-12 us whatever n is: bank copy, tile staging, a handful of dependent LDS -> MFMA -> LDS round trips, drain), and the step
// is a dependent chain on the rollout's critical path 257 times per iteration.  Four of its eight launches (conv+pool 16->32,
// res pair @16x16, conv+pool 32->32, res pair @8x8: 33 us at n = 64) become one: a 32x32x16 image (32 KB) and everything derived
// from it fit a workgroup's LDS, layer boundaries are workgroup barriers, and the ten filter banks stream through two LDS buffers
// one conv ahead of their use.
//
// Arithmetic = that of the kernels it replaces (conv_pool_fwd_bf16_kernel, resblock_pair_bf16_kernel): same bf16 banks, same
// K order per output pixel (tap-major MFMA steps accumulated in ascending order), same rounding points (conv + bias -> bf16 before the
// pool; conv1 output and block output -> bf16), same first-maximum pooling on order-preserving keys -- the outputs are bit-identical
// (tests/test_gpu_bf16.py::test_fused_rollout_tail_equals_the_four_launches).  Inference only: nothing but the block-3 output is stored.
#include "common.h"
#include <mutex>

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef short rt_s16x2 __attribute__((ext_vector_type(2)));
#define MFMA_BF16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16((a), (b), (c), 0, 0, 0)
__device__ __forceinline__ unsigned rt_relu2(unsigned w) { const unsigned neg = (w >> 15) & 0x00010001u; return w & ~(neg * 0xFFFFu); }
__device__ __forceinline__ unsigned rt_relu2_max(unsigned w) {
    return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(rt_s16x2, w), (rt_s16x2){0, 0}));
}
__device__ __forceinline__ float rt_lane(uint2 w, int r) {
    const unsigned u = (r >> 1) ? w.y : w.x;
    return (r & 1) ? __uint_as_float(u & 0xffff0000u) : __uint_as_float(u << 16);
}
__device__ __forceinline__ uint2 rt_pack(const float (&v)[4]) { return (uint2){mi_pk_bf16(v[0], v[1]), mi_pk_bf16(v[2], v[3])}; }

#ifndef RT_NT
#define RT_NT 512           // threads per workgroup (= per image).  1024 (one 16x16 pixel tile per wave, 4 waves per SIMD) needs <= 128
                            // registers, spills 32 of them and measured slower: 74.6 vs 69.3 us per n = 256 policy step
#endif
namespace rt {
constexpr int NT = RT_NT, NW = NT / 64;
constexpr int S16 = 16, S32 = 48, SCS = 48;                 // pixel strides of the 16- / 32-channel images, of the key tiles (1.5 pixels: see convpool_bf16.hip)
constexpr int P32 = 34, P16 = 18, P8 = 10;                  // haloed row lengths
constexpr int IN_ELEMS = P32 * P32 * S16;                   // block2.conv input image (18 496)
constexpr int X16_ELEMS = P16 * P16 * S32;                  // a haloed 16x16x32 image (15 552)
constexpr int X8_ELEMS = P8 * P8 * S32;                     // a haloed 8x8x32 image (4 800)
constexpr int KB2_ELEMS = 9 * 33 * SCS;                     // block2.conv key band: 9 conv rows x (1 pad + 32) cells (14 256)
constexpr int KB3_ELEMS = 17 * 17 * SCS;                    // block3.conv keys: conv rows -1 .. 15 x (1 pad + 16) cells (13 872)
constexpr int WS16 = 5 * 32 + 16, WS32 = 9 * 32 + 16;       // bank row lengths (conv_bf16.hip pack_banks_kernel)
constexpr int W_ELEMS = 32 * WS32;                          // the largest bank (9 728)
constexpr int R_IN = 0, R_X = R_IN + IN_ELEMS, R_Y = R_X + X16_ELEMS, R_W = R_Y + X16_ELEMS, R_B = R_W + 2 * W_ELEMS;
constexpr int LDS_ELEMS = R_B + 10 * 32 * 2;                // + ten bias vectors (fp32)
static_assert(KB2_ELEMS <= X16_ELEMS && KB3_ELEMS <= IN_ELEMS && 2 * X8_ELEMS <= X16_ELEMS, "region reuse");
static_assert(LDS_ELEMS * 2 <= 160 * 1024, "LDS");
constexpr int KW = (W_ELEMS / 8 + NT - 1) / NT;             // 16-byte words of a bank per thread (3)
}

struct RolloutTailArgs {
    const unsigned short* x;        // block1 output, bf16 NHWC [n][32][32][16]
    unsigned short* y;              // block3 output, bf16 NHWC [n][8][8][32]
    int n;
    const unsigned short* bank[10]; // block2.{conv, res1.conv1, res1.conv2, res2.conv1, res2.conv2}, block3.{...}: forward banks
    const float* bias[10];
};

// 3x3 conv over a haloed LDS image for MTC pixel tiles x both 16-channel output blocks (K steps in ascending order, as rb_conv)
// nmt (wave-uniform) <= MTC tiles are live: the others issue nothing
// KO: per-K-step operand offsets -- a lane-dependent table (16-channel source) or compile-time constants (32-channel sources: the
// lane's 8-channel chunk kq * 8 is part of abase, so every read is base register + immediate)
struct KoffTab { int v[5]; __device__ __forceinline__ int operator()(int m) const { return v[m]; } };
template <int P> struct KoffC { __device__ __forceinline__ constexpr int operator()(int m) const { return ((m / 3) * P + (m % 3)) * 48; } };
template <int NK, int WS, bool RELU_A, int MTC, class KO>
__device__ __forceinline__ void rt_conv(const unsigned short* s_src, const unsigned short* s_w, const KO koff, const int (&abase)[MTC],
                                        int bbase, f32x4 (&acc)[MTC][2], int nmt = MTC) {
#pragma unroll
    for (int mt = 0; mt < MTC; ++mt) { acc[mt][0] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc[mt][1] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
    for (int m = 0; m < NK; ++m) {
        bf16x8 av[MTC];
#pragma unroll
        for (int mt = 0; mt < MTC; ++mt) {
            if (mt >= nmt) continue;
            av[mt] = *(const bf16x8*)(s_src + abase[mt] + koff(m));
            if (RELU_A) {
                const uint4 u = __builtin_bit_cast(uint4, av[mt]);
                av[mt] = __builtin_bit_cast(bf16x8, (uint4){rt_relu2_max(u.x), rt_relu2_max(u.y), rt_relu2_max(u.z), rt_relu2_max(u.w)});
            }
        }
        const bf16x8 b0 = *(const bf16x8*)(s_w + bbase + m * 32), b1 = *(const bf16x8*)(s_w + bbase + 16 * WS + m * 32);
#pragma unroll
        for (int mt = 0; mt < MTC; ++mt)
            if (mt < nmt) { acc[mt][0] = MFMA_BF16(b0, av[mt], acc[mt][0]); acc[mt][1] = MFMA_BF16(b1, av[mt], acc[mt][1]); }
    }
}
// one pixel tile x ONE output block (the 8x8 stage: 4 tiles x 2 blocks = one task per wave)
template <bool RELU_A>
__device__ __forceinline__ f32x4 rt_conv1(const unsigned short* s_src, const unsigned short* s_w, int abase, int bbase) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    constexpr KoffC<10> koff;
#pragma unroll
    for (int m = 0; m < 9; ++m) {
        bf16x8 av = *(const bf16x8*)(s_src + abase + koff(m));
        if (RELU_A) {
            const uint4 u = __builtin_bit_cast(uint4, av);
            av = __builtin_bit_cast(bf16x8, (uint4){rt_relu2_max(u.x), rt_relu2_max(u.y), rt_relu2_max(u.z), rt_relu2_max(u.w)});
        }
        acc = MFMA_BF16(*(const bf16x8*)(s_w + bbase + m * 32), av, acc);
    }
    return acc;
}

__global__ __launch_bounds__(rt::NT) void rollout_tail_bf16_kernel(RolloutTailArgs a) {
    using namespace rt;
    extern __shared__ __attribute__((aligned(16))) unsigned short smem_h[];
    unsigned short* s_in = smem_h + R_IN;
    unsigned short* s_x = smem_h + R_X;
    unsigned short* s_y = smem_h + R_Y;
    unsigned short* s_w[2] = {smem_h + R_W, smem_h + R_W + W_ELEMS};
    float* s_b = (float*)(smem_h + R_B);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i = lane & 15, kq = lane >> 4;

    KoffTab koff2;
#pragma unroll
    for (int m = 0; m < 5; ++m) { int tap = 2 * m + (kq >> 1); const int chunk = kq & 1; if (tap > 8) tap = 8; koff2.v[m] = ((tap / 3) * P32 + (tap % 3)) * S16 + chunk * 8; }
    static_assert(S32 == 48, "KoffC assumes the 32-channel pixel stride");
    const int bb16 = i * WS16 + kq * 8, bb32 = i * WS32 + kq * 8;

    // Bank k lives in LDS buffer k & 1 and travels through register set k & 1: fetched TWO convs ahead of its use (a conv phase is
    // ~0.5 us, an L2 round trip ~1 us: one conv of lead left every commit waiting for its loads), written to LDS at the end of conv k - 1.
    // (staging registers are ext-vector typed: a copy of the HIP uint4 STRUCT from global to a local array and on to LDS compiles to memcpy
    //  through a private array, i.e. through scratch memory)
    typedef unsigned rt_u32x4 __attribute__((ext_vector_type(4)));
    rt_u32x4 wreg[2][KW];
    auto bank_fetch = [&](int k) {
        const int words = (k == 0 ? 32 * WS16 : W_ELEMS) / 8;
#pragma unroll
        for (int q = 0; q < KW; ++q) { const int e = tid + q * NT; wreg[k & 1][q] = e < words ? ((const rt_u32x4*)a.bank[k])[e] : (rt_u32x4){0u, 0u, 0u, 0u}; }
    };
    auto bank_commit = [&](int k) {
        const int words = (k == 0 ? 32 * WS16 : W_ELEMS) / 8;
#pragma unroll
        for (int q = 0; q < KW; ++q) { const int e = tid + q * NT; if (e < words) ((rt_u32x4*)s_w[k & 1])[e] = wreg[k & 1][q]; }
    };
    // end of conv c (c >= 1): bank c + 1 goes to LDS (its buffer's previous bank, c - 1, was last read in conv c - 1), bank c + 3 is requested
    auto bank_step = [&](int c) { bank_commit(c + 1); if (c + 3 <= 9) bank_fetch(c + 3); };

    for (int img = blockIdx.x; img < a.n; img += gridDim.x) {
        // ---- every first load of the image is requested before anything waits: the frame (4 words per thread), banks 0 / 1, the biases
        rt_u32x4 xin[2048 / NT];
        {
            const rt_u32x4* g = (const rt_u32x4*)(a.x + (long long)img * 32 * 32 * 16);
#pragma unroll
            for (int q = 0; q < 2048 / NT; ++q) xin[q] = g[tid + q * NT];
        }
        bank_fetch(0); bank_fetch(1);
        const float bias_v = tid < 320 ? a.bias[tid >> 5][tid & 31] : 0.f;
        __syncthreads();                                    // (a second image of this workgroup: everyone is done with the previous one)
        // ---- stage the 32x32x16 image (haloed, zero borders); clear the first residual image
        for (int e = tid; e < (IN_ELEMS + X16_ELEMS) / 8; e += NT) ((uint4*)smem_h)[e] = (uint4){0u, 0u, 0u, 0u};
        for (int e = tid; e < 9 * 16; e += NT) ((unsigned*)s_y)[(e >> 4) * 33 * (SCS / 2) + (e & 15)] = MI_KEY_MIN2;      // pad cell (column -1) of the 9 key rows
        if (tid < 320) s_b[tid] = bias_v;
        bank_commit(0);
        bank_fetch(2);
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 2048 / NT; ++q) {               // 2048 words of 8 channels: (row, col, half)
            const int e = tid + q * NT, c8 = e & 1, px = (e >> 1) & 31, r = e >> 6;
            *(rt_u32x4*)(s_in + ((r + 1) * P32 + px + 1) * S16 + c8 * 8) = xin[q];
        }
        __syncthreads();

        // ================= block2.conv (16 -> 32 @32x32) + MaxPool2d(3,2,1), four bands of 4 pooled rows -> s_x (16x16x32, haloed)
        for (int band = 0; band < 4; ++band) {
            const int cy0 = 8 * band - 1;                   // first conv row of the band (may be -1: outside the image)
            constexpr int MTB = (18 + NW - 1) / NW;         // band tiles per wave (3 at 8 waves, 2 at 16)
            const int nmtb = (18 - __builtin_amdgcn_readfirstlane(wave) + NW - 1) / NW;
            f32x4 acc[MTB][2];
            int abase[MTB], cbase[MTB];
            bool dead[MTB];
#pragma unroll
            for (int mt = 0; mt < MTB; ++mt) {
                int t = wave + NW * mt;
                const bool live = t < 18;
                t = live ? t : 17;
                const int pl = t * 16 + i, y = pl >> 5, x = pl & 31, cy = cy0 + y;
                dead[mt] = cy < 0;
                abase[mt] = ((cy < 0 ? 0 : cy) * P32 + x) * S16;           // window origin of conv pixel (cy, x) in the haloed image
                cbase[mt] = live ? (y * 33 + x + 1) * SCS + kq * 4 : -1;
            }
            rt_conv<5, WS16, false, MTB, KoffTab>(s_in, s_w[0], koff2, abase, bb16, acc, nmtb);
#pragma unroll
            for (int mt = 0; mt < MTB; ++mt) {
                if (cbase[mt] < 0) continue;
#pragma unroll
                for (int nb = 0; nb < 2; ++nb) {
                    const f32x4 bq = *(const f32x4*)(s_b + nb * 16 + kq * 4);
                    const unsigned k0 = mi_bf16x2_to_keys(mi_pk_bf16(acc[mt][nb][0] + bq[0], acc[mt][nb][1] + bq[1]));
                    const unsigned k1 = mi_bf16x2_to_keys(mi_pk_bf16(acc[mt][nb][2] + bq[2], acc[mt][nb][3] + bq[3]));
                    *(uint2*)(s_y + cbase[mt] + nb * 16) = (uint2){dead[mt] ? MI_KEY_MIN2 : k0, dead[mt] ? MI_KEY_MIN2 : k1};
                }
            }
            if (band == 0) { bank_commit(1); bank_fetch(3); }
            __syncthreads();
            if (tid < 256) {                                // (pooled row 0..3, pooled col, 8-channel group)
                const int c8 = tid & 3, ox = (tid >> 2) & 15, oyl = tid >> 6;
                uint4 u[9];
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) u[ky * 3 + kx] = *(const uint4*)(s_y + ((2 * oyl + ky) * 33 + 2 * ox + kx) * SCS + c8 * 8);
                uint4 pk; uint2 ar;
                mi_pool9_keys(u, pk, ar);
                *(uint4*)(s_x + ((4 * band + oyl + 1) * P16 + ox + 1) * S32 + c8 * 8) = pk;
            }
            __syncthreads();
        }
        for (int e = tid; e < X16_ELEMS / 8; e += NT) ((uint4*)s_y)[e] = (uint4){0u, 0u, 0u, 0u};       // the key band becomes the second residual image
        __syncthreads();

        // ================= block2 res1, res2 @16x16: conv k = 1..4 (two pixel tiles per wave)
        constexpr int MT16 = 16 / NW;                       // 16x16 pixel tiles per wave (2 at 8 waves, 1 at 16)
        {
            int abase[MT16], cen[MT16];
#pragma unroll
            for (int mt = 0; mt < MT16; ++mt) { const int pl = (wave + NW * mt) * 16 + i, y = pl >> 4, x = pl & 15; abase[mt] = (y * P16 + x) * S32 + kq * 8; cen[mt] = (y * P16 + x) * S32 + (P16 + 1) * S32 + kq * 4; }
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                const int k1 = 1 + 2 * st, k2 = k1 + 1;
                f32x4 acc[MT16][2];
                rt_conv<9, WS32, true, MT16, KoffC<P16>>(s_x, s_w[k1 & 1], KoffC<P16>(), abase, bb32, acc);
#pragma unroll
                for (int mt = 0; mt < MT16; ++mt)
#pragma unroll
                    for (int nb = 0; nb < 2; ++nb) {
                        const f32x4 bq = *(const f32x4*)(s_b + k1 * 32 + nb * 16 + kq * 4);
                        float v[4];
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = acc[mt][nb][r] + bq[r];
                        const uint2 raw = rt_pack(v);
                        *(uint2*)(s_y + cen[mt] + nb * 16) = (uint2){rt_relu2(raw.x), rt_relu2(raw.y)};
                    }
                bank_step(k1);
                __syncthreads();
                rt_conv<9, WS32, false, MT16, KoffC<P16>>(s_y, s_w[k2 & 1], KoffC<P16>(), abase, bb32, acc);
#pragma unroll
                for (int mt = 0; mt < MT16; ++mt)
#pragma unroll
                    for (int nb = 0; nb < 2; ++nb) {
                        const f32x4 bq = *(const f32x4*)(s_b + k2 * 32 + nb * 16 + kq * 4);
                        const uint2 sk = *(const uint2*)(s_x + cen[mt] + nb * 16);
                        float v[4];
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = acc[mt][nb][r] + bq[r] + rt_lane(sk, r);
                        *(uint2*)(s_x + cen[mt] + nb * 16) = rt_pack(v);           // raw: the next conv applies the ReLU on its operand reads
                    }
                bank_step(k2);
                __syncthreads();
            }
        }

        // ================= block3.conv (32 -> 32 @16x16, no ReLU on its input) + pool -> 8x8 image in the s_y region
        unsigned short* s_k3 = s_in;                        // key image [17 rows: conv row -1 .. 15][1 pad + 16 cells][SCS]
        unsigned short* s_x8 = s_y;
        unsigned short* s_y8 = s_x;
        {
            for (int e = tid; e < 17 * 16; e += NT) {       // conv row -1 (17 cells) and the pad column of rows 0..15: minimal keys
                const int cell = e >> 4, d = e & 15;
                ((unsigned*)s_k3)[cell * (SCS / 2) + d] = MI_KEY_MIN2;
                if (cell >= 1) ((unsigned*)s_k3)[cell * 17 * (SCS / 2) + d] = MI_KEY_MIN2;
            }
            for (int e = tid; e < X8_ELEMS / 8; e += NT) ((uint4*)s_x8)[e] = (uint4){0u, 0u, 0u, 0u};
            int abase[MT16], cbase[MT16];
#pragma unroll
            for (int mt = 0; mt < MT16; ++mt) { const int pl = (wave + NW * mt) * 16 + i, y = pl >> 4, x = pl & 15; abase[mt] = (y * P16 + x) * S32 + kq * 8; cbase[mt] = ((y + 1) * 17 + x + 1) * SCS + kq * 4; }
            f32x4 acc[MT16][2];
            rt_conv<9, WS32, false, MT16, KoffC<P16>>(s_x, s_w[5 & 1], KoffC<P16>(), abase, bb32, acc);
#pragma unroll
            for (int mt = 0; mt < MT16; ++mt)
#pragma unroll
                for (int nb = 0; nb < 2; ++nb) {
                    const f32x4 bq = *(const f32x4*)(s_b + 5 * 32 + nb * 16 + kq * 4);
                    const unsigned k0 = mi_bf16x2_to_keys(mi_pk_bf16(acc[mt][nb][0] + bq[0], acc[mt][nb][1] + bq[1]));
                    const unsigned k1 = mi_bf16x2_to_keys(mi_pk_bf16(acc[mt][nb][2] + bq[2], acc[mt][nb][3] + bq[3]));
                    *(uint2*)(s_k3 + cbase[mt] + nb * 16) = (uint2){k0, k1};
                }
            bank_step(5);
            __syncthreads();
            for (int e = tid; e < X8_ELEMS / 8; e += NT) ((uint4*)s_y8)[e] = (uint4){0u, 0u, 0u, 0u};       // s_x (16x16) is dead from here
            if (tid < 256) {
                const int c8 = tid & 3, ox = (tid >> 2) & 7, oy = tid >> 5;
                uint4 u[9];
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) u[ky * 3 + kx] = *(const uint4*)(s_k3 + ((2 * oy + ky) * 17 + 2 * ox + kx) * SCS + c8 * 8);
                uint4 pk; uint2 ar;
                mi_pool9_keys(u, pk, ar);
                *(uint4*)(s_x8 + ((oy + 1) * P8 + ox + 1) * S32 + c8 * 8) = pk;
            }
            __syncthreads();
        }

        // ================= block3 res1, res2 @8x8: conv k = 6..9, one (pixel tile, output block) task per wave
        {
            const bool on = wave < 8;                       // 4 pixel tiles x 2 output blocks = 8 tasks; further waves only move banks
            const int tile = (wave & 7) >> 1, nb = wave & 1;
            const int pl = tile * 16 + i, y = pl >> 3, x = pl & 7;
            const int abase = (y * P8 + x) * S32 + kq * 8, cen = (y * P8 + x) * S32 + (P8 + 1) * S32 + kq * 4 + nb * 16;
            const int bbn = bb32 + nb * 16 * WS32;
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                const int k1 = 6 + 2 * st, k2 = k1 + 1;
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                if (on) acc = rt_conv1<true>(s_x8, s_w[k1 & 1], abase, bbn);
                if (on) {
                    const f32x4 bq = *(const f32x4*)(s_b + k1 * 32 + nb * 16 + kq * 4);
                    float v[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = acc[r] + bq[r];
                    const uint2 raw = rt_pack(v);
                    *(uint2*)(s_y8 + cen) = (uint2){rt_relu2(raw.x), rt_relu2(raw.y)};
                }
                bank_step(k1);
                __syncthreads();
                if (on) acc = rt_conv1<false>(s_y8, s_w[k2 & 1], abase, bbn);
                if (on) {
                    const f32x4 bq = *(const f32x4*)(s_b + k2 * 32 + nb * 16 + kq * 4);
                    const uint2 sk = *(const uint2*)(s_x8 + cen);
                    float v[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = acc[r] + bq[r] + rt_lane(sk, r);
                    const uint2 raw = rt_pack(v);
                    if (k2 < 9) *(uint2*)(s_x8 + cen) = raw;
                    else *(uint2*)(a.y + ((long long)img * 64 + pl) * 32 + nb * 16 + kq * 4) = raw;
                }
                if (k2 < 9) bank_step(k2);
                __syncthreads();
            }
        }
    }
}

// x: block1 output [n][32][32][16] bf16; y: block3 output [n][8][8][32] bf16; bank / bias: the ten convs of blocks 2 and 3 in network order
void launch_rollout_tail_bf16(const void* x, void* y, int n, const unsigned short* const* bank, const float* const* bias, hipStream_t st) {
    if (n <= 0) return;
    static std::once_flag attr;          // (launchers run on up to 4 group worker threads)
    std::call_once(attr, [] { hipFuncSetAttribute((const void*)rollout_tail_bf16_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, rt::LDS_ELEMS * 2); });
    RolloutTailArgs a{};
    a.x = (const unsigned short*)x; a.y = (unsigned short*)y; a.n = n;
    for (int k = 0; k < 10; ++k) { a.bank[k] = bank[k]; a.bias[k] = bias[k]; }
    hipLaunchKernelGGL(rollout_tail_bf16_kernel, dim3(n < 256 ? n : 256), dim3(rt::NT), rt::LDS_ELEMS * 2, st, a);
}
