// First conv of an IMPALA block fused with the block's MaxPool2d(3, 2, 1) for the bf16 mode (block2.conv 16 -> 32 @32x32,
// block3.conv 32 -> 32 @16x16; block1.conv has its own uint8-input kernel in conv_bf16.hip) -- common/model.py:150-163.
//
// forward : a work item produces 4 pooled rows of one image from 9 conv rows (one halo row recomputed), which live in
//           LDS only: the conv output, the largest tensor of the block, is neither written to nor re-read from HBM; the
//           backward pass needs the pooled arg-max, not the conv output.
// backward: the gradient of the conv output is never materialised either.  PoolStage (conv_bf16.hip) rebuilds any window of it
//           in LDS from (pooled gradient, arg-max bytes) for the conv's weight-gradient and data-gradient kernels.
#include "common.h"
#include <mutex>
#include <math.h>

typedef short bf16x8 __attribute__((ext_vector_type(8)));
#define MFMA_BF16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16((a), (b), (c), 0, 0, 0)
__device__ __forceinline__ unsigned short cp_f2bf(float x) { __bf16 h = (__bf16)x; return __builtin_bit_cast(unsigned short, h); }

template <int CIN_, int COUT_, int HW_>
struct CpCfg {
    static constexpr int CIN = CIN_, COUT = COUT_, HW = HW_, HO = HW / 2;
    static constexpr int S = (CIN == 16) ? 16 : 48;                  // staged-pixel stride (conflict-free, see conv_bf16.hip)
    static constexpr int CR = 9, PH = CR + 2, PW = HW + 2;           // conv rows per item, staged input rows / cols
    static constexpr int IN_ELEMS = ((PH * PW * S + 7) / 8) * 8;
    static constexpr int NK = (CIN == 32) ? 9 : 5, WS = NK * 32 + 16, W_ELEMS = COUT * WS;
    static constexpr int SCS = COUT + COUT / 2;                      // conv-output tile pixel stride (elements): 1.5 pixels -> the pooling
                                                                     // reads (lanes 2 pixels apart) hit every bank once
    static constexpr int SC_ELEMS = CR * (HW + 1) * SCS;             // conv-output tile [row][1 pad + col][channel] of order-preserving keys (common.h);
                                                                     // the pad cell is column -1 of its row and holds minimal keys
    static constexpr int NMT = CR * HW / 16, MT = (NMT + 3) / 4, NB = COUT / 16, C8 = CIN / 8;
#ifndef CP_MTC
#define CP_MTC 3
#endif
#ifndef CP_ROLL
#define CP_ROLL 1         // 0: the item kernel at every batch size (A/B timing)
#endif
    static constexpr int MTC = MT < CP_MTC ? MT : CP_MTC;            // M tiles whose accumulators live together
    static constexpr int NSRC = PH * HW * C8, NLD = (NSRC + 255) / 256;
    static constexpr int IPI = HO / 4;                               // items per image (4 pooled rows each)
    static constexpr int NPOOL = 4 * HO * (COUT / 8);                // pooling tasks per item: (row, col, 8-channel group)
    static constexpr size_t LDS_BYTES = (size_t)(IN_ELEMS + W_ELEMS + SC_ELEMS) * 2;
    static_assert((CR * HW) % 16 == 0 && HO % 4 == 0 && NPOOL <= 256, "tiling");
};

template <class C>
__global__ __launch_bounds__(256, 3) void conv_pool_fwd_bf16_kernel(ConvArgs a, unsigned short* p_out, uint8_t* p_arg) {
    extern __shared__ __attribute__((aligned(16))) unsigned short smem_h[];
    unsigned short* s_in = smem_h;
    unsigned short* s_w = smem_h + C::IN_ELEMS;
    unsigned short* s_c = s_w + C::W_ELEMS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i = lane & 15, kq = lane >> 4;
    const unsigned short* g_in = (const unsigned short*)a.in;
    for (int e = tid; e < C::W_ELEMS / 8; e += 256) ((uint4*)s_w)[e] = ((const uint4*)a.wbank)[e];
    for (int e = tid; e < C::IN_ELEMS / 8; e += 256) ((uint4*)s_in)[e] = (uint4){0u, 0u, 0u, 0u};          // column halos stay zero
    for (int e = tid; e < C::CR * (C::COUT / 2); e += 256) ((unsigned*)s_c)[(e / (C::COUT / 2)) * (C::HW + 1) * (C::SCS / 2) + e % (C::COUT / 2)] = MI_KEY_MIN2;
    float bias_r[C::NB][4];
#pragma unroll
    for (int nb = 0; nb < C::NB; ++nb)
#pragma unroll
        for (int r = 0; r < 4; ++r) bias_r[nb][r] = a.bias ? a.bias[nb * 16 + kq * 4 + r] : 0.f;
    int koff[C::NK];
#pragma unroll
    for (int m = 0; m < C::NK; ++m) {
        int tap, chunk;
        if (C::CIN == 32) { tap = m; chunk = kq; } else { tap = 2 * m + (kq >> 1); chunk = kq & 1; if (tap > 8) tap = 8; }
        koff[m] = ((tap / 3) * C::PW + (tap % 3)) * C::S + chunk * 8;
    }
    const int nwork = a.n * C::IPI;
    uint4 regs[C::NLD];
    // staged row r = image row (first conv row of the item) - 1 + r; the item's first conv row is 2*oy0 - 1
    auto load = [&](int work) {
        const int img = work / C::IPI, gy0 = 2 * (work % C::IPI) * 4 - 2;
#pragma unroll
        for (int k = 0; k < C::NLD; ++k) {
            const int e = tid + k * 256;
            uint4 v = {0u, 0u, 0u, 0u};
            if (e < C::NSRC) {
                const int c8 = e % C::C8, px = (e / C::C8) % C::HW, r = e / (C::C8 * C::HW), gy = gy0 + r;
                if (gy >= 0 && gy < C::HW) v = *(const uint4*)(g_in + (((long long)img * C::HW + gy) * C::HW + px) * C::CIN + c8 * 8);
            }
            regs[k] = v;
        }
    };
    if ((int)blockIdx.x < nwork) load(blockIdx.x);
    for (int work = blockIdx.x; work < nwork; work += gridDim.x) {
        const int img = work / C::IPI, oy0 = (work % C::IPI) * 4, cy0 = 2 * oy0 - 1;      // first conv row of the item (may be -1)
        __syncthreads();
#pragma unroll
        for (int k = 0; k < C::NLD; ++k) {
            const int e = tid + k * 256;
            if (e < C::NSRC) {
                const int c8 = e % C::C8, px = (e / C::C8) % C::HW, r = e / (C::C8 * C::HW);
                *(uint4*)(s_in + (r * C::PW + px + 1) * C::S + c8 * 8) = regs[k];
            }
        }
        __syncthreads();
        if (work + (int)gridDim.x < nwork) load(work + gridDim.x);

        // ---- conv: tile t = wave + 4k covers 16 consecutive pixels of the 9 x HW conv rows -> s_c (bf16, + bias); tiles in groups
        // of MTC (accumulators of one group live at a time: 180 -> fewer registers, 3 waves per SIMD)
#pragma unroll
        for (int mt0 = 0; mt0 < C::MT; mt0 += C::MTC) {
            f32x4 acc[C::MTC][C::NB];
            int abase[C::MTC], cbase[C::MTC];
            bool dead[C::MTC];
#pragma unroll
            for (int mt = 0; mt < C::MTC; ++mt) {
                int t = wave + 4 * (mt0 + mt);
                const bool live = (mt0 + mt < C::MT) && t < C::NMT;
                t = live ? t : C::NMT - 1;
                const int pl = t * 16 + i, y = pl / C::HW, x = pl % C::HW;
                abase[mt] = (y * C::PW + x) * C::S;
                cbase[mt] = live ? (pl + y + 1) * C::SCS + kq * 4 : -1;
                dead[mt] = cy0 < 0 && y == 0;                       // conv row -1 of the image: minimal keys
#pragma unroll
                for (int nb = 0; nb < C::NB; ++nb) acc[mt][nb] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
            const int bbase = i * C::WS + kq * 8;
#pragma unroll
            for (int m = 0; m < C::NK; ++m) {
                bf16x8 av[C::MTC], bv[C::NB];
#pragma unroll
                for (int mt = 0; mt < C::MTC; ++mt) av[mt] = *(const bf16x8*)(s_in + abase[mt] + koff[m]);
#pragma unroll
                for (int nb = 0; nb < C::NB; ++nb) bv[nb] = *(const bf16x8*)(s_w + bbase + nb * 16 * C::WS + m * 32);
#pragma unroll
                for (int mt = 0; mt < C::MTC; ++mt)
                    if (mt0 + mt < C::MT) {
#pragma unroll
                        for (int nb = 0; nb < C::NB; ++nb) acc[mt][nb] = MFMA_BF16(bv[nb], av[mt], acc[mt][nb]);
                    }
            }
#pragma unroll
            for (int mt = 0; mt < C::MTC; ++mt) {
                if (cbase[mt] < 0) continue;
#pragma unroll
                for (int nb = 0; nb < C::NB; ++nb) {
                    const unsigned k0 = mi_bf16x2_to_keys(mi_pk_bf16(acc[mt][nb][0] + bias_r[nb][0], acc[mt][nb][1] + bias_r[nb][1]));
                    const unsigned k1 = mi_bf16x2_to_keys(mi_pk_bf16(acc[mt][nb][2] + bias_r[nb][2], acc[mt][nb][3] + bias_r[nb][3]));
                    *(uint2*)(s_c + cbase[mt] + nb * 16) = (uint2){dead[mt] ? MI_KEY_MIN2 : k0, dead[mt] ? MI_KEY_MIN2 : k1};
                }
            }
        }
        __syncthreads();
        // ---- pooling: thread = (pooled row 0..3, pooled col, 8-channel group): 9 window reads of keys, 4 v_max3_i32 per channel
        // (mi_pool9_keys, common.h); cells outside the image (row -1, column -1) hold minimal keys.
        if (tid < C::NPOOL) {
            constexpr int G8 = C::COUT / 8;
            const int c8 = tid % G8, ox = (tid / G8) % C::HO, oyl = tid / (G8 * C::HO);
            uint4 u[9];
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx)
                    u[ky * 3 + kx] = *(const uint4*)(s_c + ((2 * oyl + ky) * (C::HW + 1) + 2 * ox + kx) * C::SCS + c8 * 8);
            const size_t o = ((((size_t)img * C::HO + oy0 + oyl) * C::HO + ox) * G8 + c8) * 8;
            uint4 pk;
            uint2 ar;
            mi_pool9_keys(u, pk, ar);
            *(uint4*)(p_out + o) = pk;
            *(uint2*)(p_arg + o) = ar;
        }
    }
}

// ---- update-sized batches: WHOLE IMAGES per workgroup, rolling conv rows.  The item kernel above recomputes one conv row per item (9 rows
// for 8 new ones) and its 9 x HW / 16 pixel tiles do not divide over four waves (16 -> 32 @32x32: 18 tiles = 5,5,4,4; 32 -> 32 @16x16:
// 9 tiles = 3,2,2,2 -- a third of the conv phase idle).  Here a workgroup walks down an image in steps of 4 pooled rows = 8 NEW conv rows
// (16 / 8 tiles: 4 / 2 per wave), and the key row under the step's first pooling window is the previous step's last row, carried over in
// LDS; row -1 of an image is the minimal-key row.  10 staged input rows per step instead of 11.  Same arithmetic per conv output and
// per pooling window: results are bit-identical to the item kernel (which stays for rollout-sized batches: more, smaller items).
template <class C>
struct CpRoll {
    static constexpr int NEW = 8, PH = NEW + 2;                      // conv rows computed per step, staged input rows
    static constexpr int IN_ELEMS = ((PH * C::PW * C::S + 7) / 8) * 8;
    static constexpr int NMT = NEW * C::HW / 16, MT = NMT / 4;       // pixel tiles of a step, per wave
    static constexpr int NSRC = PH * C::HW * C::C8, NLD = (NSRC + 255) / 256;
    static constexpr int ROW_WORDS = (C::HW + 1) * C::SCS / 8;       // 16-byte words of one key row
    static constexpr size_t LDS_BYTES = (size_t)(IN_ELEMS + C::W_ELEMS + C::SC_ELEMS) * 2;
    static_assert(NMT % 4 == 0 && MT <= 4 && ((C::HW + 1) * C::SCS) % 8 == 0, "tiling");
};
template <class C>
__global__ __launch_bounds__(256, 3) void conv_pool_fwd_roll_bf16_kernel(ConvArgs a, unsigned short* p_out, uint8_t* p_arg) {
    using R = CpRoll<C>;
    extern __shared__ __attribute__((aligned(16))) unsigned short smem_h[];
    unsigned short* s_in = smem_h;
    unsigned short* s_w = smem_h + R::IN_ELEMS;
    unsigned short* s_c = s_w + C::W_ELEMS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i = lane & 15, kq = lane >> 4;
    const unsigned short* g_in = (const unsigned short*)a.in;
    // the filter bank lives in REGISTERS (10 / 18 fragments per lane, loaded once per launch): a third / half of the conv phase's LDS reads were
    // bank fragments, re-read for every K step of every step of every image
    bf16x8 breg[C::NK][C::NB];
#pragma unroll
    for (int m = 0; m < C::NK; ++m)
#pragma unroll
        for (int nb = 0; nb < C::NB; ++nb) breg[m][nb] = *(const bf16x8*)(a.wbank + (nb * 16 + i) * C::WS + m * 32 + kq * 8);
    for (int e = tid; e < R::IN_ELEMS / 8; e += 256) ((uint4*)s_in)[e] = (uint4){0u, 0u, 0u, 0u};          // column halos stay zero
    for (int e = tid; e < C::CR * (C::COUT / 2); e += 256) ((unsigned*)s_c)[(e / (C::COUT / 2)) * (C::HW + 1) * (C::SCS / 2) + e % (C::COUT / 2)] = MI_KEY_MIN2;
    float bias_r[C::NB][4];
#pragma unroll
    for (int nb = 0; nb < C::NB; ++nb)
#pragma unroll
        for (int r = 0; r < 4; ++r) bias_r[nb][r] = a.bias ? a.bias[nb * 16 + kq * 4 + r] : 0.f;
    int koff[C::NK];
#pragma unroll
    for (int m = 0; m < C::NK; ++m) {
        int tap, chunk;
        if (C::CIN == 32) { tap = m; chunk = kq; } else { tap = 2 * m + (kq >> 1); chunk = kq & 1; if (tap > 8) tap = 8; }
        koff[m] = ((tap / 3) * C::PW + (tap % 3)) * C::S + chunk * 8;
    }
    // this workgroup's images: blockIdx.x, + gridDim.x, ...; step w of the walk = (image w / IPI, pooled rows 4 * (w % IPI) ..)
    const int nimg = ((int)blockIdx.x < a.n) ? (a.n - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0;
    const int nstep = nimg * C::IPI;
    uint4 regs[R::NLD];
    // staged row r = input row 2 * oy0 - 1 + r (the step's first NEW conv row is 2 * oy0)
    auto load = [&](int w) {
        const long long img = blockIdx.x + (long long)(w / C::IPI) * gridDim.x; const int gy0 = 2 * (w % C::IPI) * 4 - 1;
#pragma unroll
        for (int k = 0; k < R::NLD; ++k) {                 // unconditional, from a clamped word / row (zeros are substituted at the LDS store): a conditional
            int e = tid + k * 256; e = e < R::NSRC ? e : R::NSRC - 1;      // prefetch load is waited for where it is issued (DESIGN.md finding 11)
            const int c8 = e % C::C8, px = (e / C::C8) % C::HW, r = e / (C::C8 * C::HW); int gy = gy0 + r;
            gy = gy < 0 ? 0 : (gy > C::HW - 1 ? C::HW - 1 : gy);
            regs[k] = *(const uint4*)(g_in + ((img * C::HW + gy) * C::HW + px) * C::CIN + c8 * 8);
        }
    };
    int abase[R::MT], cbase[R::MT];
#pragma unroll
    for (int mt = 0; mt < R::MT; ++mt) {
        const int pl = (wave + 4 * mt) * 16 + i, y = pl / C::HW, x = pl % C::HW;
        abase[mt] = (y * C::PW + x) * C::S;
        cbase[mt] = ((y + 1) * (C::HW + 1) + x + 1) * C::SCS + kq * 4;          // key row y + 1 (row 0 is the carried row), column x
    }
    if (nstep > 0) load(0);
    for (int w = 0; w < nstep; ++w) {
        const long long img = blockIdx.x + (long long)(w / C::IPI) * gridDim.x; const int sub = w % C::IPI, oy0 = sub * 4;
        __syncthreads();                                   // the previous step's pooling has read its key rows
#pragma unroll
        for (int k = 0; k < R::NLD; ++k) {
            const int e = tid + k * 256;
            if (e < R::NSRC) {
                const int c8 = e % C::C8, px = (e / C::C8) % C::HW, r = e / (C::C8 * C::HW), gy = 2 * oy0 - 1 + r;
                *(uint4*)(s_in + (r * C::PW + px + 1) * C::S + c8 * 8) = (gy >= 0 && gy < C::HW) ? regs[k] : (uint4){0u, 0u, 0u, 0u};
            }
        }
        if (tid < R::ROW_WORDS) {                          // key row 0: conv row 2 * oy0 - 1 = the previous step's last row, or row -1 of the image
            const uint4 mn = {MI_KEY_MIN2, MI_KEY_MIN2, MI_KEY_MIN2, MI_KEY_MIN2};
            ((uint4*)s_c)[tid] = sub ? ((const uint4*)s_c)[8 * R::ROW_WORDS + tid] : mn;
        }
        __syncthreads();
        load(w + 1 < nstep ? w + 1 : w);                   // (past the end: the last step again)

        f32x4 acc[R::MT][C::NB];
#pragma unroll
        for (int mt = 0; mt < R::MT; ++mt)
#pragma unroll
            for (int nb = 0; nb < C::NB; ++nb) acc[mt][nb] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int m = 0; m < C::NK; ++m) {
            bf16x8 av[R::MT];
#pragma unroll
            for (int mt = 0; mt < R::MT; ++mt) av[mt] = *(const bf16x8*)(s_in + abase[mt] + koff[m]);
#pragma unroll
            for (int mt = 0; mt < R::MT; ++mt)
#pragma unroll
                for (int nb = 0; nb < C::NB; ++nb) acc[mt][nb] = MFMA_BF16(breg[m][nb], av[mt], acc[mt][nb]);
        }
#pragma unroll
        for (int mt = 0; mt < R::MT; ++mt)
#pragma unroll
            for (int nb = 0; nb < C::NB; ++nb) {
                const unsigned k0 = mi_bf16x2_to_keys(mi_pk_bf16(acc[mt][nb][0] + bias_r[nb][0], acc[mt][nb][1] + bias_r[nb][1]));
                const unsigned k1 = mi_bf16x2_to_keys(mi_pk_bf16(acc[mt][nb][2] + bias_r[nb][2], acc[mt][nb][3] + bias_r[nb][3]));
                *(uint2*)(s_c + cbase[mt] + nb * 16) = (uint2){k0, k1};
            }
        __syncthreads();
        if (tid < C::NPOOL) {
            constexpr int G8 = C::COUT / 8;
            const int c8 = tid % G8, ox = (tid / G8) % C::HO, oyl = tid / (G8 * C::HO);
            uint4 u[9];
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx)
                    u[ky * 3 + kx] = *(const uint4*)(s_c + ((2 * oyl + ky) * (C::HW + 1) + 2 * ox + kx) * C::SCS + c8 * 8);
            const size_t o = ((((size_t)img * C::HO + oy0 + oyl) * C::HO + ox) * G8 + c8) * 8;
            uint4 pk;
            uint2 ar;
            mi_pool9_keys(u, pk, ar);
            *(uint4*)(p_out + o) = pk;
            *(uint2*)(p_arg + o) = ar;
        }
    }
}

using CP_16_32_32 = CpCfg<16, 32, 32>;
using CP_32_32_16 = CpCfg<32, 32, 16>;

template <class C>
static void launch_cp_t(const ConvArgs& a, void* p_out, uint8_t* p_arg, hipStream_t st) {
    if (a.n >= 1024 && CP_ROLL) {                           // update-sized: whole images per workgroup, rolling conv rows
        using R = CpRoll<C>;
        static std::once_flag attr_r;          // (launchers run on up to 4 group worker threads)
        std::call_once(attr_r, [] { hipFuncSetAttribute((const void*)conv_pool_fwd_roll_bf16_kernel<C>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)R::LDS_BYTES); });
        int bpc = (int)((160 * 1024) / R::LDS_BYTES);
        bpc = bpc < 1 ? 1 : (bpc > 3 ? 3 : bpc);
        const int grid = a.n > 256 * bpc ? 256 * bpc : a.n;
        hipLaunchKernelGGL(conv_pool_fwd_roll_bf16_kernel<C>, dim3(grid), dim3(256), R::LDS_BYTES, st, a, (unsigned short*)p_out, p_arg);
        return;
    }
    static std::once_flag attr;          // (launchers run on up to 4 group worker threads)
    std::call_once(attr, [] { hipFuncSetAttribute((const void*)conv_pool_fwd_bf16_kernel<C>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::LDS_BYTES); });
    int bpc = (int)((160 * 1024) / C::LDS_BYTES);
    bpc = bpc < 1 ? 1 : (bpc > 4 ? 4 : bpc);
    int grid = a.n * C::IPI;
    if (grid > 256 * bpc) grid = 256 * bpc;
    if (grid < 1) return;
    hipLaunchKernelGGL(conv_pool_fwd_bf16_kernel<C>, dim3(grid), dim3(256), C::LDS_BYTES, st, a, (unsigned short*)p_out, p_arg);
}
// a.in (bf16 NHWC), a.wbank (packed forward bank), a.bias, a.n; returns false when the shape has no fused kernel
bool launch_conv_pool_fwd_bf16(ConvShape s, const ConvArgs& a, void* p_out, uint8_t* p_arg, hipStream_t st) {
    if (!a.wbank) return false;
    switch (s) {
        case CS_16_32_32: launch_cp_t<CP_16_32_32>(a, p_out, p_arg, st); return true;
        case CS_32_32_16: launch_cp_t<CP_32_32_16>(a, p_out, p_arg, st); return true;
        default: return false;
    }
}
