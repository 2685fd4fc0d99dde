// Fused residual block forward for the bf16 mode: y = conv2(relu(conv1(relu(x)))) + x   (common/model.py:141-146)
// in ONE launch per block.  A workgroup owns whole images (the 32x32x16, 16x16x32 and 8x8x32 maps of the IMPALA-CNN
// all fit in LDS with their zero halo): relu(x) is staged once, conv1's output goes through the epilogue into a
// second haloed LDS image (ReLU applied), conv2 reads it from there -- the intermediate never makes an HBM round
// trip inside the forward pass (it is still written once when the backward pass will need it), and the rollout
// step (n = n_envs, launch-latency bound) runs 6 launches instead of 12 for its residual blocks.
#include "common.h"

typedef short bf16x8 __attribute__((ext_vector_type(8)));
#define MFMA_BF16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16((a), (b), (c), 0, 0, 0)
__device__ __forceinline__ unsigned short rb_f2bf(float x) { __bf16 h = (__bf16)x; return __builtin_bit_cast(unsigned short, h); }
__device__ __forceinline__ float rb_bf2f(unsigned short h) { return __uint_as_float(((unsigned)h) << 16); }
__device__ __forceinline__ unsigned rb_relu2(unsigned w) { const unsigned neg = (w >> 15) & 0x00010001u; return w & ~(neg * 0xFFFFu); }

template <int C_, int HW_, int NIMG_>
struct RbCfg {
    static constexpr int C = C_, HW = HW_, NIMG = NIMG_;
    static constexpr int S = (C == 16) ? 16 : 48;            // conflict-free pixel strides (see conv_bf16.hip)
    static constexpr int P = HW + 2, NPIX = NIMG * P * P;
    static constexpr int IMG_ELEMS = ((NPIX * S + 7) / 8) * 8;
    static constexpr int NK = (C == 32) ? 9 : 5, WS = NK * 32 + 16, W_ELEMS = C * WS;
    static constexpr int NMT = NIMG * HW * HW / 16, MT = NMT / 4, NB = C / 16, C8 = C / 8;
    static constexpr int MTC = MT < 4 ? MT : 4;              // M tiles computed together (bounds the register use)
    static constexpr int NLD = (NIMG * HW * HW * C8 + 255) / 256;
    static constexpr size_t LDS_BYTES = (size_t)(2 * IMG_ELEMS + 2 * W_ELEMS) * 2;
    static_assert(NMT % 4 == 0 && MT % MTC == 0, "tiling");
};

struct ResblockArgs {
    const unsigned short* x;      // bf16 NHWC [n][HW][HW][C]
    const float *w1, *b1, *w2, *b2;   // fp32 device layout [co][tap][ci]
    unsigned short* a_out;        // conv1 output (pre-ReLU) for the backward pass, or null (rollout)
    unsigned short* y_out;
    int n;
    const unsigned short *bank1, *bank2;   // pre-packed bf16 filter banks (conv_bf16.hip pack_banks_kernel) or null
};

template <class C>
__device__ __forceinline__ void rb_stage_weights(unsigned short* s_w, const float* w) {
    for (int e = threadIdx.x; e < C::C * C::WS; e += 256) {
        const int j = e / C::WS, k = e % C::WS, tap = k / C::C, ci = k % C::C;
        s_w[e] = rb_f2bf((k < C::NK * 32 && tap < 9) ? w[(j * 9 + tap) * C::C + ci] : 0.f);
    }
}

// one 3x3 conv over the haloed LDS image s_src for MTC consecutive M tiles starting at mt0
template <class C>
__device__ __forceinline__ void rb_conv(const unsigned short* s_src, const unsigned short* s_w, const int (&koff)[C::NK], int mt0, int wave,
                                        int i, int kq, f32x4 (&acc)[C::MTC][C::NB]) {
#pragma unroll
    for (int mt = 0; mt < C::MTC; ++mt)
#pragma unroll
        for (int nb = 0; nb < C::NB; ++nb) acc[mt][nb] = (f32x4){0.f, 0.f, 0.f, 0.f};
    int abase[C::MTC];
#pragma unroll
    for (int mt = 0; mt < C::MTC; ++mt) {
        const int pl = (wave * C::MT + mt0 + mt) * 16 + i, y = pl / C::HW, x = pl % C::HW;
        abase[mt] = (((y / C::HW) * C::P + (y % C::HW)) * C::P + x) * C::S;
    }
    const int bbase = i * C::WS + kq * 8;
#pragma unroll
    for (int m = 0; m < C::NK; ++m) {
        bf16x8 av[C::MTC], bv[C::NB];
#pragma unroll
        for (int mt = 0; mt < C::MTC; ++mt) av[mt] = *(const bf16x8*)(s_src + abase[mt] + koff[m]);
#pragma unroll
        for (int nb = 0; nb < C::NB; ++nb) bv[nb] = *(const bf16x8*)(s_w + bbase + nb * 16 * C::WS + m * 32);
#pragma unroll
        for (int mt = 0; mt < C::MTC; ++mt)
#pragma unroll
            for (int nb = 0; nb < C::NB; ++nb) acc[mt][nb] = MFMA_BF16(bv[nb], av[mt], acc[mt][nb]);      // filters as A: 4 channels of one pixel per lane
    }
}

template <class C>
__global__ __launch_bounds__(256) void resblock_bf16_kernel(ResblockArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned short smem_h[];
    unsigned short* s_x = smem_h;                         // relu(x), haloed
    unsigned short* s_y = smem_h + C::IMG_ELEMS;          // relu(conv1 output), haloed
    unsigned short* s_w1 = smem_h + 2 * C::IMG_ELEMS;
    unsigned short* s_w2 = s_w1 + C::W_ELEMS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i = lane & 15, kq = lane >> 4;
    if (a.bank1 && a.bank2) {
        for (int e = tid; e < C::W_ELEMS / 8; e += 256) { ((uint4*)s_w1)[e] = ((const uint4*)a.bank1)[e]; ((uint4*)s_w2)[e] = ((const uint4*)a.bank2)[e]; }
    } else {
        rb_stage_weights<C>(s_w1, a.w1);
        rb_stage_weights<C>(s_w2, a.w2);
    }
    for (int e = tid; e < 2 * C::IMG_ELEMS / 8; e += 256) ((uint4*)smem_h)[e] = (uint4){0u, 0u, 0u, 0u};   // zero halos (interiors are rewritten)
    float b1r[C::NB][4], b2r[C::NB][4];                   // channels nb*16 + 4*kq + r: the accumulator quad of this lane
#pragma unroll
    for (int nb = 0; nb < C::NB; ++nb)
#pragma unroll
        for (int r = 0; r < 4; ++r) { b1r[nb][r] = a.b1[nb * 16 + kq * 4 + r]; b2r[nb][r] = a.b2[nb * 16 + kq * 4 + r]; }
    int koff[C::NK];
#pragma unroll
    for (int m = 0; m < C::NK; ++m) {
        int tap, chunk;
        if (C::C == 32) { tap = m; chunk = kq; } else { tap = 2 * m + (kq >> 1); chunk = kq & 1; if (tap > 8) tap = 8; }
        koff[m] = ((tap / 3) * C::P + (tap % 3)) * C::S + chunk * 8;
    }
    const int nwork = (a.n + C::NIMG - 1) / C::NIMG;
    uint4 regs[C::NLD];
    auto load = [&](int img0) {
#pragma unroll
        for (int k = 0; k < C::NLD; ++k) {
            const int e = tid + k * 256;
            uint4 v = {0u, 0u, 0u, 0u};
            if (e < C::NIMG * C::HW * C::HW * C::C8) {
                const int pix = e / C::C8, c8 = e % C::C8, n = img0 + pix / (C::HW * C::HW);
                if (n < a.n) v = *(const uint4*)(a.x + ((long long)img0 * C::HW * C::HW + pix) * C::C + c8 * 8);
            }
            regs[k] = v;
        }
    };
    if ((int)blockIdx.x < nwork) load(blockIdx.x * C::NIMG);
    for (int work = blockIdx.x; work < nwork; work += gridDim.x) {
        const int img0 = work * C::NIMG;
        __syncthreads();                                   // previous item's LDS reads done (and the zero fill, first time)
#pragma unroll
        for (int k = 0; k < C::NLD; ++k) {
            const int e = tid + k * 256;
            if (e < C::NIMG * C::HW * C::HW * C::C8) {
                const int pix = e / C::C8, c8 = e % C::C8, img = pix / (C::HW * C::HW), q = pix % (C::HW * C::HW);
                uint4 v = regs[k];
                v.x = rb_relu2(v.x); v.y = rb_relu2(v.y); v.z = rb_relu2(v.z); v.w = rb_relu2(v.w);
                *(uint4*)(s_x + ((img * C::P + q / C::HW + 1) * C::P + q % C::HW + 1) * C::S + c8 * 8) = v;
            }
        }
        __syncthreads();
        if (work + (int)gridDim.x < nwork) load((work + gridDim.x) * C::NIMG);

        // ---- conv1 -> a (HBM, optional) and relu(a) -> s_y
#pragma unroll 1
        for (int mt0 = 0; mt0 < C::MT; mt0 += C::MTC) {
            f32x4 acc[C::MTC][C::NB];
            rb_conv<C>(s_x, s_w1, koff, mt0, wave, i, kq, acc);
#pragma unroll
            for (int mt = 0; mt < C::MTC; ++mt) {
                const int pl = (wave * C::MT + mt0 + mt) * 16 + i, y = pl / C::HW, x = pl % C::HW, img = y / C::HW, n = img0 + img;
#pragma unroll
                for (int nb = 0; nb < C::NB; ++nb) {
                    unsigned short h[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) h[r] = rb_f2bf(acc[mt][nb][r] + b1r[nb][r]);
                    const uint2 raw = {(unsigned)h[0] | ((unsigned)h[1] << 16), (unsigned)h[2] | ((unsigned)h[3] << 16)};
                    if (a.a_out && n < a.n) *(uint2*)(a.a_out + ((long long)img0 * C::HW * C::HW + pl) * C::C + nb * 16 + kq * 4) = raw;
                    *(uint2*)(s_y + ((img * C::P + (y % C::HW) + 1) * C::P + x + 1) * C::S + nb * 16 + kq * 4) = (uint2){rb_relu2(raw.x), rb_relu2(raw.y)};
                }
            }
        }
        __syncthreads();
        // ---- conv2 + residual -> y
#pragma unroll 1
        for (int mt0 = 0; mt0 < C::MT; mt0 += C::MTC) {
            uint2 e_res[C::MTC][C::NB];
#pragma unroll
            for (int mt = 0; mt < C::MTC; ++mt) {
                const int pl = (wave * C::MT + mt0 + mt) * 16 + i;
                const int n = img0 + pl / (C::HW * C::HW);
                const long long o = ((long long)(n < a.n ? img0 : 0) * C::HW * C::HW + (n < a.n ? pl : 0)) * C::C + kq * 4;
#pragma unroll
                for (int nb = 0; nb < C::NB; ++nb) e_res[mt][nb] = *(const uint2*)(a.x + o + nb * 16);        // raw x (L2-resident: just staged)
            }
            f32x4 acc[C::MTC][C::NB];
            rb_conv<C>(s_y, s_w2, koff, mt0, wave, i, kq, acc);
#pragma unroll
            for (int mt = 0; mt < C::MTC; ++mt) {
                const int pl = (wave * C::MT + mt0 + mt) * 16 + i, n = img0 + pl / (C::HW * C::HW);
                if (n < a.n) {
#pragma unroll
                    for (int nb = 0; nb < C::NB; ++nb) {
                        const unsigned rw[2] = {e_res[mt][nb].x, e_res[mt][nb].y};
                        unsigned short h[4];
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float rs = (r & 1) ? __uint_as_float(rw[r >> 1] & 0xffff0000u) : __uint_as_float(rw[r >> 1] << 16);
                            h[r] = rb_f2bf(acc[mt][nb][r] + b2r[nb][r] + rs);
                        }
                        *(uint2*)(a.y_out + ((long long)img0 * C::HW * C::HW + pl) * C::C + nb * 16 + kq * 4) =
                            (uint2){(unsigned)h[0] | ((unsigned)h[1] << 16), (unsigned)h[2] | ((unsigned)h[3] << 16)};
                    }
                }
            }
        }
    }
}

using RB_16_32 = RbCfg<16, 32, 1>;
using RB_32_16 = RbCfg<32, 16, 1>;
using RB_32_8  = RbCfg<32,  8, 4>;

template <class C>
static void launch_rb_t(const ResblockArgs& a, hipStream_t st) {
    static bool attr = false;
    if (!attr) { hipFuncSetAttribute((const void*)resblock_bf16_kernel<C>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::LDS_BYTES); attr = true; }
    int bpc = (int)((160 * 1024) / C::LDS_BYTES);
    bpc = bpc < 1 ? 1 : (bpc > 4 ? 4 : bpc);
    int grid = (a.n + C::NIMG - 1) / C::NIMG;
    if (grid > 256 * bpc) grid = 256 * bpc;
    if (grid < 1) return;
    hipLaunchKernelGGL(resblock_bf16_kernel<C>, dim3(grid), dim3(256), C::LDS_BYTES, st, a);
}
// shape = the residual convs' ConvShape (CS_16_16_32 / CS_32_32_16 / CS_32_32_8)
void launch_resblock_bf16(ConvShape s, const void* x, const float* w1, const float* b1, const float* w2, const float* b2, void* a_out,
                          void* y_out, int n, const unsigned short* bank1, const unsigned short* bank2, hipStream_t st) {
    ResblockArgs a{(const unsigned short*)x, w1, b1, w2, b2, (unsigned short*)a_out, (unsigned short*)y_out, n, bank1, bank2};
    switch (s) {
        case CS_16_16_32: launch_rb_t<RB_16_32>(a, st); break;
        case CS_32_32_16: launch_rb_t<RB_32_16>(a, st); break;
        case CS_32_32_8:  launch_rb_t<RB_32_8>(a, st); break;
        default: break;
    }
}
