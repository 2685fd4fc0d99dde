// Fused residual block for the bf16 mode, ONE launch per block and direction (common/model.py:141-146):
//   forward   a = conv1(relu(x)) + b1 ;  y = conv2(relu(a)) + b2 + x
//   backward  da = convT2(dy) * (a > 0) ; dx = convT1(da) * (x > 0) + dy      (the two data gradients)
// Both are "conv -> elementwise -> conv -> elementwise + skip": the first conv's output goes through its epilogue
// into a second haloed LDS image and the second conv reads it from there, so the intermediate makes no HBM round
// trip inside the pass (it is written once: the backward pass / the weight-gradient kernels read it).
//
// Work item = TH output rows of one image (or NIMG whole 8x8 images).  For a row tile the first conv is evaluated on
// TH+2 rows (one halo row each side, recomputed by the neighbour tile) from TH+4 staged input rows; rows outside
// the image are written as zeros -- they are the second conv's zero padding.  Row tiles keep the LDS footprint at
// 2-3 workgroups per CU: with whole 32x32 / 16x16 images (85 / 101 KB) a CU held ONE workgroup and every global
// load of its serial load -> conv -> conv -> store chain was exposed.
// MFMA operand order as in conv_bf16.hip: filter rows = A, pixels = B, so a lane owns 4 consecutive channels of a
// pixel and every global / LDS access of the epilogues is an 8-byte word.
#include "common.h"
#include <hip/hip_ext.h>
#include <mutex>

typedef short bf16x8 __attribute__((ext_vector_type(8)));
#define MFMA_BF16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16((a), (b), (c), 0, 0, 0)
__device__ __forceinline__ unsigned short rb_f2bf(float x) { __bf16 h = (__bf16)x; return __builtin_bit_cast(unsigned short, h); }
__device__ __forceinline__ unsigned rb_relu2(unsigned w) { const unsigned neg = (w >> 15) & 0x00010001u; return w & ~(neg * 0xFFFFu); }
__device__ __forceinline__ float rb_lane(uint2 w, int r) {          // bf16 element r (0..3) of an 8-byte word, widened
    const unsigned u = (r >> 1) ? w.y : w.x;
    return (r & 1) ? __uint_as_float(u & 0xffff0000u) : __uint_as_float(u << 16);
}
__device__ __forceinline__ uint2 rb_pack(const float (&v)[4]) {
    return (uint2){mi_pk_bf16(v[0], v[1]), mi_pk_bf16(v[2], v[3])};
}

#ifndef RB_ROWPAD8
#define RB_ROWPAD8 1
#endif
#ifndef RB_SWZ16
#define RB_SWZ16 1       // 16-channel LDS images: chunk swap keyed on bit 2 of the column (0 = plain layout, for A/B timing)
#endif
#ifndef RB_S32
#define RB_S32 48         // pixel stride (bf16 elements) of the 32-channel LDS tiles
#endif
template <int C_, int HW_, int TH_, int NIMG_, int NT_ = 256>
struct RbCfg {
    static constexpr int C = C_, HW = HW_, TH = TH_, NIMG = NIMG_, NT = NT_, NW = NT_ / 64;      // NT threads = NW waves per workgroup
    static constexpr bool WHOLE = (TH == HW);                // whole images: no halo rows to recompute
    static_assert(WHOLE || NIMG == 1, "row tiles hold one image");
    static_assert(HW % TH == 0, "tiles cover the image");
    static constexpr int S = (C == 16) ? 16 : RB_S32;        // conflict-free pixel strides (see conv_bf16.hip)
    static constexpr int P = HW + 2;                         // haloed row length
    static constexpr int R1 = WHOLE ? HW : TH + 2;           // rows the first conv is evaluated on (per image)
    static constexpr int XR = R1 + 2;                        // staged input rows
    static constexpr int YR = TH + 2;                        // rows of the intermediate image (conv2's haloed input)
    // row pitch of the haloed LDS images.  8x8 images: a 16-pixel MFMA tile is TWO rows of 8, and at 10 x 96 bytes per row the pixels (y, x)
    // and (y + 1, x) sit 48 banks apart -- on the same bank pairs as their row-mates 8 and 2 columns on: every 8-byte epilogue / skip access
    // 2-way (39 % of the pair kernel's LDS cycles).  16 bytes of padding per row move row y + 1 onto the other half of every 8-bank group.
    static constexpr int RP = P * S + ((HW == 8 && RB_ROWPAD8) ? 8 : 0);
    static constexpr int X_ELEMS = ((NIMG * XR * RP + 7) / 8) * 8, Y_ELEMS = ((NIMG * YR * RP + 7) / 8) * 8;
    static constexpr int NK = (C == 32) ? 9 : 5, WS = NK * 32 + 16, W_ELEMS = C * WS;
    static constexpr int NB = C / 16, C8 = C / 8;
    static constexpr int NMT1 = NIMG * R1 * HW / 16, NMT2 = NIMG * TH * HW / 16;     // M tiles (16 pixels) of the two convs
    static constexpr int MT1 = (NMT1 + NW - 1) / NW, MT2 = (NMT2 + NW - 1) / NW;   // per wave (tile t = wave + NW*k)
    static constexpr int pick(int mt) { return mt % 4 == 0 ? 4 : mt % 3 == 0 ? 3 : mt % 2 == 0 ? 2 : 1; }
    static constexpr int MTC1 = pick(MT1), MTC2 = pick(MT2);   // tiles computed together (bounds the register use)
    static constexpr int NSRC = NIMG * XR * HW * C8, NLD = (NSRC + NT - 1) / NT;     // 16-byte words staged per item
    static constexpr int TPI = HW / TH;
    static constexpr size_t LDS_BYTES = (size_t)(X_ELEMS + Y_ELEMS + 2 * W_ELEMS) * 2;
    static_assert((NIMG * R1 * HW) % 16 == 0 && (NIMG * TH * HW) % 16 == 0, "whole M tiles");
};

struct ResblockArgs {
    const unsigned short* x;      // forward: block input; backward: dy          bf16 NHWC [n][HW][HW][C]
    const float *b1, *b2;         // forward biases (null in the backward instantiation)
    unsigned short* a_out;        // forward: conv1 output (pre-ReLU) or null (rollout); backward: d(conv1 output)
    unsigned short* y_out;        // forward: block output; backward: d(block input)
    int n;
    const unsigned short *bank1, *bank2;   // packed bf16 filter banks of the first / second conv of the PASS (conv_bf16.hip pack_banks_kernel)
    const unsigned short *m1, *m2;         // backward: ReLU mask sources of the first / second conv's output (conv1 forward output, block input)
};

// one 3x3 conv over a haloed LDS image for up to MTC M tiles (pixel tiles whose LDS base offsets are given)
typedef short rb_s16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned rb_relu2_max(unsigned w) {      // ReLU of two packed bf16 as a signed 16-bit max with 0 (one v_pk_max_i16)
    return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(rb_s16x2, w), (rb_s16x2){0, 0}));
}
template <class C, int MTC, bool RELU_A = false>
__device__ __forceinline__ void rb_conv(const unsigned short* s_src, const unsigned short* s_w, const int (&koff)[C::NK], const int (&abase)[MTC],
                                        int i, int kq, f32x4 (&acc)[MTC][C::NB]) {
#pragma unroll
    for (int mt = 0; mt < MTC; ++mt)
#pragma unroll
        for (int nb = 0; nb < C::NB; ++nb) acc[mt][nb] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int bbase = i * C::WS + kq * 8;
#pragma unroll
    for (int m = 0; m < C::NK; ++m) {
        bf16x8 av[MTC], bv[C::NB];
#pragma unroll
        for (int mt = 0; mt < MTC; ++mt) {
            av[mt] = *(const bf16x8*)(s_src + abase[mt] + koff[m]);
            if (RELU_A) {                                   // the image holds raw values (they also serve as the skip connection)
                const uint4 u = __builtin_bit_cast(uint4, av[mt]);
                av[mt] = __builtin_bit_cast(bf16x8, (uint4){rb_relu2_max(u.x), rb_relu2_max(u.y), rb_relu2_max(u.z), rb_relu2_max(u.w)});
            }
        }
#pragma unroll
        for (int nb = 0; nb < C::NB; ++nb) bv[nb] = *(const bf16x8*)(s_w + bbase + nb * 16 * C::WS + m * 32);
#pragma unroll
        for (int mt = 0; mt < MTC; ++mt)
#pragma unroll
            for (int nb = 0; nb < C::NB; ++nb) acc[mt][nb] = MFMA_BF16(bv[nb], av[mt], acc[mt][nb]);
    }
}

template <class C, bool BWD>
__global__ __launch_bounds__(C::NT) void resblock_bf16_kernel(ResblockArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned short smem_h[];
    unsigned short* s_x = smem_h;                         // staged input (ReLU applied in the forward pass), haloed
    unsigned short* s_y = smem_h + C::X_ELEMS;            // first conv's output after its epilogue, haloed
    unsigned short* s_w1 = s_y + C::Y_ELEMS;
    unsigned short* s_w2 = s_w1 + C::W_ELEMS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i = lane & 15, kq = lane >> 4;
    for (int e = tid; e < C::W_ELEMS / 8; e += C::NT) { ((uint4*)s_w1)[e] = ((const uint4*)a.bank1)[e]; ((uint4*)s_w2)[e] = ((const uint4*)a.bank2)[e]; }
    for (int e = tid; e < (C::X_ELEMS + C::Y_ELEMS) / 8; e += C::NT) ((uint4*)smem_h)[e] = (uint4){0u, 0u, 0u, 0u};   // column halos stay zero
    float b1r[C::NB][4], b2r[C::NB][4];                   // channels nb*16 + 4*kq + r: the accumulator quad of this lane
#pragma unroll
    for (int nb = 0; nb < C::NB; ++nb)
#pragma unroll
        for (int r = 0; r < 4; ++r) { b1r[nb][r] = BWD ? 0.f : a.b1[nb * 16 + kq * 4 + r]; b2r[nb][r] = BWD ? 0.f : a.b2[nb * 16 + kq * 4 + r]; }
    int koff[C::NK];
#pragma unroll
    for (int m = 0; m < C::NK; ++m) {
        int tap, chunk;
        if (C::C == 32) { tap = m; chunk = kq; } else { tap = 2 * m + (kq >> 1); chunk = kq & 1; if (tap > 8) tap = 8; }
        koff[m] = ((tap / 3) * C::P + (tap % 3)) * C::S + chunk * 8;
    }
    const int nwork = C::WHOLE ? (a.n + C::NIMG - 1) / C::NIMG : a.n * C::TPI;
    auto item = [&](int work, int& img0, int& ty0) {
        if (C::WHOLE) { img0 = work * C::NIMG; ty0 = 0; } else { img0 = work / C::TPI; ty0 = (work % C::TPI) * C::TH; }
    };
    // staged row r of image slot `img` is image row gy0 + r
    uint4 regs[C::NLD];
    auto load = [&](int img0, int ty0) {
        const int gy0 = C::WHOLE ? -1 : ty0 - 2;
#pragma unroll
        for (int k = 0; k < C::NLD; ++k) {
            const int e = tid + k * C::NT;
            uint4 v = {0u, 0u, 0u, 0u};
            if (e < C::NSRC) {
                const int c8 = e % C::C8, px = (e / C::C8) % C::HW, r = (e / (C::C8 * C::HW)) % C::XR, img = e / (C::C8 * C::HW * C::XR);
                const int gy = gy0 + r, n = img0 + img;
                if (n < a.n && gy >= 0 && gy < C::HW) v = *(const uint4*)(a.x + (((long long)n * C::HW + gy) * C::HW + px) * C::C + c8 * 8);
            }
            regs[k] = v;
        }
    };
    int img0, ty0;
    if ((int)blockIdx.x < nwork) { item(blockIdx.x, img0, ty0); load(img0, ty0); }
    for (int work = blockIdx.x; work < nwork; work += gridDim.x) {
        item(work, img0, ty0);
        __syncthreads();                                   // previous item's LDS reads done (and the zero fill, first time)
#pragma unroll
        for (int k = 0; k < C::NLD; ++k) {
            const int e = tid + k * C::NT;
            if (e < C::NSRC) {
                const int c8 = e % C::C8, px = (e / C::C8) % C::HW, rr = e / (C::C8 * C::HW);      // rr = img * XR + r
                uint4 v = regs[k];
                if (!BWD) { v.x = rb_relu2(v.x); v.y = rb_relu2(v.y); v.z = rb_relu2(v.z); v.w = rb_relu2(v.w); }
                *(uint4*)(s_x + (rr * C::P + px + 1) * C::S + c8 * 8) = v;
            }
        }
        __syncthreads();
        if (work + (int)gridDim.x < nwork) { int i2, y2; item(work + gridDim.x, i2, y2); load(i2, y2); }

        // ---- first conv: rows gy1 + ry (ry < R1) -> global (owned rows only) and the intermediate LDS image
        const int gy1 = C::WHOLE ? 0 : ty0 - 1;
#pragma unroll 1
        for (int k0 = 0; k0 < C::MT1; k0 += C::MTC1) {
            int abase[C::MTC1], ybase[C::MTC1];
            long long goff[C::MTC1];
            bool inimg[C::MTC1], owned[C::MTC1];
#pragma unroll
            for (int mt = 0; mt < C::MTC1; ++mt) {
                int t = wave + C::NW * (k0 + mt);
                const bool live = t < C::NMT1;
                t = live ? t : C::NMT1 - 1;
                const int pl = t * 16 + i, px = pl % C::HW, ry = (pl / C::HW) % C::R1, img = pl / (C::HW * C::R1);
                const int gy = gy1 + ry, n = img0 + img;
                abase[mt] = ((img * C::XR + ry) * C::P + px) * C::S;
                ybase[mt] = ((img * C::YR + ry + (C::WHOLE ? 1 : 0)) * C::P + px + 1) * C::S + kq * 4;
                inimg[mt] = live && gy >= 0 && gy < C::HW && n < a.n;
                owned[mt] = inimg[mt] && (C::WHOLE || (ry >= 1 && ry <= C::TH));
                goff[mt] = inimg[mt] ? (((long long)n * C::HW + gy) * C::HW + px) * C::C + kq * 4 : (long long)kq * 4;
                if (!live) ybase[mt] = -1;
            }
            uint2 e_m[C::MTC1][C::NB];                       // backward: mask source, requested before the MFMAs
            if (BWD) {
#pragma unroll
                for (int mt = 0; mt < C::MTC1; ++mt)
#pragma unroll
                    for (int nb = 0; nb < C::NB; ++nb) e_m[mt][nb] = *(const uint2*)(a.m1 + goff[mt] + nb * 16);
            }
            f32x4 acc[C::MTC1][C::NB];
            rb_conv<C, C::MTC1>(s_x, s_w1, koff, abase, i, kq, acc);
#pragma unroll
            for (int mt = 0; mt < C::MTC1; ++mt) {
                if (ybase[mt] < 0) continue;
#pragma unroll
                for (int nb = 0; nb < C::NB; ++nb) {
                    float v[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        v[r] = acc[mt][nb][r] + b1r[nb][r];
                        if (BWD) v[r] = rb_lane(e_m[mt][nb], r) > 0.f ? v[r] : 0.f;
                    }
                    uint2 raw = rb_pack(v);
                    if (!inimg[mt]) raw = (uint2){0u, 0u};                          // outside the image: the second conv's zero padding
                    if (a.a_out && owned[mt]) *(uint2*)(a.a_out + goff[mt] + nb * 16) = raw;
                    *(uint2*)(s_y + ybase[mt] + nb * 16) = BWD ? raw : (uint2){rb_relu2(raw.x), rb_relu2(raw.y)};
                }
            }
        }
        __syncthreads();
        // ---- second conv + skip connection: rows ty0 + oy (oy < TH)
#pragma unroll 1
        for (int k0 = 0; k0 < C::MT2; k0 += C::MTC2) {
            int abase[C::MTC2];
            long long goff[C::MTC2];
            bool on[C::MTC2];
#pragma unroll
            for (int mt = 0; mt < C::MTC2; ++mt) {
                int t = wave + C::NW * (k0 + mt);
                const bool live = t < C::NMT2;
                t = live ? t : C::NMT2 - 1;
                const int pl = t * 16 + i, px = pl % C::HW, oy = (pl / C::HW) % C::TH, img = pl / (C::HW * C::TH);
                const int n = img0 + img;
                abase[mt] = ((img * C::YR + oy) * C::P + px) * C::S;
                on[mt] = live && n < a.n;
                goff[mt] = on[mt] ? (((long long)n * C::HW + ty0 + oy) * C::HW + px) * C::C + kq * 4 : (long long)kq * 4;
            }
            uint2 e_res[C::MTC2][C::NB], e_m[C::MTC2][C::NB];
#pragma unroll
            for (int mt = 0; mt < C::MTC2; ++mt)
#pragma unroll
                for (int nb = 0; nb < C::NB; ++nb) {
                    e_res[mt][nb] = *(const uint2*)(a.x + goff[mt] + nb * 16);          // raw x / dy (staged moments ago; taking dy from its
                                                                                        // LDS tile instead measured no faster)
                    if (BWD) e_m[mt][nb] = *(const uint2*)(a.m2 + goff[mt] + nb * 16);
                }
            f32x4 acc[C::MTC2][C::NB];
            rb_conv<C, C::MTC2>(s_y, s_w2, koff, abase, i, kq, acc);
#pragma unroll
            for (int mt = 0; mt < C::MTC2; ++mt) {
                if (!on[mt]) continue;
#pragma unroll
                for (int nb = 0; nb < C::NB; ++nb) {
                    float v[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        v[r] = acc[mt][nb][r] + b2r[nb][r];
                        if (BWD) v[r] = rb_lane(e_m[mt][nb], r) > 0.f ? v[r] : 0.f;
                        v[r] += rb_lane(e_res[mt][nb], r);
                    }
                    *(uint2*)(a.y_out + goff[mt] + nb * 16) = rb_pack(v);
                }
            }
        }
    }
}

//                      C  HW  TH NIMG
#ifndef RB16_CFG
#define RB16_CFG 32, 1, 1024
#endif
using RB_16_32 = RbCfg<16, 32, RB16_CFG>;   // whole image (85 KB: one workgroup per CU) with 1024 threads = 4 waves per SIMD: no halo rows;
                                            // measured 7.6 ms per iteration vs 8.9 for 16-row tiles x 256 threads x 3 per CU, 8.0 for 512 threads
#ifndef RB32_NT
#define RB32_NT 512
#endif
using RB_32_16 = RbCfg<32, 16, 16, 1, RB32_NT>; // whole image, 101 KB: ONE workgroup per CU, so it is 512 threads (2 waves per SIMD); 8-row
                                            // tiles (77 KB, 2 x 256 threads per CU, 25 % halo recompute) measured slower: 7.7 vs 6.3 ms
using RB_32_8  = RbCfg<32,  8,  8, 2>;      // 70 KB: 2 per CU
using RB_32_8P = RbCfg<32,  8,  8, 4, 512>; // pair kernel (four banks): 155 KB, one 512-thread workgroup per CU = 2 waves per SIMD (2 images x 256 threads: 116 KB, ONE wave per SIMD)
using RB_32_8S = RbCfg<32,  8,  8, 1>;      // rollout-sized batches: one image per workgroup (more workgroups, less serial work each)

template <class C, bool BWD = false>
static void launch_rb_t(const ResblockArgs& a, hipStream_t st) {
    static std::once_flag attr;          // (launchers run on up to 4 group worker threads)
    std::call_once(attr, [] { hipFuncSetAttribute((const void*)resblock_bf16_kernel<C, BWD>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::LDS_BYTES); });
    int bpc = (int)((160 * 1024) / C::LDS_BYTES);
    bpc = bpc < 1 ? 1 : (bpc > 4 ? 4 : bpc);
    int grid = C::WHOLE ? (a.n + C::NIMG - 1) / C::NIMG : a.n * C::TPI;
    if (grid > 256 * bpc) grid = 256 * bpc;
    if (grid < 1) return;
    hipLaunchKernelGGL((resblock_bf16_kernel<C, BWD>), dim3(grid), dim3(C::NT), C::LDS_BYTES, st, a);
}
// shape = the residual convs' ConvShape (CS_16_16_32 / CS_32_32_16 / CS_32_32_8); bank1 / bank2 = forward banks of conv1 / conv2
void launch_resblock_bf16(ConvShape s, const void* x, const float* b1, const float* b2, void* a_out, void* y_out, int n,
                          const unsigned short* bank1, const unsigned short* bank2, hipStream_t st) {
    ResblockArgs a{(const unsigned short*)x, b1, b2, (unsigned short*)a_out, (unsigned short*)y_out, n, bank1, bank2, nullptr, nullptr};
    switch (s) {
        case CS_16_16_32: launch_rb_t<RB_16_32>(a, st); break;
        case CS_32_32_16: launch_rb_t<RB_32_16>(a, st); break;
        case CS_32_32_8:  if (n <= 1024) launch_rb_t<RB_32_8S>(a, st); else launch_rb_t<RB_32_8>(a, st); break;
        default: break;
    }
}

// Data gradients of a residual block in one launch:
//   dA = convT2(dy) * (A > 0)      -> da_out        (A = conv1's forward output; the weight-gradient kernels read dA)
//   dx = convT1(dA) * (x > 0) + dy -> dx_out        (x = the block input)
// bank2_t / bank1_t: the transposed ("dgrad") filter banks of conv2 / conv1.
void launch_resblock_bwd_bf16(ConvShape s, const void* dy, const void* a_fwd, const void* x_fwd, void* da_out, void* dx_out, int n,
                              const unsigned short* bank2_t, const unsigned short* bank1_t, hipStream_t st) {
    ResblockArgs a{(const unsigned short*)dy, nullptr, nullptr, (unsigned short*)da_out, (unsigned short*)dx_out, n,
                   bank2_t, bank1_t, (const unsigned short*)a_fwd, (const unsigned short*)x_fwd};
    switch (s) {
        case CS_16_16_32: launch_rb_t<RB_16_32, true>(a, st); break;
        case CS_32_32_16: launch_rb_t<RB_32_16, true>(a, st); break;
        case CS_32_32_8:  launch_rb_t<RB_32_8, true>(a, st); break;
        default: break;
    }
}

// ------------------------------------------------------------------------------------------ both residual blocks of a stage, forward
// res1 and res2 of an IMPALA block in ONE launch (whole-image configurations only): the output of res1 becomes res2's
// input through LDS, which holds RAW values: conv1 applies the ReLU on its operand reads (one v_pk_max_i16 per dword) and
// the skip connection is read back from the same tile, so x is fetched from HBM once and res1's output is never read back; the rollout step runs 3 launches for its 6 residual blocks.  Four filter banks stay in LDS.
struct ResblockPairArgs {
    const unsigned short* x;                 // block input (pooled map) bf16 NHWC
    const float* b[4];                       // biases of res1.conv1, res1.conv2, res2.conv1, res2.conv2
    unsigned short *a1_out, *y1_out, *a2_out, *y2_out;     // conv1 outputs (null in the rollout), res1 output (null in the rollout), res2 output
    int n;
    const unsigned short* bank[4];
};

template <class C>
__global__ __launch_bounds__(C::NT) void resblock_pair_bf16_kernel(ResblockPairArgs a) {
    static_assert(C::WHOLE && C::MT1 == C::MTC1 && C::MT2 == C::MTC2, "whole images, one tile group per conv");
    extern __shared__ __attribute__((aligned(16))) unsigned short smem_h[];
    unsigned short* s_x = smem_h;
    unsigned short* s_y = smem_h + C::X_ELEMS;
    unsigned short* s_w = s_y + C::Y_ELEMS;              // 4 banks
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i = lane & 15, kq = lane >> 4;
    // 16-channel images (32-byte pixels): the two 16-byte chunks of a pixel trade places where bit 2 of its haloed column is set.  The
    // 8-byte epilogue stores of 16 pixels x one channel quad (ds_write_b64: 16 consecutive lanes per LDS cycle, 32 banks) hit every
    // bank from 4 pixels (columns c, c+4, c+8, c+12) in the plain layout and from 2 with the swap; the 16-byte operand reads stay
    // conflict-free (their 16-lane groups pair columns c and c+8 from different chunks, and the swap is the same for both).
    constexpr bool SWZ = (C::C == 16) && RB_SWZ16;
    const int eq = SWZ ? ((kq * 4) ^ ((((i + 1) >> 2) & 1) * 8)) : kq * 4;      // this lane's channel quad inside its pixel (tile columns start at 0 / 16)
    for (int e = tid; e < C::W_ELEMS / 8; e += C::NT)
#pragma unroll
        for (int l = 0; l < 4; ++l) ((uint4*)(s_w + l * C::W_ELEMS))[e] = ((const uint4*)a.bank[l])[e];
    for (int e = tid; e < (C::X_ELEMS + C::Y_ELEMS) / 8; e += C::NT) ((uint4*)smem_h)[e] = (uint4){0u, 0u, 0u, 0u};   // halos stay zero
    float* s_b = (float*)(s_w + 4 * C::W_ELEMS);          // the four bias vectors (kept out of the registers: 1024-thread variant is at the 128 limit)
    for (int e = tid; e < 4 * C::C; e += C::NT) s_b[e] = a.b[e / C::C][e % C::C];
    int koff[C::NK];
#pragma unroll
    for (int m = 0; m < C::NK; ++m) {
        int tap, chunk;
        if (C::C == 32) { tap = m; chunk = kq; } else { tap = 2 * m + (kq >> 1); chunk = kq & 1; if (tap > 8) tap = 8; }
        if (SWZ) chunk ^= ((i + tap % 3) >> 2) & 1;        // column of this lane's pixel under the tap (tiles start at columns 0 / 16)
        koff[m] = (tap / 3) * C::RP + (tap % 3) * C::S + chunk * 8;
    }
    const int nwork = (a.n + C::NIMG - 1) / C::NIMG;
    uint4 regs[C::NLD];
    auto load = [&](int img0) {
#pragma unroll
        for (int k = 0; k < C::NLD; ++k) {
            const int e = tid + k * C::NT;
            uint4 v = {0u, 0u, 0u, 0u};
            if (e < C::NSRC) {
                const int c8 = e % C::C8, px = (e / C::C8) % C::HW, r = (e / (C::C8 * C::HW)) % C::XR, img = e / (C::C8 * C::HW * C::XR);
                const int gy = r - 1, n = img0 + img;
                if (n < a.n && gy >= 0 && gy < C::HW) v = *(const uint4*)(a.x + (((long long)n * C::HW + gy) * C::HW + px) * C::C + c8 * 8);
            }
            regs[k] = v;
        }
    };
    // tile geometry: whole images, so both convs of both stages use the same pixel tiles (tile k of this wave = wave + NW*k) and the
    // haloed images s_x / s_y have the same shape: one LDS base + one global offset per tile serve everything
    static_assert(C::NMT1 == C::NMT2 && C::XR == C::YR, "whole-image geometry");
    constexpr int CENTER = C::RP + C::S;                 // from a tile pixel's window origin to the pixel itself
    int abase[C::MT1], poff[C::MT1];                     // window origin in the haloed image; element offset of the pixel's channel quad in the item
    bool live[C::MT1];
#pragma unroll
    for (int mt = 0; mt < C::MT1; ++mt) {
        int t = wave + C::NW * mt; live[mt] = t < C::NMT1; t = live[mt] ? t : C::NMT1 - 1;
        const int pl = t * 16 + i, px = pl % C::HW, ry = (pl / C::HW) % C::HW, img = pl / (C::HW * C::HW);
        abase[mt] = (img * C::XR + ry) * C::RP + px * C::S;
        poff[mt] = pl * C::C + kq * 4;
    }
    if ((int)blockIdx.x < nwork) load(blockIdx.x * C::NIMG);
    for (int work = blockIdx.x; work < nwork; work += gridDim.x) {
        const int img0 = work * C::NIMG;
        const long long base = (long long)img0 * C::HW * C::HW * C::C;
        const int left = a.n - img0;                        // images of this item that exist
        __syncthreads();
#pragma unroll
        for (int k = 0; k < C::NLD; ++k) {
            const int e = tid + k * C::NT;
            if (e < C::NSRC) {
                const int c8 = e % C::C8, px = (e / C::C8) % C::HW, rr = e / (C::C8 * C::HW);
                const uint4 v = regs[k];
                *(uint4*)(s_x + rr * C::RP + (px + 1) * C::S + (SWZ ? (c8 ^ (((px + 1) >> 2) & 1)) : c8) * 8) = v;      // RAW: conv1 applies the ReLU on its operand reads, the skip reads it back
            }
        }
        __syncthreads();
        if (work + (int)gridDim.x < nwork) load((work + gridDim.x) * C::NIMG);
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            unsigned short* a_out = st ? a.a2_out : a.a1_out;
            unsigned short* y_out = st ? a.y2_out : a.y1_out;
            // ---- conv1 (+ bias) -> a (HBM, optional) and relu(a) -> s_y
            {
                f32x4 acc[C::MT1][C::NB];
                rb_conv<C, C::MT1, true>(s_x, s_w + (2 * st) * C::W_ELEMS, koff, abase, i, kq, acc);
#pragma unroll
                for (int mt = 0; mt < C::MT1; ++mt) {
                    if (!live[mt]) continue;
                    const bool on = (poff[mt] / (C::HW * C::HW * C::C)) < left;
#pragma unroll
                    for (int nb = 0; nb < C::NB; ++nb) {
                        float v[4];
                        const f32x4 bq = *(const f32x4*)(s_b + (2 * st) * C::C + nb * 16 + kq * 4);
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = acc[mt][nb][r] + bq[r];
                        const uint2 raw = rb_pack(v);
                        if (a_out && on) *(uint2*)(a_out + base + poff[mt] + nb * 16) = raw;
                        *(uint2*)(s_y + abase[mt] + CENTER + eq + nb * 16) = (uint2){rb_relu2(raw.x), rb_relu2(raw.y)};
                    }
                }
            }
            __syncthreads();
            // ---- conv2 (+ bias) + skip -> y (HBM; res1's output only when a backward pass follows) and, after res1, relu(y) -> s_x
            {
                f32x4 acc[C::MT2][C::NB];
                rb_conv<C, C::MT2>(s_y, s_w + (2 * st + 1) * C::W_ELEMS, koff, abase, i, kq, acc);
#pragma unroll
                for (int mt = 0; mt < C::MT2; ++mt) {
                    if (!live[mt]) continue;
                    const bool on = (poff[mt] / (C::HW * C::HW * C::C)) < left;
#pragma unroll
                    for (int nb = 0; nb < C::NB; ++nb) {
                        float v[4];
                        const f32x4 bq = *(const f32x4*)(s_b + (2 * st + 1) * C::C + nb * 16 + kq * 4);
                        const uint2 sk = *(const uint2*)(s_x + abase[mt] + CENTER + eq + nb * 16);      // skip connection: this stage's raw input
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = acc[mt][nb][r] + bq[r] + rb_lane(sk, r);
                        const uint2 raw = rb_pack(v);
                        if (y_out && on) *(uint2*)(y_out + base + poff[mt] + nb * 16) = raw;
                        if (st == 0) *(uint2*)(s_x + abase[mt] + CENTER + eq + nb * 16) = raw;      // res2's input (raw: conv operand ReLU'd on read, skip as is)
                    }
                }
            }
            if (st == 0) __syncthreads();
        }
    }
}

// ---- 32 channels @16x16, update-sized batches: the two residual blocks as two wave ROLES, pipelined over images (round 3).
// resblock_pair_bf16_kernel keeps its four filter banks in LDS (78 KB) and, with two pixel tiles per wave, reads a bank fragment from
// LDS for every two MFMAs: 1 KB of LDS traffic per MFMA, the LDS pipe 0.66 busy, the matrix pipe 0.40 (187 us per 8192 images).  Here
// waves 0-3 run res1 of image k + 1 while waves 4-7 run res2 of image k, each role with ITS two banks in registers (36 fragments, loaded
// once per launch): no bank reads at all, four pixel tiles per wave and conv, and the LDS holds five 16x16x32 tiles instead -- res1's
// input, relu(res1.conv1), res1's output twice (the hand-over between the roles, double-buffered), relu(res2.conv1).  Two barriers per
// image (after the staging, between the two convs), shared by the roles.  Arithmetic per output element as before (same banks, same K
// order, same rounding points): the four stored tensors are bit-identical.
template <int HW_, int NIMG_>
struct RbPair32R {                                       // HW 16: one image per step; HW 8: two images per step (8 pixel tiles: two per wave and conv)
    static constexpr int C = 32, HW = HW_, NIMG = NIMG_, P = HW + 2, S = RB_S32, IMG_ELEMS = P * P * S, T_ELEMS = NIMG * IMG_ELEMS, WS = 9 * 32 + 16;
    static constexpr int NPX = NIMG * HW * HW, NMT = NPX / 16, NWORD = NPX * 4 / 256;       // pixels, pixel tiles, staged 16-byte words per thread of role A
    static constexpr size_t LDS_BYTES = (size_t)5 * T_ELEMS * 2;
    static_assert(LDS_BYTES <= 160 * 1024 && NMT % 8 == 0 && NPX * 4 % 256 == 0, "one workgroup per CU; tile pairs per wave; whole staging words");
    // pixel pl of the step -> element offset of its window origin in a tile set
    static __device__ __forceinline__ int org(int pl) { return (pl / (HW * HW)) * IMG_ELEMS + (((pl / HW) % HW) * P + pl % HW) * S; }
};
template <class C>
__global__ __launch_bounds__(512, 2) void resblock_pair32r_bf16_kernel(ResblockPairArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned short smem_h[];
    unsigned short* s_x = smem_h;                         // res1 input (raw: ReLU on the operand reads, the skip connection reads it as is)
    unsigned short* s_a1 = s_x + C::T_ELEMS;              // relu(res1.conv1 output)
    unsigned short* s_h = s_a1 + C::T_ELEMS;              // res1 output (raw) [2]: written by role A for image k, read by role B one step later
    unsigned short* s_a2 = s_h + 2 * C::T_ELEMS;          // relu(res2.conv1 output)
    const int tid = threadIdx.x, lane = tid & 63, i = lane & 15, kq = lane >> 4;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool role_a = wv < 4;
    const int rw = wv & 3;
    for (int e = tid; e < 5 * C::T_ELEMS / 8; e += 512) ((uint4*)smem_h)[e] = (uint4){0u, 0u, 0u, 0u};      // halos stay zero
    // this role's two banks: st[2m + nb] = (tap m, output block nb) of its conv1, st[18 + 2m + nb] of its conv2; its two bias quads per output block
    f32x4 st[36];
    f32x4 bq[2][2];
    {
        const unsigned short* bk1 = a.bank[role_a ? 0 : 2];
        const unsigned short* bk2 = a.bank[role_a ? 1 : 3];
        const float* b1 = a.b[role_a ? 0 : 2];
        const float* b2 = a.b[role_a ? 1 : 3];
#pragma unroll
        for (int m = 0; m < 9; ++m)
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) {
                st[2 * m + nb] = __builtin_bit_cast(f32x4, *(const uint4*)(bk1 + (nb * 16 + i) * C::WS + m * 32 + kq * 8));
                st[18 + 2 * m + nb] = __builtin_bit_cast(f32x4, *(const uint4*)(bk2 + (nb * 16 + i) * C::WS + m * 32 + kq * 8));
            }
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) { bq[0][nb] = *(const f32x4*)(b1 + nb * 16 + kq * 4); bq[1][nb] = *(const f32x4*)(b2 + nb * 16 + kq * 4); }
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);                   // vmcnt(0)
    auto koffc = [](int m) { return ((m / 3) * C::P + (m % 3)) * C::S; };
    constexpr int CENTER = (C::P + 1) * C::S;
    // items (NIMG images) of this workgroup: blockIdx.x, + gridDim.x, ...; role A works on item `step`, role B on item `step - 1`
    const int nitem = (a.n + C::NIMG - 1) / C::NIMG;
    const int nimg = ((int)blockIdx.x < nitem) ? (nitem - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0;
    typedef unsigned rp_u32x4 __attribute__((ext_vector_type(4)));
    rp_u32x4 rx[C::NWORD];                                // role A: the next item's input (pixels x 4 chunks / 256 threads words per thread)
    auto load = [&](int k) {
        const long long it = blockIdx.x + (long long)(k < nimg ? k : nimg - 1) * gridDim.x;        // past the end: the last item again (unconditional loads)
#pragma unroll
        for (int q = 0; q < C::NWORD; ++q) {
            if constexpr (C::NIMG == 1) rx[q] = *(const rp_u32x4*)(a.x + it * (C::HW * C::HW * C::C) + (size_t)(tid + q * 256) * 8);
            else {
                const int e = tid + q * 256, pl = e >> 2;
                long long n = it * C::NIMG + pl / (C::HW * C::HW); n = n < a.n ? n : a.n - 1;      // an image past the end: a valid one (its outputs are not stored)
                rx[q] = *(const rp_u32x4*)(a.x + (n * (C::HW * C::HW) + pl % (C::HW * C::HW)) * C::C + (e & 3) * 8);
            }
        }
    };
    // one 3x3 conv of this role for TWO pixel tiles at once (four independent accumulator chains, ten operand reads in flight): operands
    // from s_src (ReLU on read if relu), bank fragments st[b0 ..]
    auto conv_tiles = [&](const unsigned short* s_src, const int (&org)[2], const int b0, bool relu, f32x4 (&acc)[2][2]) {
        bf16x8 av[2][5];
        auto rd = [&](int q, int m) {
            bf16x8 v = *(const bf16x8*)(s_src + org[q] + kq * 8 + koffc(m));
            if (relu) { const uint4 u = __builtin_bit_cast(uint4, v); v = __builtin_bit_cast(bf16x8, (uint4){rb_relu2_max(u.x), rb_relu2_max(u.y), rb_relu2_max(u.z), rb_relu2_max(u.w)}); }
            return v;
        };
#pragma unroll
        for (int m = 0; m < 5; ++m) { av[0][m] = rd(0, m); av[1][m] = rd(1, m); }
        asm volatile("" ::: "memory");
#pragma unroll
        for (int q = 0; q < 2; ++q) { acc[q][0] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc[q][1] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int m = 0; m < 9; ++m) {
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const bf16x8 cur = av[q][m % 5];
                acc[q][0] = MFMA_BF16(__builtin_bit_cast(bf16x8, st[b0 + 2 * m]), cur, acc[q][0]);
                acc[q][1] = MFMA_BF16(__builtin_bit_cast(bf16x8, st[b0 + 2 * m + 1]), cur, acc[q][1]);
            }
            if (m < 4) { av[0][m] = rd(0, m + 5); av[1][m] = rd(1, m + 5); }
        }
    };
    if (role_a && nimg > 0) load(0);
    for (int step = 0; step <= nimg; ++step) {
        const bool on = role_a ? step < nimg : step >= 1;                   // (wave-uniform) this role has an image in this step
        const int k = role_a ? step : step - 1;
        const long long item0 = (blockIdx.x + (long long)(k < 0 ? 0 : k) * gridDim.x) * C::NIMG;       // first image of the role's item
        const long long base = item0 * (C::HW * C::HW * C::C);
        const int left = a.n - (int)item0;                                  // images of the item that exist
        unsigned short* s_in = role_a ? s_x : s_h + ((step - 1) & 1) * C::T_ELEMS;      // the role's block input
        unsigned short* s_mid = role_a ? s_a1 : s_a2;
        unsigned short* a_out = role_a ? a.a1_out : a.a2_out;
        unsigned short* y_out = role_a ? a.y1_out : a.y2_out;
        __syncthreads();                                                    // everyone is done with the previous step's tiles
        if (role_a && on) {
#pragma unroll
            for (int q = 0; q < C::NWORD; ++q) {
                const int e = tid + q * 256, pl = e >> 2;
                *(rp_u32x4*)(s_x + C::org(pl) + CENTER + (e & 3) * 8) = rx[q];
            }
            load(step + 1);
        }
        __syncthreads();
        if (on) {
            // ---- conv1 (+ bias) -> a (HBM) and relu(a) -> s_mid: tiles (rw, rw + 4), then (rw + 8, rw + 12)
            for (int t = rw; t < C::NMT; t += 8) {
                int pl[2], org[2];
#pragma unroll
                for (int q2 = 0; q2 < 2; ++q2) { pl[q2] = (t + 4 * q2) * 16 + i; org[q2] = C::org(pl[q2]); }
                f32x4 acc[2][2];
                conv_tiles(s_in, org, 0, true, acc);
#pragma unroll
                for (int q2 = 0; q2 < 2; ++q2)
#pragma unroll
                    for (int nb = 0; nb < 2; ++nb) {
                        float v[4];
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = acc[q2][nb][r] + bq[0][nb][r];
                        const uint2 raw = rb_pack(v);
                        if (a_out && (C::NIMG == 1 || pl[q2] / (C::HW * C::HW) < left)) *(uint2*)(a_out + base + (long long)pl[q2] * C::C + nb * 16 + kq * 4) = raw;
                        *(uint2*)(s_mid + org[q2] + CENTER + nb * 16 + kq * 4) = (uint2){rb_relu2(raw.x), rb_relu2(raw.y)};
                    }
            }
        }
        __syncthreads();
        if (on) {
            // ---- conv2 (+ bias) + skip -> y (HBM); role A also hands it to role B through s_h[step & 1]
            unsigned short* s_nxt = s_h + (step & 1) * C::T_ELEMS;
            for (int t = rw; t < C::NMT; t += 8) {
                int pl[2], org[2];
                uint2 sk[2][2];
#pragma unroll
                for (int q2 = 0; q2 < 2; ++q2) {
                    pl[q2] = (t + 4 * q2) * 16 + i; org[q2] = C::org(pl[q2]);
                    sk[q2][0] = *(const uint2*)(s_in + org[q2] + CENTER + kq * 4); sk[q2][1] = *(const uint2*)(s_in + org[q2] + CENTER + 16 + kq * 4);
                }
                f32x4 acc[2][2];
                conv_tiles(s_mid, org, 18, false, acc);
#pragma unroll
                for (int q2 = 0; q2 < 2; ++q2)
#pragma unroll
                    for (int nb = 0; nb < 2; ++nb) {
                        float v[4];
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = acc[q2][nb][r] + bq[1][nb][r] + rb_lane(sk[q2][nb], r);
                        const uint2 raw = rb_pack(v);
                        if (y_out && (C::NIMG == 1 || pl[q2] / (C::HW * C::HW) < left)) *(uint2*)(y_out + base + (long long)pl[q2] * C::C + nb * 16 + kq * 4) = raw;
                        if (role_a) *(uint2*)(s_nxt + org[q2] + CENTER + nb * 16 + kq * 4) = raw;
                    }
            }
        }
    }
}
#ifndef RB32_PAIR_ROLES
#define RB32_PAIR_ROLES 1          // bit 0: 32 channels @16x16 (n >= 1024), bit 1: @8x8 (n > 1024; two images per step: 55.4 against 56.1 us, not taken) run resblock_pair32r_bf16_kernel instead of resblock_pair_bf16_kernel
#endif
template <class C>
static void launch_rbp32r(const ResblockPairArgs& a, hipStream_t st) {
    static std::once_flag attr;
    std::call_once(attr, [] { hipFuncSetAttribute((const void*)resblock_pair32r_bf16_kernel<C>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::LDS_BYTES); });
    const int items = (a.n + C::NIMG - 1) / C::NIMG, grid = items > 256 ? 256 : items;
    if (grid < 1) return;
    hipLaunchKernelGGL(resblock_pair32r_bf16_kernel<C>, dim3(grid), dim3(512), C::LDS_BYTES, st, a);
}
using RBR_32_16 = RbPair32R<16, 1>;
using RBR_32_8 = RbPair32R<8, 2>;

template <class C>
static void launch_rbp_t(const ResblockPairArgs& a, hipStream_t st) {
    constexpr size_t LDS = (size_t)(C::X_ELEMS + C::Y_ELEMS + 4 * C::W_ELEMS) * 2 + 4 * C::C * 4;
    static std::once_flag attr;          // (launchers run on up to 4 group worker threads)
    std::call_once(attr, [] { hipFuncSetAttribute((const void*)resblock_pair_bf16_kernel<C>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS); });
    int bpc = (int)((160 * 1024) / LDS);
    bpc = bpc < 1 ? 1 : (bpc > 4 ? 4 : bpc);
    int grid = (a.n + C::NIMG - 1) / C::NIMG;
    if (grid > 256 * bpc) grid = 256 * bpc;
    if (grid < 1) return;
    hipLaunchKernelGGL(resblock_pair_bf16_kernel<C>, dim3(grid), dim3(C::NT), LDS, st, a);
}

// ------------------------------------------------------------------------------------------ whole backward of a residual block
// Data gradients AND both weight gradients of a residual block in one launch (16-channel blocks @32x32, the largest
// share of the update's HBM traffic).  Separately, the four kernels move 9.75 tensor passes per block (fused data
// gradients 5.25 + two weight-gradient kernels 2.25 each); here the tile of every operand is staged once:
//   reads  dy (12 rows per 8), conv1 output a and block input x (10 rows per 8, ReLU applied while staging)
//   writes dx only -- the gradient of conv1's output lives in LDS (second transposed conv AND conv1's weight gradient
//   read it there) and goes to HBM only if the caller asks for it.
// Weight gradients as in conv3x3_wgrad_bf16_kernel: M = 16 output channels, N = 16 input channels, K = 32 pixels, both
// operands through ds_read_b64_tr_b16 from the [pixel][channel] tiles, accumulators kept across the persistent loop,
// waves summed through LDS in fixed order, one slab per workgroup and layer; bias gradients = MFMA against ones.
// ReLU masks come from the staged relu(a) / relu(x) tiles (> 0 there <=> > 0 before the ReLU).
typedef short rb_s16x4 __attribute__((ext_vector_type(4)));
typedef rb_s16x4 __attribute__((address_space(3))) * rb_lds_s16x4_ptr;

struct RbFullArgs {
    const unsigned short *dy, *a_fwd, *x_fwd;   // bf16 NHWC [n][32][32][16]
    unsigned short *dx_out, *da_out;            // da_out may be null
    const unsigned short *bank2_t, *bank1_t;    // transposed banks of conv2 / conv1
    float *slab2, *slab1;                       // per-workgroup slabs [grid][2304 + 16] of conv2 / conv1
    int n;
};

#ifndef RBFULL_WG_REUSE
#define RBFULL_WG_REUSE 1          // weight-gradient operand rows shared between a wave's two consecutive pixel rows (see the kernel)
#endif
#ifndef RBFULL_SPREAD_LOADS
#define RBFULL_SPREAD_LOADS 1
#endif
#ifndef RBFULL_CFG
#define RBFULL_CFG 8, 256          // tile rows, threads per workgroup: every wave holds both layers' 20 weight-gradient tiles, 250 registers, 2 waves per SIMD.
#endif                             // Measured alternatives (scratch/kbench_rb16.hip, us per 8192-sample launch on random data, this config 349-365):
                                   //  8 x 512 = two workgroups of 8 waves per CU = 4 waves per SIMD at <= 128 registers, weight-gradient accumulators split by
                                   //  LAYER over the wave halves (waves 0-3 conv2's, 4-7 conv1's), four staging words per thread in one index space: needs 160
                                   //  registers, so 19 spill at 128 -- 378-384; 16 x 512, 92 KB, one workgroup per CU: 14.8 vs 13.8 ms per iteration (round 1)
template <int TH_, int NT_>
struct RbFullT {                                 // C = 16, HW = 32
    static constexpr int C = 16, HW = 32, TH = TH_, NT = NT_, NW = NT_ / 64, S = 16, P = HW + 2, TPI = HW / TH;
    static constexpr int XR = TH + 4, YR = TH + 2;
    static constexpr int X_ELEMS = XR * P * S, Y_ELEMS = YR * P * S;
    static constexpr int NK = 5, WS = NK * 32 + 16, W_ELEMS = C * WS;
    static constexpr int NMT1 = YR * HW / 16, NMT2 = TH * HW / 16, MT1 = (NMT1 + NW - 1) / NW, MT2 = NMT2 / NW;     // TH 8 x 4 waves: 20 / 16 tiles = 5 / 4 per wave
    static_assert(NMT2 % NW == 0, "second conv: whole tiles per wave");
    static constexpr int NX = XR * HW * 2, NA = YR * HW * 2;                                          // 16-byte words staged per tensor
    static constexpr int KX = (NX + NT - 1) / NT, KA = (NA + NT - 1) / NT;
    static constexpr int NSTEP = TH * HW / 32;                                                         // 8 pixel steps of 32
    static constexpr int WLEN = C * 9 * C, SLAB = WLEN + C;
    static constexpr size_t TILE_BYTES = (size_t)(X_ELEMS + 3 * Y_ELEMS + 2 * W_ELEMS) * 2, RED_BYTES = (size_t)(2 * WLEN + 2 * NW * C) * 4;
    static constexpr size_t LDS_BYTES = TILE_BYTES > RED_BYTES ? TILE_BYTES : RED_BYTES;
};
using RbFull = RbFullT<RBFULL_CFG>;

#ifdef RBF_TIMING      // scratch/kbench_rb16.hip: per-phase shader-clock totals of wave 0, summed over workgroups
__device__ unsigned long long g_rbf_timing[8];
#define RBF_TCK(k) do { if (tid == 0) { const long long now_ = clock64(); tacc_[k] += now_ - tlast_; tlast_ = now_; } } while (0)
#else
#define RBF_TCK(k) do { } while (0)
#endif
__global__ __launch_bounds__(RbFull::NT, RbFull::NT == 512 ? 4 : 2) void resblock_bwd_full_bf16_kernel(RbFullArgs a) {      // 512 threads: 4 waves per SIMD, <= 128 registers
    using C = RbFull;
    extern __shared__ __attribute__((aligned(16))) unsigned short smem_h[];
    unsigned short* s_x = smem_h;                         // dy rows ty0-2 .. ty0+TH+1
    unsigned short* s_y = s_x + C::X_ELEMS;               // d(conv1 output) rows ty0-1 .. ty0+TH
    unsigned short* s_a = s_y + C::Y_ELEMS;               // relu(conv1 output), same rows
    unsigned short* s_p = s_a + C::Y_ELEMS;               // relu(block input), same rows
    unsigned short* s_w1 = s_p + C::Y_ELEMS;              // transposed bank of conv2 (first conv of this pass)
    unsigned short* s_w2 = s_w1 + C::W_ELEMS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i = lane & 15, kq = lane >> 4, rq = (lane & 15) >> 2, cp = lane & 3;
    for (int e = tid; e < C::W_ELEMS / 8; e += C::NT) { ((uint4*)s_w1)[e] = ((const uint4*)a.bank2_t)[e]; ((uint4*)s_w2)[e] = ((const uint4*)a.bank1_t)[e]; }
    for (int e = tid; e < (C::X_ELEMS + 3 * C::Y_ELEMS) / 8; e += C::NT) ((uint4*)smem_h)[e] = (uint4){0u, 0u, 0u, 0u};   // column halos stay zero
    int koff[C::NK];
#pragma unroll
    for (int m = 0; m < C::NK; ++m) {
        int tap = 2 * m + (kq >> 1); const int chunk = kq & 1; if (tap > 8) tap = 8;
        koff[m] = ((tap / 3) * C::P + (tap % 3)) * C::S + chunk * 8;
    }
    constexpr bool SPLIT = C::NW == 8;               // weight gradients: waves 0-3 own conv2's tiles (acc2), waves 4-7 conv1's (held in acc2 as well)
    constexpr int NACC1 = 9;                         // (SPLIT: acc1 / accb1 are never used and vanish)
    f32x4 acc2[9], acc1[NACC1], accb2 = {0.f, 0.f, 0.f, 0.f}, accb1 = {0.f, 0.f, 0.f, 0.f};     // weight-gradient tiles of conv2 / conv1, bias rows
#pragma unroll
    for (int t = 0; t < 9; ++t) acc2[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < NACC1; ++t) acc1[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const bf16x8 ones = __builtin_bit_cast(bf16x8, (uint4){0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u});

    const int nwork = a.n * C::TPI;
#ifdef RBF_TIMING
    long long tacc_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast_ = clock64();
#endif
    // 512 threads: the three tiles of an item are 768 + 640 + 640 = 2048 16-byte words = exactly four per thread in ONE index space (a
    // thread's word k is e = tid + 512 k: dy for e < 768, then relu(a), then relu(x)): 16 staging registers instead of 24 with a per-tensor
    // split that leaves the second word of every tensor half empty -- the kernel has 128 registers per thread.
    constexpr bool UNI = C::NT == 512;
    static_assert(!UNI || C::NX + 2 * C::NA == 4 * C::NT, "unified staging: four words per thread");
    uint4 ru[UNI ? 4 : 1];
    auto uni_word = [&](int e, int& t, int& le) { t = e < C::NX ? 0 : (e < C::NX + C::NA ? 1 : 2); le = e - (t == 0 ? 0 : (t == 1 ? C::NX : C::NX + C::NA)); };
    uint4 rx[UNI ? 1 : C::KX], ra[UNI ? 1 : C::KA], rp[UNI ? 1 : C::KA];
    // Next item's tiles into registers while this one is computed.  Every load is UNCONDITIONAL from a row clamped into the image and a
    // word index clamped into the tile (rows outside the image / threads past the tile are replaced by zeros when the registers are
    // stored): `v = 0; if (row in image) v = load` merges the loaded registers with older values and the compiler then waits for the
    // loads right where they are issued -- the item paid a full HBM round trip there (1180 of 9200 cycles per item, scratch/kbench_rb16.hip).
    auto load = [&](int work, int part = 7) {            // part bits: 1 = dy words, 2 = relu(a) words, 4 = relu(x) words
        const long long img = work / C::TPI; const int ty0 = (work % C::TPI) * C::TH;
        if constexpr (UNI) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                int t, le; uni_word(tid + k * C::NT, t, le);
                const int c8 = le & 1, px = (le >> 1) % C::HW; int gy = ty0 - (t == 0 ? 2 : 1) + le / (2 * C::HW);
                gy = gy < 0 ? 0 : (gy > C::HW - 1 ? C::HW - 1 : gy);
                const unsigned short* src = t == 0 ? a.dy : (t == 1 ? a.a_fwd : a.x_fwd);
                ru[k] = *(const uint4*)(src + ((img * C::HW + gy) * C::HW + px) * C::C + c8 * 8);
            }
            return;
        }
        if (part & 1) {
#pragma unroll
        for (int k = 0; k < C::KX; ++k) {
            int e = tid + k * C::NT; e = e < C::NX ? e : C::NX - 1;
            const int c8 = e & 1, px = (e >> 1) % C::HW; int gy = ty0 - 2 + e / (2 * C::HW);
            gy = gy < 0 ? 0 : (gy > C::HW - 1 ? C::HW - 1 : gy);
            rx[k] = *(const uint4*)(a.dy + ((img * C::HW + gy) * C::HW + px) * C::C + c8 * 8);
        }
        }
#pragma unroll
        for (int k = 0; k < C::KA; ++k) {
            int e = tid + k * C::NT; e = e < C::NA ? e : C::NA - 1;
            const int c8 = e & 1, px = (e >> 1) % C::HW; int gy = ty0 - 1 + e / (2 * C::HW);
            gy = gy < 0 ? 0 : (gy > C::HW - 1 ? C::HW - 1 : gy);
            const long long o = ((img * C::HW + gy) * C::HW + px) * C::C + c8 * 8;
            if (part & 2) ra[k] = *(const uint4*)(a.a_fwd + o);
            if (part & 4) rp[k] = *(const uint4*)(a.x_fwd + o);
        }
    };
    auto item = [&](int w) { return w < nwork ? w : nwork - 1; };            // past the end: the last item again (loads stay unconditional)
    if ((int)blockIdx.x < nwork) load(blockIdx.x);
    for (int work = blockIdx.x; work < nwork; work += gridDim.x) {
        const long long img = work / C::TPI; const int ty0 = (work % C::TPI) * C::TH;
        RBF_TCK(0);
        __syncthreads();
        RBF_TCK(1);
        if constexpr (UNI) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                int t, le; uni_word(tid + k * C::NT, t, le);
                const int ry = le / (2 * C::HW), gy = ty0 - (t == 0 ? 2 : 1) + ry;
                const bool in = gy >= 0 && gy < C::HW;
                unsigned short* dst = (t == 0 ? s_x : (t == 1 ? s_a : s_p)) + (ry * C::P + (le >> 1) % C::HW + 1) * C::S + (le & 1) * 8;
                const uint4 v = ru[k];
                const uint4 r = t == 0 ? v : (uint4){rb_relu2(v.x), rb_relu2(v.y), rb_relu2(v.z), rb_relu2(v.w)};
                *(uint4*)dst = in ? r : (uint4){0u, 0u, 0u, 0u};
            }
        } else {
#pragma unroll
        for (int k = 0; k < C::KX; ++k) {
            const int e = tid + k * C::NT, gy = ty0 - 2 + e / (2 * C::HW);
            const bool in = gy >= 0 && gy < C::HW;
            if (e < C::NX) *(uint4*)(s_x + ((e / (2 * C::HW)) * C::P + (e >> 1) % C::HW + 1) * C::S + (e & 1) * 8) = in ? rx[k] : (uint4){0u, 0u, 0u, 0u};
        }
#pragma unroll
        for (int k = 0; k < C::KA; ++k) {
            const int e = tid + k * C::NT, gy = ty0 - 1 + e / (2 * C::HW);
            const bool in = gy >= 0 && gy < C::HW;
            if (e < C::NA) {
                const int o = ((e / (2 * C::HW)) * C::P + (e >> 1) % C::HW + 1) * C::S + (e & 1) * 8;
                *(uint4*)(s_a + o) = in ? (uint4){rb_relu2(ra[k].x), rb_relu2(ra[k].y), rb_relu2(ra[k].z), rb_relu2(ra[k].w)} : (uint4){0u, 0u, 0u, 0u};
                *(uint4*)(s_p + o) = in ? (uint4){rb_relu2(rp[k].x), rb_relu2(rp[k].y), rb_relu2(rp[k].z), rb_relu2(rp[k].w)} : (uint4){0u, 0u, 0u, 0u};
            }
        }
        }
        RBF_TCK(2);
        __syncthreads();
        RBF_TCK(3);
#if RBFULL_SPREAD_LOADS
        // the next item's 9 loads per thread are ISSUED in three pieces, in front of each compute phase: in one piece their issue alone took
        // ~1000 of an item's ~9200 cycles per workgroup (36 KB through a 64 B/clk path shared with the CU's other workgroup) with nothing else running
        const int wnext = item(work + gridDim.x);
        load(wnext, 1);
#else
        load(item(work + gridDim.x));
#endif
        RBF_TCK(4);

        // ---- da = convT2(dy) * (a > 0) on rows ty0-1 .. ty0+TH -> s_y (rows outside the image: relu(a) is 0 there, so da is 0)
        {
            int abase[C::MT1], ybase[C::MT1];
#pragma unroll
            for (int mt = 0; mt < C::MT1; ++mt) {
                int t = wave + C::NW * mt; t = t < C::NMT1 ? t : C::NMT1 - 1;          // (a clamped duplicate rewrites the same values)
                const int pl = t * 16 + i, px = pl % C::HW, ry = pl / C::HW;
                abase[mt] = (ry * C::P + px) * C::S;
                ybase[mt] = (ry * C::P + px + 1) * C::S + kq * 4;
            }
            f32x4 acc[C::MT1][1];
            {
#pragma unroll
                for (int mt = 0; mt < C::MT1; ++mt) acc[mt][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
                const int bbase = i * C::WS + kq * 8;
#pragma unroll
                for (int m = 0; m < C::NK; ++m) {
                    const bf16x8 bv = *(const bf16x8*)(s_w1 + bbase + m * 32);
#pragma unroll
                    for (int mt = 0; mt < C::MT1; ++mt) acc[mt][0] = MFMA_BF16(bv, *(const bf16x8*)(s_x + abase[mt] + koff[m]), acc[mt][0]);
#if RBFULL_SPREAD_LOADS == 2
                    if (m == 1) load(wnext, 2);
#endif
                    if constexpr (SPLIT) __builtin_amdgcn_sched_barrier(0);      // (128 registers: no operand reads hoisted across K steps; 4 waves per SIMD cover the LDS latency)
                }
            }
#pragma unroll
            for (int mt = 0; mt < C::MT1; ++mt) {
                const uint2 mk = *(const uint2*)(s_a + ybase[mt]);
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = rb_lane(mk, r) > 0.f ? acc[mt][0][r] : 0.f;
                const uint2 raw = rb_pack(v);
                *(uint2*)(s_y + ybase[mt]) = raw;
                if (a.da_out) {
                    int t = wave + C::NW * mt; t = t < C::NMT1 ? t : C::NMT1 - 1;
                    const int pl = t * 16 + i, px = pl % C::HW, ry = pl / C::HW, gy = ty0 - 1 + ry;
                    if (ry >= 1 && ry <= C::TH) *(uint2*)(a.da_out + ((img * C::HW + gy) * C::HW + px) * C::C + kq * 4) = raw;
                }
            }
        }
        RBF_TCK(5);
        __syncthreads();
        RBF_TCK(6);
#if RBFULL_SPREAD_LOADS == 2
        load(wnext, 4);
#elif RBFULL_SPREAD_LOADS
        load(wnext, 2);
#endif
        // ---- dx = convT1(da) * (x > 0) + dy on rows ty0 .. ty0+TH-1 -> HBM
        {
            int abase[C::MT2];
#pragma unroll
            for (int mt = 0; mt < C::MT2; ++mt) { const int pl = (wave + C::NW * mt) * 16 + i; abase[mt] = ((pl / C::HW) * C::P + pl % C::HW) * C::S; }
            f32x4 acc[C::MT2];
#pragma unroll
            for (int mt = 0; mt < C::MT2; ++mt) acc[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
            const int bbase = i * C::WS + kq * 8;
#pragma unroll
            for (int m = 0; m < C::NK; ++m) {
                const bf16x8 bv = *(const bf16x8*)(s_w2 + bbase + m * 32);
#pragma unroll
                for (int mt = 0; mt < C::MT2; ++mt) acc[mt] = MFMA_BF16(bv, *(const bf16x8*)(s_y + abase[mt] + koff[m]), acc[mt]);
                if constexpr (SPLIT) __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int mt = 0; mt < C::MT2; ++mt) {
                const int pl = (wave + C::NW * mt) * 16 + i, px = pl % C::HW, oy = pl / C::HW;
                const uint2 mk = *(const uint2*)(s_p + ((oy + 1) * C::P + px + 1) * C::S + kq * 4);
                const uint2 sk = *(const uint2*)(s_x + ((oy + 2) * C::P + px + 1) * C::S + kq * 4);
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = (rb_lane(mk, r) > 0.f ? acc[mt][r] : 0.f) + rb_lane(sk, r);
                *(uint2*)(a.dx_out + ((img * C::HW + ty0 + oy) * C::HW + px) * C::C + kq * 4) = rb_pack(v);
            }
        }
        RBF_TCK(7);
#if RBFULL_SPREAD_LOADS == 1
        load(wnext, 4);
#endif
        // ---- weight gradients: conv2 from (dy, relu(a)), conv1 from (da, relu(x)); pixel steps of 32 dealt to the waves
#if RBFULL_WG_REUSE
        // A pixel step is one image row (HW = 32), and the tap (ky, kx) operand of row r is row r + ky of the staged relu(a) / relu(x)
        // tile shifted by kx: a wave that owns CONSECUTIVE rows r0, r0 + 1 needs tile rows r0 .. r0 + 3 only once each -- 12 operand
        // fragments per layer for its 18 (row, tap) products instead of 18 (56 transposing LDS reads per wave and item instead of 80;
        // the kernel is LDS-pipe / latency bound).  Every accumulator still receives its rows in ascending order.
        static_assert(C::HW == 32 && (C::NSTEP == 2 * C::NW || (SPLIT && C::NSTEP == C::NW)), "one row per pixel step, two consecutive rows per wave (pair)");
        if constexpr (SPLIT) {
            // one code path for both halves (two paths made the compiler keep two accumulator sets): role 0 = conv2 from (dy, relu(a)),
            // role 1 = conv1 from (da, relu(x)); the operand tiles differ by base pointer and by the row offset of the gradient tile
            const int role = wave >> 2, r0 = 2 * (wave & 3);
            const unsigned short* s_d = role ? s_y : s_x;
            const unsigned short* s_b = role ? s_p : s_a;
            const int drow = role ? 1 : 2;
            int ocol[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) ocol[h] = (16 * (kq >> 1) + 8 * h + 4 * (kq & 1) + rq) * C::S + 4 * cp;
            auto tr = [&](const unsigned short* base, int off) {
                const rb_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((rb_lds_s16x4_ptr)(base + ocol[0] + off));
                const rb_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((rb_lds_s16x4_ptr)(base + ocol[1] + off));
                return (bf16x8)__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
            };
            bf16x8 dd[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) dd[j] = tr(s_d, ((r0 + j + drow) * C::P + 1) * C::S);
#pragma unroll
            for (int j = 0; j < 2; ++j) accb2 = MFMA_BF16(dd[j], ones, accb2);
#pragma unroll
            for (int R = 0; R < 4; ++R)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const bf16x8 fb = tr(s_b, ((r0 + R) * C::P + kx) * C::S);
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const int ky = R - j;
                        if (ky < 0 || ky > 2) continue;
                        acc2[ky * 3 + kx] = MFMA_BF16(dd[j], fb, acc2[ky * 3 + kx]);
                    }
                    __builtin_amdgcn_sched_barrier(0);            // (128 registers: at most one operand fragment ahead of its products)
                }
        } else {
            const int r0 = 2 * wave;
            int ocol[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) ocol[h] = (16 * (kq >> 1) + 8 * h + 4 * (kq & 1) + rq) * C::S + 4 * cp;
            auto tr = [&](const unsigned short* base, int off) {
                const rb_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((rb_lds_s16x4_ptr)(base + ocol[0] + off));
                const rb_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((rb_lds_s16x4_ptr)(base + ocol[1] + off));
                return (bf16x8)__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
            };
            bf16x8 d2[2], d1[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                d2[j] = tr(s_x, ((r0 + j + 2) * C::P + 1) * C::S);      // dy at the pixel   (s_x row 0 = ty0-2, col 0 = -1)
                d1[j] = tr(s_y, ((r0 + j + 1) * C::P + 1) * C::S);      // da at the pixel   (s_y row 0 = ty0-1)
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) { accb2 = MFMA_BF16(d2[j], ones, accb2); accb1 = MFMA_BF16(d1[j], ones, accb1); }
#pragma unroll
            for (int R = 0; R < 4; ++R)                                  // tile row r0 + R of s_a / s_p (row 0 = ty0-1, col 0 = -1)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const int off = ((r0 + R) * C::P + kx) * C::S;
                    const bf16x8 fa = tr(s_a, off), fp = tr(s_p, off);
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const int ky = R - j;
                        if (ky < 0 || ky > 2) continue;
                        acc2[ky * 3 + kx] = MFMA_BF16(d2[j], fa, acc2[ky * 3 + kx]);
                        acc1[ky * 3 + kx] = MFMA_BF16(d1[j], fp, acc1[ky * 3 + kx]);
                    }
                }
        }
#else
        for (int t = wave; t < C::NSTEP; t += C::NW) {
            int orow[2];                                   // this lane's two source pixels (interior coordinates), MFMA k permutation as in conv_bf16.hip
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int pl = 32 * t + 16 * (kq >> 1) + 8 * h + 4 * (kq & 1) + rq;
                orow[h] = ((pl / C::HW) * C::P + pl % C::HW) * C::S + 4 * cp;
            }
            auto tr = [&](const unsigned short* base, int off) {
                const rb_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((rb_lds_s16x4_ptr)(base + orow[0] + off));
                const rb_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((rb_lds_s16x4_ptr)(base + orow[1] + off));
                return (bf16x8)__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
            };
            const bf16x8 d2 = tr(s_x, (2 * C::P + 1) * C::S);      // dy at the pixel   (s_x row 0 = ty0-2, col 0 = -1)
            const bf16x8 d1 = tr(s_y, (1 * C::P + 1) * C::S);      // da at the pixel   (s_y row 0 = ty0-1)
            accb2 = MFMA_BF16(d2, ones, accb2);
            accb1 = MFMA_BF16(d1, ones, accb1);
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int toff = ((tap / 3) * C::P + (tap % 3)) * C::S;      // s_a / s_p row 0 = ty0-1, col 0 = -1: tap (0,0) is the pixel's upper-left neighbour
                acc2[tap] = MFMA_BF16(d2, tr(s_a, toff), acc2[tap]);
                acc1[tap] = MFMA_BF16(d1, tr(s_p, toff), acc1[tap]);
            }
        }
#endif
    }
#ifdef RBF_TIMING
    if (tid == 0) for (int k = 0; k < 8; ++k) atomicAdd(&g_rbf_timing[k], (unsigned long long)tacc_[k]);
#endif
    // ---- waves summed through LDS in fixed order; one slab per workgroup and layer
    __syncthreads();
    float* red = (float*)smem_h;                              // [2][WLEN] then [2][NW][16] bias partials
    float* redb = red + 2 * C::WLEN;
    if constexpr (SPLIT) {
        // waves 0-3 hold conv2's tiles, waves 4-7 conv1's (both in acc2 / accb2): the two halves sum side by side, each in wave order
        const int role = wave >> 2, wl = wave & 3;
        for (int w = 0; w < 4; ++w) {
            if (wl == w) {
#pragma unroll
                for (int tap = 0; tap < 9; ++tap)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int o = role * C::WLEN + ((kq * 4 + r) * 9 + tap) * C::C + i;
                        red[o] = (w == 0) ? acc2[tap][r] : red[o] + acc2[tap][r];
                    }
            }
            __syncthreads();
        }
        if (i == 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) redb[(role * 4 + wl) * 16 + kq * 4 + r] = accb2[r];
        }
        __syncthreads();
        float* sl2 = a.slab2 + (long long)blockIdx.x * C::SLAB;
        float* sl1 = a.slab1 + (long long)blockIdx.x * C::SLAB;
        for (int e = tid; e < C::WLEN; e += C::NT) { sl2[e] = red[e]; sl1[e] = red[C::WLEN + e]; }
        if (tid < 16) {
            float t2 = 0.f, t1 = 0.f;
            for (int w = 0; w < 4; ++w) { t2 += redb[w * 16 + tid]; t1 += redb[(4 + w) * 16 + tid]; }
            sl2[C::WLEN + tid] = t2; sl1[C::WLEN + tid] = t1;
        }
    } else {
    for (int w = 0; w < C::NW; ++w) {
        if (wave == w) {
#pragma unroll
            for (int tap = 0; tap < 9; ++tap)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int o = ((kq * 4 + r) * 9 + tap) * C::C + i;
                    red[o] = (w == 0) ? acc2[tap][r] : red[o] + acc2[tap][r];
                    red[C::WLEN + o] = (w == 0) ? acc1[tap][r] : red[C::WLEN + o] + acc1[tap][r];
                }
        }
        __syncthreads();
    }
    if (i == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) { redb[wave * 16 + kq * 4 + r] = accb2[r]; redb[C::NW * 16 + wave * 16 + kq * 4 + r] = accb1[r]; }
    }
    __syncthreads();
    float* sl2 = a.slab2 + (long long)blockIdx.x * C::SLAB;
    float* sl1 = a.slab1 + (long long)blockIdx.x * C::SLAB;
    for (int e = tid; e < C::WLEN; e += C::NT) { sl2[e] = red[e]; sl1[e] = red[C::WLEN + e]; }
    if (tid < 16) {
        float t2 = 0.f, t1 = 0.f;
        for (int w = 0; w < C::NW; ++w) { t2 += redb[w * 16 + tid]; t1 += redb[C::NW * 16 + w * 16 + tid]; }
        sl2[C::WLEN + tid] = t2; sl1[C::WLEN + tid] = t1;
    }
    }
}
// ---- wave-specialised variant of the kernel above (16 channels @32x32, 512 threads, ONE workgroup per CU).
// scratch/kbench_rb16.hip phase clocks of the kernel above: conv phases A + B = 54 % of an item's cycles with the matrix pipe 28 % and the
// LDS pipe 48 % busy -- every wave walks load-issue -> LDS stores -> conv A -> conv B -> weight gradients one after the other, two
// waves per SIMD, each phase waiting on its own LDS / MFMA latencies.  Here, as in resblock_bwd_full32s_bf16_kernel, the eight waves take two
// ROLES and each SIMD hosts one wave of each, so the matrix work of one role overlaps the LDS work of the other:
//   waves 0-3  the two transposed convs, BOTH filter banks in registers (2 x 5 fragments = 40 VGPRs, loaded once per launch: no weight
//              reads from LDS, no bank copies in LDS); a wave's operand reads of a whole phase go out before its first MFMA;
//   waves 4-7  global loads + LDS staging of the tiles (four 16-byte words per thread and tensor row group, one index space), conv2's
//              weight gradient while the conv waves produce da, conv1's while they produce dx -- two CONSECUTIVE pixel rows per wave and
//              quarter tile, operand rows shared between them.
// Arithmetic per output element is unchanged (same K order): dx / da are bit-identical to the kernel above; weight-gradient slabs sum
// the same products (rows dealt to other waves).
template <int TH_>
struct RbFull16ST {
    static constexpr int C = 16, HW = 32, TH = TH_, NT = 512, S = 16, P = HW + 2, TPI = HW / TH;
    static constexpr int XR = TH + 4, YR = TH + 2;
    static constexpr int X_ELEMS = XR * P * S, Y_ELEMS = YR * P * S;
    static constexpr int NK = 5, WS = NK * 32 + 16;
    static constexpr int NMT1 = YR * HW / 16, NMT2 = TH * HW / 16, MT1 = (NMT1 + 3) / 4, MT2 = NMT2 / 4;       // tiles per conv wave
    static constexpr int NX = XR * HW * 2, NA = YR * HW * 2, NWORD = NX + 2 * NA, KW = (NWORD + 255) / 256;    // 16-byte words; per staging thread
    static constexpr int NR = TH / 4;                                                                       // pixel rows per weight-gradient wave
    static constexpr int WLEN = C * 9 * C, SLAB = WLEN + C;
    static constexpr size_t TILE_BYTES = (size_t)(X_ELEMS + 3 * Y_ELEMS) * 2, RED_BYTES = (size_t)(2 * WLEN + 2 * 4 * C) * 4;
    static constexpr size_t LDS_BYTES = TILE_BYTES > RED_BYTES ? TILE_BYTES : RED_BYTES;
    static_assert(NMT2 % 4 == 0 && TH % 4 == 0, "whole tiles / rows per wave");
};
#ifndef RBFULL16_SPECIALISED
#define RBFULL16_SPECIALISED 2     // (see the end of this section)
#endif
#ifndef RBFULL16S_TH
#define RBFULL16S_TH (RBFULL16_SPECIALISED == 2 ? 16 : 8)      // tile rows of the two-role kernels: 16 with the LDS-DMA fill (two items per image: rows
#endif                                                         // re-read for the halo 1.17x instead of 1.33x; 2 x 60 KB of DMA-filled tiles + 19 KB)
#ifndef RBFULL16S_CG
#define RBFULL16S_CG 2
#endif
using RbFull16S = RbFull16ST<RBFULL16S_TH>;
#ifdef RBF_TIMING
#define S16_TCK(k) do { if ((tid & 255) == 0) { const long long now_ = clock64(); tacc_[k] += now_ - tlast_; tlast_ = now_; } } while (0)
#else
#define S16_TCK(k) do { } while (0)
#endif
#if RBFULL16_SPECIALISED == 1

__global__ __launch_bounds__(512, 2) void resblock_bwd_full16s_bf16_kernel(RbFullArgs a) {
    using C = RbFull16S;
    extern __shared__ __attribute__((aligned(16))) unsigned short smem_h[];
    unsigned short* s_x = smem_h;                         // dy rows ty0-2 .. ty0+TH+1
    unsigned short* s_y = s_x + C::X_ELEMS;               // d(conv1 output) rows ty0-1 .. ty0+TH
    unsigned short* s_a = s_y + C::Y_ELEMS;               // relu(conv1 output), same rows
    unsigned short* s_p = s_a + C::Y_ELEMS;               // relu(block input), same rows
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i = lane & 15, kq = lane >> 4, rq = (lane & 15) >> 2, cp = lane & 3;
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    const bool conv_role = wv < 4;
    const int rw = wv & 3;
#ifdef RBF_TIMING
    long long tacc_[4] = {0, 0, 0, 0}, tlast_ = clock64();
#endif
    for (int e = tid; e < (C::X_ELEMS + 3 * C::Y_ELEMS) / 8; e += C::NT) ((uint4*)smem_h)[e] = (uint4){0u, 0u, 0u, 0u};   // column halos stay zero
    // ONE register array for both roles (two sets: the kernel's allocation is the union of what its waves may keep live).
    // conv role: st[m] = fragment of K step m of conv2's transposed bank, st[5 + m] of conv1's.
    // weight-gradient role: st[tap] / st[10 + tap] accumulator tiles of conv2 / conv1, st[9] / st[19] their bias rows, st[20 ..] the prefetch words.
    constexpr int NST = 20 + C::KW;
    f32x4 st[NST];
    if (conv_role) {
#pragma unroll
        for (int m = 0; m < C::NK; ++m) {
            st[m] = __builtin_bit_cast(f32x4, *(const uint4*)(a.bank2_t + i * C::WS + m * 32 + kq * 8));
            st[5 + m] = __builtin_bit_cast(f32x4, *(const uint4*)(a.bank1_t + i * C::WS + m * 32 + kq * 8));
        }
    } else {
#pragma unroll
        for (int q = 0; q < NST; ++q) st[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);                   // vmcnt(0): the bank fragments are in (see resblock_bwd_full32s_bf16_kernel)
    const bf16x8 ones = __builtin_bit_cast(bf16x8, (uint4){0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u});
    int koff[C::NK];
#pragma unroll
    for (int m = 0; m < C::NK; ++m) {
        int tap = 2 * m + (kq >> 1); const int chunk = kq & 1; if (tap > 8) tap = 8;
        koff[m] = ((tap / 3) * C::P + (tap % 3)) * C::S + chunk * 8;
    }
    constexpr int CG = RBFULL16S_CG;                       // conv tiles whose operand reads are in flight together
    const int nwork = a.n * C::TPI;
    const int t2 = tid - 256;                              // staging thread index (weight-gradient waves)
    auto word = [&](int e, int& t, int& le) { t = e < C::NX ? 0 : (e < C::NX + C::NA ? 1 : 2); le = e - (t == 0 ? 0 : (t == 1 ? C::NX : C::NX + C::NA)); };
    auto load = [&](int work) {                            // unconditional, clamped (replaced by zeros at the store where outside the image)
        const long long img = work / C::TPI; const int ty0 = (work % C::TPI) * C::TH;
#pragma unroll
        for (int k = 0; k < C::KW; ++k) {
            int e = t2 + k * 256; e = e < C::NWORD ? e : C::NWORD - 1;
            int t, le; word(e, t, le);
            const int c8 = le & 1, px = (le >> 1) % C::HW; int gy = ty0 - (t == 0 ? 2 : 1) + le / (2 * C::HW);
            gy = gy < 0 ? 0 : (gy > C::HW - 1 ? C::HW - 1 : gy);
            const unsigned short* src = t == 0 ? a.dy : (t == 1 ? a.a_fwd : a.x_fwd);
            st[20 + k] = __builtin_bit_cast(f32x4, *(const uint4*)(src + ((img * C::HW + gy) * C::HW + px) * C::C + c8 * 8));
        }
    };
    auto item = [&](int w) { return w < nwork ? w : nwork - 1; };
    // weight gradient of one layer over this wave's NR consecutive pixel rows: d = output-gradient tile (row offset d_row), b = input tile
    auto wgrad = [&](const unsigned short* s_d, int d_row, const unsigned short* s_b, const int a0) {
        const int r0 = C::NR * rw;
        int ocol[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) ocol[h] = (16 * (kq >> 1) + 8 * h + 4 * (kq & 1) + rq) * C::S + 4 * cp;
        auto tr = [&](const unsigned short* base, int off) {
            const rb_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((rb_lds_s16x4_ptr)(base + ocol[0] + off));
            const rb_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((rb_lds_s16x4_ptr)(base + ocol[1] + off));
            return (bf16x8)__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        };
        bf16x8 dd[C::NR];
#pragma unroll
        for (int j = 0; j < C::NR; ++j) dd[j] = tr(s_d, ((r0 + j + d_row) * C::P + 1) * C::S);
#pragma unroll
        for (int j = 0; j < C::NR; ++j) st[a0 + 9] = MFMA_BF16(dd[j], ones, st[a0 + 9]);
#pragma unroll
        for (int R = 0; R < C::NR + 2; ++R) {                // tile row r0 + R of the input tile (row 0 = ty0-1, col 0 = -1)
            bf16x8 fb[3];
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) fb[kx] = tr(s_b, ((r0 + R) * C::P + kx) * C::S);
#pragma unroll
            for (int j = 0; j < C::NR; ++j) {
                const int ky = R - j;
                if (ky < 0 || ky > 2) continue;
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) st[a0 + ky * 3 + kx] = MFMA_BF16(dd[j], fb[kx], st[a0 + ky * 3 + kx]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    if (!conv_role && (int)blockIdx.x < nwork) load(blockIdx.x);
    for (int work = blockIdx.x; work < nwork; work += gridDim.x) {
        const long long img = work / C::TPI; const int ty0 = (work % C::TPI) * C::TH;
        __syncthreads();
        S16_TCK(3);                                         // phase-2 work + wait at the top barrier
        if (!conv_role) {
#pragma unroll
            for (int k = 0; k < C::KW; ++k) {
                const int e = t2 + k * 256;
                int t, le; word(e < C::NWORD ? e : C::NWORD - 1, t, le);
                const int ry = le / (2 * C::HW), gy = ty0 - (t == 0 ? 2 : 1) + ry;
                const bool in = gy >= 0 && gy < C::HW;
                unsigned short* dst = (t == 0 ? s_x : (t == 1 ? s_a : s_p)) + (ry * C::P + (le >> 1) % C::HW + 1) * C::S + (le & 1) * 8;
                const uint4 v = __builtin_bit_cast(uint4, st[20 + k]);
                const uint4 r = t == 0 ? v : (uint4){rb_relu2(v.x), rb_relu2(v.y), rb_relu2(v.z), rb_relu2(v.w)};
                if (e < C::NWORD) *(uint4*)dst = in ? r : (uint4){0u, 0u, 0u, 0u};
            }
        }
        __syncthreads();
        S16_TCK(0);                                         // staging (+ the conv waves' wait for it)
        if (!conv_role) load(item(work + gridDim.x));

        if (conv_role) {
            // ---- da = convT2(dy) * (a > 0) on rows ty0-1 .. ty0+TH -> s_y: tiles rw, rw+4, ... in groups of CG: a group's operand reads go
            // out together, then its MFMAs (K-step-major: the accumulator chains of the group interleave)
#pragma unroll
            for (int g0 = 0; g0 < C::MT1; g0 += CG) {
                bf16x8 av[CG][C::NK];
                uint2 mk[CG];
                int yb[CG];
#pragma unroll
                for (int q = 0; q < CG; ++q) {
                    const int mt = g0 + q < C::MT1 ? g0 + q : C::MT1 - 1;
                    int t = rw + 4 * mt; t = t < C::NMT1 ? t : C::NMT1 - 1;            // (a clamped duplicate rewrites the same values)
                    const int pl = t * 16 + i, px = pl % C::HW, ry = pl / C::HW;
                    const unsigned short* src = s_x + (ry * C::P + px) * C::S;
                    yb[q] = (ry * C::P + px + 1) * C::S + kq * 4;
#pragma unroll
                    for (int m = 0; m < C::NK; ++m) av[q][m] = *(const bf16x8*)(src + koff[m]);
                    mk[q] = *(const uint2*)(s_a + yb[q]);
                }
                f32x4 acc[CG];
#pragma unroll
                for (int q = 0; q < CG; ++q) acc[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int m = 0; m < C::NK; ++m)
#pragma unroll
                    for (int q = 0; q < CG; ++q) acc[q] = MFMA_BF16(__builtin_bit_cast(bf16x8, st[m]), av[q][m], acc[q]);
#pragma unroll
                for (int q = 0; q < CG; ++q) {
                    float v[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = rb_lane(mk[q], r) > 0.f ? acc[q][r] : 0.f;
                    const uint2 raw = rb_pack(v);
                    *(uint2*)(s_y + yb[q]) = raw;
                    if (a.da_out) {
                        const int mt = g0 + q < C::MT1 ? g0 + q : C::MT1 - 1;
                        int t = rw + 4 * mt; t = t < C::NMT1 ? t : C::NMT1 - 1;
                        const int pl = t * 16 + i, px = pl % C::HW, ry = pl / C::HW, gy = ty0 - 1 + ry;
                        if (ry >= 1 && ry <= C::TH) *(uint2*)(a.da_out + ((img * C::HW + gy) * C::HW + px) * C::C + kq * 4) = raw;
                    }
                }
                __builtin_amdgcn_sched_barrier(0);            // (register budget: the next group's reads stay behind this group's epilogue)
            }
        } else {
            wgrad(s_x, 2, s_a, 0);                           // conv2's weight / bias gradient from (dy, relu(a)): nothing the conv waves are producing
        }
        S16_TCK(1);                                         // phase-1 work
        __syncthreads();
        S16_TCK(2);                                         // wait for the other role
        if (conv_role) {
            // ---- dx = convT1(da) * (x > 0) + dy on rows ty0 .. ty0+TH-1 -> HBM
            static_assert(C::MT2 % CG == 0, "whole groups");
#pragma unroll
            for (int g0 = 0; g0 < C::MT2; g0 += CG) {
                bf16x8 av[CG][C::NK];
                uint2 mk[CG], sk[CG];
#pragma unroll
                for (int q = 0; q < CG; ++q) {
                    const int pl = (rw + 4 * (g0 + q)) * 16 + i, px = pl % C::HW, oy = pl / C::HW;
                    const unsigned short* src = s_y + (oy * C::P + px) * C::S;
#pragma unroll
                    for (int m = 0; m < C::NK; ++m) av[q][m] = *(const bf16x8*)(src + koff[m]);
                    mk[q] = *(const uint2*)(s_p + ((oy + 1) * C::P + px + 1) * C::S + kq * 4);
                    sk[q] = *(const uint2*)(s_x + ((oy + 2) * C::P + px + 1) * C::S + kq * 4);
                }
                f32x4 acc[CG];
#pragma unroll
                for (int q = 0; q < CG; ++q) acc[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int m = 0; m < C::NK; ++m)
#pragma unroll
                    for (int q = 0; q < CG; ++q) acc[q] = MFMA_BF16(__builtin_bit_cast(bf16x8, st[5 + m]), av[q][m], acc[q]);
#pragma unroll
                for (int q = 0; q < CG; ++q) {
                    const int pl = (rw + 4 * (g0 + q)) * 16 + i, px = pl % C::HW, oy = pl / C::HW;
                    float v[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = (rb_lane(mk[q], r) > 0.f ? acc[q][r] : 0.f) + rb_lane(sk[q], r);
                    *(uint2*)(a.dx_out + ((img * C::HW + ty0 + oy) * C::HW + px) * C::C + kq * 4) = rb_pack(v);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
            wgrad(s_y, 1, s_p, 10);                          // conv1's weight / bias gradient from (da, relu(x))
        }
    }
#ifdef RBF_TIMING
    if ((tid & 255) == 0) for (int q = 0; q < 4; ++q) atomicAdd(&g_rbf_timing[(tid >> 8) * 4 + q], (unsigned long long)tacc_[q]);
#endif
    // ---- the four weight-gradient waves summed through LDS in wave order; one slab per workgroup and layer
    __syncthreads();
    float* red = (float*)smem_h;                              // [2][WLEN] then [2][4][16] bias partials
    float* redb = red + 2 * C::WLEN;
    for (int w = 0; w < 4; ++w) {
        if (!conv_role && rw == w) {
#pragma unroll
            for (int tap = 0; tap < 9; ++tap)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int o = ((kq * 4 + r) * 9 + tap) * C::C + i;
                    red[o] = (w == 0) ? st[tap][r] : red[o] + st[tap][r];
                    red[C::WLEN + o] = (w == 0) ? st[10 + tap][r] : red[C::WLEN + o] + st[10 + tap][r];
                }
        }
        __syncthreads();
    }
    if (!conv_role && i == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) { redb[rw * 16 + kq * 4 + r] = st[9][r]; redb[4 * 16 + rw * 16 + kq * 4 + r] = st[19][r]; }
    }
    __syncthreads();
    float* sl2 = a.slab2 + (long long)blockIdx.x * C::SLAB;
    float* sl1 = a.slab1 + (long long)blockIdx.x * C::SLAB;
    for (int e = tid; e < C::WLEN; e += C::NT) { sl2[e] = red[e]; sl1[e] = red[C::WLEN + e]; }
    if (tid < 16) {
        float t2s = 0.f, t1s = 0.f;
        for (int w = 0; w < 4; ++w) { t2s += redb[w * 16 + tid]; t1s += redb[4 * 16 + w * 16 + tid]; }
        sl2[C::WLEN + tid] = t2s; sl1[C::WLEN + tid] = t1s;
    }
}
#endif      // RBFULL16_SPECIALISED == 1
// ---- the same two roles with the tiles filled by LDS-DMA into TWO tile buffers (RBFULL16_SPECIALISED == 2, the default).
// Phase clocks of the kernel above (scratch/kbench_rb16.hip): 28 % of an item is "staging" -- the weight-gradient waves wait for their
// prefetched words, apply the ReLU and write 8 x 16 bytes per thread to LDS while the conv waves idle.  A row of a tile is 32 pixels x 32
// bytes = 1 KB, contiguous in HBM (NHWC) and in LDS: exactly ONE global_load_lds_dwordx4 wave-instruction (lane = 16-byte word of the
// row), no registers, no LDS store instructions.  So: two tile buffers (16-row tiles: 2 x 79 KB = 161 KB, one workgroup per CU); at the top of
// item k the weight-gradient waves issue the row DMAs (56 per 16-row item) of item k + 1 into the other buffer (a whole item of flight time), drain them (vmcnt(0)) just before the item's last barrier.
// What changes for the consumers: relu(a) / relu(x) are no longer applied while staging -- the tiles hold RAW a and x; the ReLU masks
// read them as before (> 0 is the same truth), the weight-gradient operand fragments take the ReLU in registers (one v_pk_max_i16 per
// dword: max(x, 0) on bf16 bits = the staged relu, -0 -> +0 included).  Rows outside the image (first / last item of an image) are
// zero-filled with one 16-byte LDS store per lane by the wave that would have issued their DMA.  Two barriers per item instead of three,
// both raw (s_waitcnt lgkmcnt(0) + s_barrier): a __syncthreads() would also wait for the conv waves' dx stores.
// The DMAs are issued from inline asm (the compiler, seeing an LDS write it cannot place, would wait for a builtin's load before the next
// LDS read of the issuing wave): M0 is set in the same statement, the waits are explicit, and the kernel drains its last prefetch before exit.
__device__ __forceinline__ void rb_glds16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void rb_raw_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
template <int N> __device__ __forceinline__ void rb_wait_vm() { static_assert(N >= 0 && N < 64, "vmcnt"); asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }
__device__ unsigned short g_rb_zero_row[64 * 512];     // 64 x 1 KB of zeros: the DMA source of tile rows outside the image (one per workgroup mod 64:
                                                        // every workgroup reading the SAME 1 KB hit one L2 channel -- 352 vs 338 us per launch)
#ifndef RBFULL16D_NBUF
#define RBFULL16D_NBUF 2           // (three buffers of 8-row tiles measured like two: 340-352 vs 339-349 us)
#endif
#ifndef RBFULL16D_CG
#define RBFULL16D_CG 5
#endif
#ifndef RBFULL16D_INTERLEAVE
#define RBFULL16D_INTERLEAVE 1
#endif
#ifndef RBFULL16D_N1
#define RBFULL16D_N1 10             // rows (of a weight-gradient wave's 14) issued during phase 1; the rest during phase 2 (engine, us per launch: all up front 268, 14: 256, 10: 249)
#endif
#ifndef RBFULL16D_NC
#define RBFULL16D_NC 0             // tile rows (of 32) whose DMA each conv wave issues (measured: 0 best -- 338 us; 2 / 3 / 4: 374-387 / 376-380 / 365-371)
#endif
__global__ __launch_bounds__(512, 2) void resblock_bwd_full16d_bf16_kernel(RbFullArgs a) {
    using C = RbFull16S;
    constexpr int TILE = C::X_ELEMS + 3 * C::Y_ELEMS;       // elements of one tile buffer: s_x, s_y, s_a, s_p
    extern __shared__ __attribute__((aligned(16))) unsigned short smem_h[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i = lane & 15, kq = lane >> 4, rq = (lane & 15) >> 2, cp = lane & 3;
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    const bool conv_role = wv < 4;
    const int rw = wv & 3;
#ifdef RBF_TIMING
    long long tacc_[4] = {0, 0, 0, 0}, tlast_ = clock64();
#endif
    constexpr int NBUF = RBFULL16D_NBUF;                    // tile buffers: item k computes in buffer k % NBUF while the rows of items k+1 .. k+NBUF-1 are in flight / landed
    for (int e = tid; e < NBUF * TILE / 8; e += C::NT) ((uint4*)smem_h)[e] = (uint4){0u, 0u, 0u, 0u};   // column halos of all buffers stay zero
    constexpr int NST = 20;
    f32x4 st[NST];                                          // conv role: st[m] / st[5 + m] bank fragments; weight-gradient role: accumulators (as above)
    if (conv_role) {
#pragma unroll
        for (int m = 0; m < C::NK; ++m) {
            st[m] = __builtin_bit_cast(f32x4, *(const uint4*)(a.bank2_t + i * C::WS + m * 32 + kq * 8));
            st[5 + m] = __builtin_bit_cast(f32x4, *(const uint4*)(a.bank1_t + i * C::WS + m * 32 + kq * 8));
        }
#pragma unroll
        for (int q = 10; q < NST; ++q) st[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
    } else {
#pragma unroll
        for (int q = 0; q < NST; ++q) st[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);                   // vmcnt(0): the bank fragments are in
    const bf16x8 ones = __builtin_bit_cast(bf16x8, (uint4){0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u});
    int koff[C::NK];
#pragma unroll
    for (int m = 0; m < C::NK; ++m) {
        int tap = 2 * m + (kq >> 1); const int chunk = kq & 1; if (tap > 8) tap = 8;
        koff[m] = ((tap / 3) * C::P + (tap % 3)) * C::S + chunk * 8;
    }
    constexpr int CG = RBFULL16D_CG;
    const int nwork = a.n * C::TPI;
    const unsigned lds0 = (unsigned)(unsigned long long)(void*)smem_h;          // LDS byte address of the tile buffers (low half of the flat address)
    auto item = [&](int w) { return w < nwork ? w : nwork - 1; };
    // the 32 tile rows of item `work` -> buffer b, rows rw, rw + 4, ... by this (weight-gradient) wave: a DMA per row inside the image, zeros otherwise
    // (issuing one row costs its wave ~150 cycles -- 1 KB through the CU's load path -- so the rows are dealt to BOTH roles: NC per conv
    //  wave, the rest to the weight-gradient waves, whose phases are the shorter ones)
    constexpr int NROW = C::XR + 2 * C::YR, NC = RBFULL16D_NC, NWR = (NROW - 4 * NC + 3) / 4;
    // row jj (of this wave's share) of item `work` -> buffer b
    auto fill_row = [&](int work, int b, int jj) {
        const long long img = work / C::TPI; const int ty0 = (work % C::TPI) * C::TH;
        if (jj >= (conv_role ? NC : NWR)) return;           // (wave-uniform)
        const int j = conv_role ? 4 * NWR + rw + 4 * jj : rw + 4 * jj;
        if (j >= NROW) return;
        const int t = j < C::XR ? 0 : (j < C::XR + C::YR ? 1 : 2), ry = j - (t == 0 ? 0 : (t == 1 ? C::XR : C::XR + C::YR));
        const int gy = ty0 - (t == 0 ? 2 : 1) + ry;
        // element offset of the row's first interior pixel inside the buffer: s_x | s_y | s_a | s_p
        const int eoff = (t == 0 ? 0 : (t == 1 ? C::X_ELEMS + C::Y_ELEMS : C::X_ELEMS + 2 * C::Y_ELEMS)) + (ry * C::P + 1) * C::S;
        // a row outside the image comes from a row of zeros: every wave then issues the SAME number of DMAs per item, so the counted
        // wait for "all but the younger items' rows" is an immediate
        const unsigned short* src = (gy >= 0 && gy < C::HW) ? (t == 0 ? a.dy : (t == 1 ? a.a_fwd : a.x_fwd)) + ((img * C::HW + gy) * C::HW) * C::C + lane * 8
                                                            : g_rb_zero_row + (blockIdx.x & 63) * 512 + lane * 8;
        rb_glds16(src, lds0 + (unsigned)(b * TILE + eoff) * 2u);
    };
    auto fill = [&](int work, int b) {
#pragma unroll
        for (int jj = 0; jj < (NC > NWR ? NC : NWR); ++jj) fill_row(work, b, jj);
    };
    auto relu8 = [](bf16x8 v) {
        const uint4 u = __builtin_bit_cast(uint4, v);
        return __builtin_bit_cast(bf16x8, (uint4){rb_relu2_max(u.x), rb_relu2_max(u.y), rb_relu2_max(u.z), rb_relu2_max(u.w)});
    };
    // weight gradient of one layer over this wave's NR consecutive pixel rows: d = output-gradient tile (row offset d_row), b = RAW input tile
    // RBFULL16D_INTERLEAVE: conv2's weight gradient (phase 1) issues the NEXT item's row DMAs between its MFMA groups, a few per pixel row --
    // issued back to back at the top of the item they cost the wave ~190 cycles each (the CU's load path drains at its share of HBM and the
    // wave waits for queue space), 2600 cycles per item in front of phase 1 while the conv waves waited at the mid barrier.
    // The first N1 rows go out during phase 1, the rest during phase 2 (whose weight-gradient work is the shorter role's).
    constexpr int N1 = RBFULL16D_N1 < NWR ? RBFULL16D_N1 : NWR;
    auto wgrad = [&](const unsigned short* s_d, int d_row, const unsigned short* s_b, const int a0, int fill_work, int fill_b, const int f_lo, const int f_hi) {
        const int FPR = (f_hi - f_lo + C::NR + 1) / (C::NR + 2);      // DMAs per pixel-row iteration (NR + 2 iterations cover rows f_lo .. f_hi - 1)
        const int r0 = C::NR * rw;
        int ocol[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) ocol[h] = (16 * (kq >> 1) + 8 * h + 4 * (kq & 1) + rq) * C::S + 4 * cp;
        auto tr = [&](const unsigned short* base, int off) {
            const rb_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((rb_lds_s16x4_ptr)(base + ocol[0] + off));
            const rb_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((rb_lds_s16x4_ptr)(base + ocol[1] + off));
            return (bf16x8)__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        };
        bf16x8 dd[C::NR];
#pragma unroll
        for (int j = 0; j < C::NR; ++j) dd[j] = tr(s_d, ((r0 + j + d_row) * C::P + 1) * C::S);
#pragma unroll
        for (int j = 0; j < C::NR; ++j) st[a0 + 9] = MFMA_BF16(dd[j], ones, st[a0 + 9]);
#pragma unroll
        for (int R = 0; R < C::NR + 2; ++R) {
            bf16x8 fb[3];
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) fb[kx] = relu8(tr(s_b, ((r0 + R) * C::P + kx) * C::S));
            if (RBFULL16D_INTERLEAVE) {
#pragma unroll
                for (int f = 0; f < 4; ++f)
                    if (f < FPR && f_lo + R * FPR + f < f_hi) fill_row(fill_work, fill_b, f_lo + R * FPR + f);
            }
#pragma unroll
            for (int j = 0; j < C::NR; ++j) {
                const int ky = R - j;
                if (ky < 0 || ky > 2) continue;
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) st[a0 + ky * 3 + kx] = MFMA_BF16(dd[j], fb[kx], st[a0 + ky * 3 + kx]);
            }
        }
    };

    rb_raw_barrier();                                       // the zero fill is in
#pragma unroll
    for (int d = 0; d < NBUF - 1; ++d) fill(item(blockIdx.x + d * gridDim.x), d);
    // rows of the first item landed: all but the (NBUF - 2) x rows-per-wave youngest DMAs of this wave
    if (conv_role) rb_wait_vm<(NBUF - 2) * NC>(); else rb_wait_vm<(NBUF - 2) * NWR>();
    rb_raw_barrier();
    int b = 0;
    for (int work = blockIdx.x; work < nwork; work += gridDim.x, b = (b + 1 == NBUF ? 0 : b + 1)) {
        const long long img = work / C::TPI; const int ty0 = (work % C::TPI) * C::TH;
        unsigned short* s_x = smem_h + b * TILE;
        unsigned short* s_y = s_x + C::X_ELEMS;
        unsigned short* s_a = s_y + C::Y_ELEMS;
        unsigned short* s_p = s_a + C::Y_ELEMS;
        S16_TCK(3);                                         // (phase-2 work + wait at the end barrier of the previous item)
        if (!RBFULL16D_INTERLEAVE || conv_role) fill(item(work + (NBUF - 1) * gridDim.x), (b + NBUF - 1) % NBUF);      // that buffer's readers (item k - 1) finished before the barrier that ended the previous item
        S16_TCK(0);                                         // DMA issue
        if (conv_role) {
            // ---- da = convT2(dy) * (a > 0) on rows ty0-1 .. ty0+TH -> s_y
#pragma unroll
            for (int g0 = 0; g0 < C::MT1; g0 += CG) {
                bf16x8 av[CG][C::NK];
                uint2 mk[CG];
                int yb[CG];
#pragma unroll
                for (int q = 0; q < CG; ++q) {
                    const int mt = g0 + q < C::MT1 ? g0 + q : C::MT1 - 1;
                    int t = rw + 4 * mt; t = t < C::NMT1 ? t : C::NMT1 - 1;            // (a clamped duplicate rewrites the same values)
                    const int pl = t * 16 + i, px = pl % C::HW, ry = pl / C::HW;
                    const unsigned short* src = s_x + (ry * C::P + px) * C::S;
                    yb[q] = (ry * C::P + px + 1) * C::S + kq * 4;
#pragma unroll
                    for (int m = 0; m < C::NK; ++m) av[q][m] = *(const bf16x8*)(src + koff[m]);
                    mk[q] = *(const uint2*)(s_a + yb[q]);
                }
                f32x4 acc[CG];
#pragma unroll
                for (int q = 0; q < CG; ++q) acc[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int m = 0; m < C::NK; ++m)
#pragma unroll
                    for (int q = 0; q < CG; ++q) acc[q] = MFMA_BF16(__builtin_bit_cast(bf16x8, st[m]), av[q][m], acc[q]);
#pragma unroll
                for (int q = 0; q < CG; ++q) {
                    float v[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = rb_lane(mk[q], r) > 0.f ? acc[q][r] : 0.f;
                    const uint2 raw = rb_pack(v);
                    *(uint2*)(s_y + yb[q]) = raw;
                    if (a.da_out) {
                        const int mt = g0 + q < C::MT1 ? g0 + q : C::MT1 - 1;
                        int t = rw + 4 * mt; t = t < C::NMT1 ? t : C::NMT1 - 1;
                        const int pl = t * 16 + i, px = pl % C::HW, ry = pl / C::HW, gy = ty0 - 1 + ry;
                        if (ry >= 1 && ry <= C::TH) *(uint2*)(a.da_out + ((img * C::HW + gy) * C::HW + px) * C::C + kq * 4) = raw;
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
            wgrad(s_x, 2, s_a, 0, item(work + (NBUF - 1) * gridDim.x), (b + NBUF - 1) % NBUF, 0, N1);      // conv2's weight / bias gradient from (dy, relu(a)) + the next item's first rows
        }
        S16_TCK(1);                                         // phase-1 work
        rb_raw_barrier();                                   // da is complete (LDS only: the DMAs of the next item stay in flight)
        S16_TCK(2);
        if (conv_role) {
            // ---- dx = convT1(da) * (x > 0) + dy on rows ty0 .. ty0+TH-1 -> HBM
            constexpr int CG2 = C::MT2 % CG == 0 ? CG : (C::MT2 % 2 == 0 ? 2 : 1);
#pragma unroll
            for (int g0 = 0; g0 < C::MT2; g0 += CG2) {
                bf16x8 av[CG2][C::NK];
                uint2 mk[CG2], sk[CG2];
#pragma unroll
                for (int q = 0; q < CG2; ++q) {
                    const int pl = (rw + 4 * (g0 + q)) * 16 + i, px = pl % C::HW, oy = pl / C::HW;
                    const unsigned short* src = s_y + (oy * C::P + px) * C::S;
#pragma unroll
                    for (int m = 0; m < C::NK; ++m) av[q][m] = *(const bf16x8*)(src + koff[m]);
                    mk[q] = *(const uint2*)(s_p + ((oy + 1) * C::P + px + 1) * C::S + kq * 4);
                    sk[q] = *(const uint2*)(s_x + ((oy + 2) * C::P + px + 1) * C::S + kq * 4);
                }
                f32x4 acc[CG2];
#pragma unroll
                for (int q = 0; q < CG2; ++q) acc[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int m = 0; m < C::NK; ++m)
#pragma unroll
                    for (int q = 0; q < CG2; ++q) acc[q] = MFMA_BF16(__builtin_bit_cast(bf16x8, st[5 + m]), av[q][m], acc[q]);
#pragma unroll
                for (int q = 0; q < CG2; ++q) {
                    const int pl = (rw + 4 * (g0 + q)) * 16 + i, px = pl % C::HW, oy = pl / C::HW;
                    float v[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = (rb_lane(mk[q], r) > 0.f ? acc[q][r] : 0.f) + rb_lane(sk[q], r);
                    *(uint2*)(a.dx_out + ((img * C::HW + ty0 + oy) * C::HW + px) * C::C + kq * 4) = rb_pack(v);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            // this wave's row DMAs are older than its MT2 dx stores and the counter retires in order: all but the MT2 youngest done = rows landed
            rb_wait_vm<(NBUF - 2) * NC + C::MT2>();
        } else {
            wgrad(s_y, 1, s_p, 10, item(work + (NBUF - 1) * gridDim.x), (b + NBUF - 1) % NBUF, N1, NWR);      // conv1's weight / bias gradient from (da, relu(x)) + the remaining rows
            rb_wait_vm<(NBUF - 2) * NWR>();                  // the NEXT item's rows have landed: all but the rows of the items after it
        }
        rb_raw_barrier();                                   // everyone is done with this buffer; the other one is complete
    }
#ifdef RBF_TIMING
    if ((tid & 255) == 0) for (int q = 0; q < 4; ++q) atomicAdd(&g_rbf_timing[(tid >> 8) * 4 + q], (unsigned long long)tacc_[q]);
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // (prefetches past the last item: nothing may land in LDS after this workgroup is gone)
    rb_raw_barrier();
    // ---- the four weight-gradient waves summed through LDS in wave order; one slab per workgroup and layer
    float* red = (float*)smem_h;                              // [2][WLEN] then [2][4][16] bias partials
    float* redb = red + 2 * C::WLEN;
    for (int w = 0; w < 4; ++w) {
        if (!conv_role && rw == w) {
#pragma unroll
            for (int tap = 0; tap < 9; ++tap)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int o = ((kq * 4 + r) * 9 + tap) * C::C + i;
                    red[o] = (w == 0) ? st[tap][r] : red[o] + st[tap][r];
                    red[C::WLEN + o] = (w == 0) ? st[10 + tap][r] : red[C::WLEN + o] + st[10 + tap][r];
                }
        }
        __syncthreads();
    }
    if (!conv_role && i == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) { redb[rw * 16 + kq * 4 + r] = st[9][r]; redb[4 * 16 + rw * 16 + kq * 4 + r] = st[19][r]; }
    }
    __syncthreads();
    float* sl2 = a.slab2 + (long long)blockIdx.x * C::SLAB;
    float* sl1 = a.slab1 + (long long)blockIdx.x * C::SLAB;
    for (int e = tid; e < C::WLEN; e += C::NT) { sl2[e] = red[e]; sl1[e] = red[C::WLEN + e]; }
    if (tid < 16) {
        float t2s = 0.f, t1s = 0.f;
        for (int w = 0; w < 4; ++w) { t2s += redb[w * 16 + tid]; t1s += redb[4 * 16 + w * 16 + tid]; }
        sl2[C::WLEN + tid] = t2s; sl1[C::WLEN + tid] = t1s;
    }
}
// Which whole-backward kernel the 16-channel blocks run (RBFULL16_SPECIALISED): 0 = resblock_bwd_full_bf16_kernel (256 threads, two
// workgroups per CU), 1 = the wave-specialised one (register staging), 2 = wave-specialised + LDS-DMA into RBFULL16D_NBUF tile buffers
// (DEFAULT, with 16-row tiles).  Measured (round 3, us per 8192-sample launch): micro-bench on random data 0: 348-371, 1: 368-376,
// 2 with 8-row tiles: 330-363 (2 or 3 buffers alike), 2 with 16-row tiles: 346-347 where 0 ran 362-371; in the engine (HIP events, A/B of
// library builds on one box, three rounds) 0: 272-278, 2 / 8 rows: 265-267, 2 / 16 rows: 258-263.  dx is bit-identical in all of them.
// All variants end up bound by the same thing -- how fast a CU gets an item's bytes through its load path (a row's DMA costs its wave
// ~150 cycles of issue: the queue drains at the CU's share of HBM) -- which is why fewer BYTES (16-row tiles: 36 instead of 41 KB per 8
// rows) bought more than any re-arrangement of the work.
// (Also tried on variant 2: two more waves, 640 threads, that only issue the row DMAs and wait for them -- the kernel then has to fit
//  168 registers, spills 28-43 whatever the conv group size, and runs 560 us; issuing the rows from a run-time loop instead of the
//  unrolled one: 415 against 335.)
#ifndef RBFULL16S_BPC
#define RBFULL16S_BPC 1            // workgroups per CU of the specialised kernel
#endif
int resblock_bwd_full_grid(int n) {
    if (RBFULL16_SPECIALISED) { const int w = n * RbFull16S::TPI; return w > 256 * RBFULL16S_BPC ? 256 * RBFULL16S_BPC : w; }
    int bpc = (int)((160 * 1024) / RbFull::LDS_BYTES);
    bpc = bpc < 1 ? 1 : (bpc > 4 ? 4 : bpc);
    const int w = n * RbFull::TPI;
    return w > 256 * bpc ? 256 * bpc : w;
}
// 16-channel residual blocks @32x32 only (CS_16_16_32).  slab2 / slab1: [grid][2320] floats each (grid = resblock_bwd_full_grid(n)).
bool launch_resblock_bwd_full_event_ok() { return RBFULL16_SPECIALISED == 2; }
void launch_resblock_bwd_full_bf16(const void* dy, const void* a_fwd, const void* x_fwd, void* dx_out, void* da_out, int n,
                                   const unsigned short* bank2_t, const unsigned short* bank1_t, float* slab2, float* slab1, hipStream_t st, hipEvent_t done_ev) {
    static std::once_flag attr;          // (launchers run on up to 4 group worker threads)
    std::call_once(attr, [] { hipFuncSetAttribute((const void*)resblock_bwd_full_bf16_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)RbFull::LDS_BYTES); });
    const int grid = resblock_bwd_full_grid(n);
    if (grid < 1) return;
    RbFullArgs a{(const unsigned short*)dy, (const unsigned short*)a_fwd, (const unsigned short*)x_fwd, (unsigned short*)dx_out, (unsigned short*)da_out,
                 bank2_t, bank1_t, slab2, slab1, n};
    if (RBFULL16_SPECIALISED) {
        static std::once_flag attr_s;
        // one workgroup per CU: request more LDS than the tiles need so that a second workgroup never lands on the same CU (RBFULL16S_BPC == 1)
        constexpr size_t LDS_S = RBFULL16S_BPC == 1 ? (RbFull16S::LDS_BYTES > 84 * 1024 ? RbFull16S::LDS_BYTES : 84 * 1024) : RbFull16S::LDS_BYTES;
#if RBFULL16_SPECIALISED == 1
        std::call_once(attr_s, [] { hipFuncSetAttribute((const void*)resblock_bwd_full16s_bf16_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_S); });
#endif
        if (RBFULL16_SPECIALISED == 2) {
            static std::once_flag attr_d;
            constexpr size_t LDS_D = RBFULL16D_NBUF * RbFull16S::TILE_BYTES > RbFull16S::RED_BYTES ? RBFULL16D_NBUF * RbFull16S::TILE_BYTES : RbFull16S::RED_BYTES;
            static_assert(LDS_D > 80 * 1024 && LDS_D <= 160 * 1024, "one workgroup per CU");
            std::call_once(attr_d, [] { hipFuncSetAttribute((const void*)resblock_bwd_full16d_bf16_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_D); });
            // done_ev: this launch's completion as an event (the dispatch packet's own signal: no marker packet behind the kernel)
            if (done_ev) hipExtLaunchKernelGGL(resblock_bwd_full16d_bf16_kernel, dim3(grid), dim3(512), (unsigned)LDS_D, st, nullptr, done_ev, 0, a);
            else
            hipLaunchKernelGGL(resblock_bwd_full16d_bf16_kernel, dim3(grid), dim3(512), LDS_D, st, a);
            return;
        }
#if RBFULL16_SPECIALISED == 1
        hipLaunchKernelGGL(resblock_bwd_full16s_bf16_kernel, dim3(grid), dim3(512), LDS_S, st, a);
#endif
        return;
    }
    hipLaunchKernelGGL(resblock_bwd_full_bf16_kernel, dim3(grid), dim3(RbFull::NT), RbFull::LDS_BYTES, st, a);
}

// ------------------------------------------------------------------------------------------ whole backward, 32-channel blocks @16x16
// Same scheme as resblock_bwd_full_bf16_kernel with the sizes forcing two changes: tiles (72.6 KB) + the two transposed
// filter banks (39 KB) leave room for ONE workgroup per CU, so the workgroup is 512 threads (8 waves = 2 per SIMD, what two
// 256-thread workgroups would give); and the 2 x 36 weight-gradient accumulator tiles are dealt to the waves by (tap, input
// block) column as in the column-split stand-alone kernel: every wave walks all 4 pixel steps with its 2-3 columns of both
// layers and owns its slab entries outright (no cross-wave reduction).  (A 256-thread variant with the banks streamed from
// L2 instead of LDS measured 410 us per launch: every K step waited ~0.5 us for its filter fragment.)
template <int HW_, int NT_, int TH_ = 8>
struct RbFull32T {                               // HW 16: 8-row tiles, 512 threads; HW 8: whole image, 256 threads (79 KB: two per CU)
    static constexpr int C = 32, HW = HW_, TH = TH_, S = RB_S32, P = HW + 2, TPI = HW / TH, NT = NT_, NW = NT_ / 64;
    static constexpr int XR = TH + 4, YR = TH + 2;
    static constexpr int X_ELEMS = XR * P * S, Y_ELEMS = YR * P * S;
    static constexpr int NK = 9, WS = NK * 32 + 16, W_ELEMS = C * WS;
    static constexpr int NMT1 = YR * HW / 16, NMT2 = TH * HW / 16;                                            // HW 16: 10 / 8 tiles over 8 waves; HW 8: 5 / 4 over 4 waves
    static_assert(TH != 8 || (NMT2 == NW && NMT1 <= 2 * NW), "one conv2 tile and at most two conv1 tiles per wave");
    static constexpr int NX = XR * HW * 4, NA = YR * HW * 4;                                                  // 16-byte words staged per tensor
    static constexpr int KX = (NX + NT - 1) / NT, KA = (NA + NT - 1) / NT;
    static constexpr int NSTEP = TH * HW / 32;                                                                 // 4 pixel steps of 32
    static constexpr int NQ = 18, QMAX = (NQ + NW - 1) / NW;                                                   // (tap, input block) columns; per wave
    static constexpr int WLEN = C * 9 * C, SLAB = WLEN + C;
    static constexpr size_t LDS_BYTES = (size_t)(X_ELEMS + 3 * Y_ELEMS + 2 * W_ELEMS) * 2;
};
using RbFull32 = RbFull32T<16, 512>;
#ifndef RB32S_TH
#define RB32S_TH 16          // rows per item of the wave-specialised kernel: 16 = whole image (128 KB of LDS now that no bank sits there): no halo rows re-read or
#endif                       // recomputed, three barriers per image instead of six.  It needs the weight-gradient step loops unrolled by 4, not 8 (spills: 1007 us);
                             // micro-bench per 8192 images: 8-row items 220-222 us, 16-row items 203 (step loops rolled / by 2: 206-212)
using RbFull32W = RbFull32T<16, 512, RB32S_TH>;
using RbFull32S = RbFull32T<8, 256>;

template <class C>
__global__ __launch_bounds__(C::NT, C::NT == 256 ? 2 : 1) void resblock_bwd_full32_bf16_kernel(RbFullArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned short smem_h[];
    unsigned short* s_x = smem_h;                         // dy rows ty0-2 .. ty0+TH+1
    unsigned short* s_y = s_x + C::X_ELEMS;               // d(conv1 output) rows ty0-1 .. ty0+TH
    unsigned short* s_a = s_y + C::Y_ELEMS;               // relu(conv1 output), same rows
    unsigned short* s_p = s_a + C::Y_ELEMS;               // relu(block input), same rows
    unsigned short* s_w1 = s_p + C::Y_ELEMS;              // transposed bank of conv2 (first conv of this pass)
    unsigned short* s_w2 = s_w1 + C::W_ELEMS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i = lane & 15, kq = lane >> 4, rq = (lane & 15) >> 2, cp = lane & 3;
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    const int qcnt = (C::NQ - wv + C::NW - 1) / C::NW;    // columns q = wv + NW*qq, qq < qcnt (8 waves: 3, 3, 2, 2, 2, 2, 2, 2)
    for (int e = tid; e < C::W_ELEMS / 8; e += C::NT) { ((uint4*)s_w1)[e] = ((const uint4*)a.bank2_t)[e]; ((uint4*)s_w2)[e] = ((const uint4*)a.bank1_t)[e]; }
    for (int e = tid; e < (C::X_ELEMS + 3 * C::Y_ELEMS) / 8; e += C::NT) ((uint4*)smem_h)[e] = (uint4){0u, 0u, 0u, 0u};   // column halos stay zero
    int koff[C::NK];
#pragma unroll
    for (int m = 0; m < C::NK; ++m) koff[m] = ((m / 3) * C::P + (m % 3)) * C::S + kq * 8;     // 32 channels: one tap per K step, lane quarter = 8-channel chunk
    f32x4 acc2[C::QMAX][2], acc1[C::QMAX][2], accb[2];                                        // weight-gradient tiles [column][output block]; bias rows (waves 6, 7)
#pragma unroll
    for (int q = 0; q < C::QMAX; ++q)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) { acc2[q][cb] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc1[q][cb] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
    accb[0] = accb[1] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const bf16x8 ones = __builtin_bit_cast(bf16x8, (uint4){0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u});
    const int bbase = i * C::WS + kq * 8;

    const int nwork = a.n * C::TPI;
    uint4 rx[C::KX], ra[C::KA], rp[C::KA];
    auto load = [&](int work) {
        const long long img = work / C::TPI; const int ty0 = (work % C::TPI) * C::TH;
#pragma unroll
        for (int k = 0; k < C::KX; ++k) {
            const int e = tid + k * C::NT;
            uint4 v = {0u, 0u, 0u, 0u};
            if (e < C::NX) { const int c8 = e & 3, px = (e >> 2) % C::HW, gy = ty0 - 2 + e / (4 * C::HW);
                             if (gy >= 0 && gy < C::HW) v = *(const uint4*)(a.dy + ((img * C::HW + gy) * C::HW + px) * C::C + c8 * 8); }
            rx[k] = v;
        }
#pragma unroll
        for (int k = 0; k < C::KA; ++k) {
            const int e = tid + k * C::NT;
            uint4 va = {0u, 0u, 0u, 0u}, vp = {0u, 0u, 0u, 0u};
            if (e < C::NA) { const int c8 = e & 3, px = (e >> 2) % C::HW, gy = ty0 - 1 + e / (4 * C::HW);
                             if (gy >= 0 && gy < C::HW) { const long long o = ((img * C::HW + gy) * C::HW + px) * C::C + c8 * 8;
                                                          va = *(const uint4*)(a.a_fwd + o); vp = *(const uint4*)(a.x_fwd + o); } }
            ra[k] = va; rp[k] = vp;
        }
    };
    if ((int)blockIdx.x < nwork) load(blockIdx.x);
    for (int work = blockIdx.x; work < nwork; work += gridDim.x) {
        const long long img = work / C::TPI; const int ty0 = (work % C::TPI) * C::TH;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < C::KX; ++k) {
            const int e = tid + k * C::NT;
            if (e < C::NX) *(uint4*)(s_x + ((e / (4 * C::HW)) * C::P + (e >> 2) % C::HW + 1) * C::S + (e & 3) * 8) = rx[k];
        }
#pragma unroll
        for (int k = 0; k < C::KA; ++k) {
            const int e = tid + k * C::NT;
            if (e < C::NA) {
                const int o = ((e / (4 * C::HW)) * C::P + (e >> 2) % C::HW + 1) * C::S + (e & 3) * 8;
                *(uint4*)(s_a + o) = (uint4){rb_relu2(ra[k].x), rb_relu2(ra[k].y), rb_relu2(ra[k].z), rb_relu2(ra[k].w)};
                *(uint4*)(s_p + o) = (uint4){rb_relu2(rp[k].x), rb_relu2(rp[k].y), rb_relu2(rp[k].z), rb_relu2(rp[k].w)};
            }
        }
        __syncthreads();
        if (work + (int)gridDim.x < nwork) load(work + gridDim.x);

        // ---- da = convT2(dy) * (a > 0) on rows ty0-1 .. ty0+TH -> s_y (rows outside the image: relu(a) is 0 there, so da is 0)
        {
            const int nmt = (C::NMT1 - wv + C::NW - 1) / C::NW;          // 8 waves: 2, 2, 1, 1, 1, 1, 1, 1
            int abase[2], ybase[2];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                int t = wave + C::NW * mt; t = t < C::NMT1 ? t : C::NMT1 - 1;
                const int pl = t * 16 + i, px = pl % C::HW, ry = pl / C::HW;
                abase[mt] = (ry * C::P + px) * C::S;
                ybase[mt] = (ry * C::P + px + 1) * C::S + kq * 4;
            }
            f32x4 acc[2][2];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) { acc[mt][0] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc[mt][1] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
            for (int m = 0; m < C::NK; ++m) {
                const bf16x8 b0 = *(const bf16x8*)(s_w1 + bbase + m * 32), b1 = *(const bf16x8*)(s_w1 + bbase + 16 * C::WS + m * 32);
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
                    if (mt < nmt) {
                        const bf16x8 av = *(const bf16x8*)(s_x + abase[mt] + koff[m]);
                        acc[mt][0] = MFMA_BF16(b0, av, acc[mt][0]);
                        acc[mt][1] = MFMA_BF16(b1, av, acc[mt][1]);
                    }
            }
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
                if (mt < nmt) {
#pragma unroll
                    for (int nb = 0; nb < 2; ++nb) {
                        const uint2 mk = *(const uint2*)(s_a + ybase[mt] + nb * 16);
                        float v[4];
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = rb_lane(mk, r) > 0.f ? acc[mt][nb][r] : 0.f;
                        const uint2 raw = rb_pack(v);
                        *(uint2*)(s_y + ybase[mt] + nb * 16) = raw;
                        if (a.da_out) {
                            const int pl = (wave + C::NW * mt) * 16 + i, px = pl % C::HW, ry = pl / C::HW, gy = ty0 - 1 + ry;
                            if (ry >= 1 && ry <= C::TH) *(uint2*)(a.da_out + ((img * C::HW + gy) * C::HW + px) * C::C + nb * 16 + kq * 4) = raw;
                        }
                    }
                }
        }
        __syncthreads();
        // ---- dx = convT1(da) * (x > 0) + dy on rows ty0 .. ty0+TH-1 -> HBM: one 16-pixel tile per wave
        {
            const int pl = wave * 16 + i, px = pl % C::HW, oy = pl / C::HW;
            const int abase = (oy * C::P + px) * C::S;
            f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
            for (int m = 0; m < C::NK; ++m) {
                const bf16x8 av = *(const bf16x8*)(s_y + abase + koff[m]);
                acc[0] = MFMA_BF16(*(const bf16x8*)(s_w2 + bbase + m * 32), av, acc[0]);
                acc[1] = MFMA_BF16(*(const bf16x8*)(s_w2 + bbase + 16 * C::WS + m * 32), av, acc[1]);
            }
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) {
                const uint2 mk = *(const uint2*)(s_p + ((oy + 1) * C::P + px + 1) * C::S + nb * 16 + kq * 4);
                const uint2 sk = *(const uint2*)(s_x + ((oy + 2) * C::P + px + 1) * C::S + nb * 16 + kq * 4);
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = (rb_lane(mk, r) > 0.f ? acc[nb][r] : 0.f) + rb_lane(sk, r);
                *(uint2*)(a.dx_out + ((img * C::HW + ty0 + oy) * C::HW + px) * C::C + nb * 16 + kq * 4) = rb_pack(v);
            }
        }
        // ---- weight gradients: conv2 from (dy, relu(a)), conv1 from (da, relu(x)); every wave walks all pixel steps for its columns
#pragma unroll
        for (int t = 0; t < C::NSTEP; ++t) {
            int orow[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int pl = 32 * t + 16 * (kq >> 1) + 8 * h + 4 * (kq & 1) + rq;
                orow[h] = ((pl / C::HW) * C::P + pl % C::HW) * C::S + 4 * cp;
            }
            auto tr = [&](const unsigned short* base, int off) {
                const rb_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((rb_lds_s16x4_ptr)(base + orow[0] + off));
                const rb_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((rb_lds_s16x4_ptr)(base + orow[1] + off));
                return (bf16x8)__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
            };
            bf16x8 d2[2], d1[2];
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) { d2[cb] = tr(s_x, (2 * C::P + 1) * C::S + cb * 16); d1[cb] = tr(s_y, (C::P + 1) * C::S + cb * 16); }
            if (wv == C::NW - 2) { accb[0] = MFMA_BF16(d2[0], ones, accb[0]); accb[1] = MFMA_BF16(d2[1], ones, accb[1]); }      // bias of conv2
            if (wv == C::NW - 1) { accb[0] = MFMA_BF16(d1[0], ones, accb[0]); accb[1] = MFMA_BF16(d1[1], ones, accb[1]); }      // bias of conv1
#pragma unroll
            for (int qq = 0; qq < C::QMAX; ++qq)
                if (qq < qcnt) {                                   // wave-uniform: EXEC stays full for the transpose reads
                    const int q = wv + C::NW * qq, tap = q >> 1, ib = q & 1;
                    const int toff = ((tap / 3) * C::P + (tap % 3)) * C::S + ib * 16;
                    const bf16x8 b2 = tr(s_a, toff), b1 = tr(s_p, toff);
#pragma unroll
                    for (int cb = 0; cb < 2; ++cb) { acc2[qq][cb] = MFMA_BF16(d2[cb], b2, acc2[qq][cb]); acc1[qq][cb] = MFMA_BF16(d1[cb], b1, acc1[qq][cb]); }
                }
        }
    }
    // ---- every wave owns its columns: straight to the slabs
    float* sl2 = a.slab2 + (long long)blockIdx.x * C::SLAB;
    float* sl1 = a.slab1 + (long long)blockIdx.x * C::SLAB;
#pragma unroll
    for (int qq = 0; qq < C::QMAX; ++qq)
        if (qq < qcnt) {
            const int q = wv + C::NW * qq, tap = q >> 1, ib = q & 1;
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int o = ((cb * 16 + kq * 4 + r) * 9 + tap) * C::C + ib * 16 + i;
                    sl2[o] = acc2[qq][cb][r]; sl1[o] = acc1[qq][cb][r];
                }
        }
    if (i == 0 && wv >= C::NW - 2) {
        float* sl = wv == C::NW - 2 ? sl2 : sl1;
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int r = 0; r < 4; ++r) sl[C::WLEN + cb * 16 + kq * 4 + r] = accb[cb][r];
    }
}
// ---- wave-specialised variant (HW = 16, 512 threads).  The kernel above is LDS-bandwidth-bound in its conv phases: with one or two
// 16-pixel tiles per wave every wave re-reads the whole filter bank (18 x 1 KB per conv and wave: 288 of the 738 KB an item reads),
// i.e. ~1.5 operand reads per MFMA, 6 LDS cycles against 16 MFMA cycles with eight waves on one LDS pipe.  Here the eight waves take
// two ROLES: waves 0-3 (one per SIMD) run the two transposed convs with BOTH filter banks in registers (2 x 18 fragments = 144 VGPRs,
// loaded once per launch: no weight reads at all), waves 4-7 run the weight gradients (conv2's while the conv waves produce da,
// conv1's while they produce dx); each SIMD hosts one wave of each role, so matrix work of one overlaps the LDS work of the other.
// Per item: LDS reads 738 -> 370 KB, same MFMA count, same three barriers.  Arithmetic per output element and per slab entry is
// unchanged (same K order, same pixel-step order): results are bit-identical to the kernel above.
#ifdef WG_TIMING      // scratch/kbench_rb.hip: per-phase shader-clock totals of wave 0 (conv role, slots 0-3) and wave 4 (weight-gradient role, 4-7)
__device__ unsigned long long g_rb_timing[8];
#define RTCK(k) do { if ((tid & 255) == 0) { const long long now_ = clock64(); tacc_[k] += now_ - tlast_; tlast_ = now_; } } while (0)
#else
#define RTCK(k) do { } while (0)
#endif
template <class C>
__global__ __launch_bounds__(512, 2) void resblock_bwd_full32s_bf16_kernel(RbFullArgs a) {
    static_assert(C::NT == 512 && C::HW == 16, "16x16 images, 8 waves");
    extern __shared__ __attribute__((aligned(16))) unsigned short smem_h[];
    unsigned short* s_x = smem_h;                         // dy rows ty0-2 .. ty0+TH+1
    unsigned short* s_y = s_x + C::X_ELEMS;               // d(conv1 output) rows ty0-1 .. ty0+TH
    unsigned short* s_a = s_y + C::Y_ELEMS;               // relu(conv1 output), same rows
    unsigned short* s_p = s_a + C::Y_ELEMS;               // relu(block input), same rows
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i = lane & 15, kq = lane >> 4, rq = (lane & 15) >> 2, cp = lane & 3;
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    const bool conv_role = wv < 4;
#ifdef WG_TIMING
    long long tacc_[4] = {0, 0, 0, 0}, tlast_ = clock64();
#endif
    const int rw = wv & 3;                                // index inside the role
    for (int e = tid; e < (C::X_ELEMS + 3 * C::Y_ELEMS) / 8; e += C::NT) ((uint4*)smem_h)[e] = (uint4){0u, 0u, 0u, 0u};   // column halos stay zero
    // Per-wave persistent state, ONE register array for both roles (a kernel's registers are allocated for the union of what its
    // waves may keep live: two separate sets spilled 471 registers).  conv role: st[2m + nb] = B fragment (tap m, output block nb) of
    // conv2's transposed bank, st[18 + 2m + nb] of conv1's.  weight-gradient role: st[2qq + cb] / st[10 + 2qq + cb] = accumulator
    // tiles of conv2 / conv1 for this wave's column qq and output block cb, st[20 + cb] = bias accumulators (rw 2: conv2, rw 3: conv1).
    constexpr int NT2 = 256, KX2 = (C::NX + NT2 - 1) / NT2, KA2 = (C::NA + NT2 - 1) / NT2;
    constexpr int NST = (22 + KX2 + 2 * KA2) > 36 ? (22 + KX2 + 2 * KA2) : 36;      // 36 fragments (conv role) / 22 accumulators + the prefetch words
    f32x4 st[NST];
    constexpr int QM = 5;
    const int qcnt = (C::NQ - rw + 3) / 4;                // columns q = rw + 4 * qq of the 18 (tap, input block) columns: 5, 5, 4, 4
    if (conv_role) {
#pragma unroll
        for (int m = 0; m < 9; ++m)
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) {
                st[2 * m + nb] = __builtin_bit_cast(f32x4, *(const uint4*)(a.bank2_t + (nb * 16 + i) * C::WS + m * 32 + kq * 8));
                st[18 + 2 * m + nb] = __builtin_bit_cast(f32x4, *(const uint4*)(a.bank1_t + (nb * 16 + i) * C::WS + m * 32 + kq * 8));
            }
    } else {
#pragma unroll
        for (int q = 0; q < NST; ++q) st[q] = (f32x4){0.f, 0.f, 0.f, 0.f};     // accumulators (0 .. 21); 22 .. are the prefetch words
    }
    // Drain the 36 fragment loads HERE.  Otherwise the compiler, which sees one register array, guards every first touch of st[k] inside
    // the loop with a counted s_waitcnt vmcnt(35 - k): harmless for the conv waves, but the weight-gradient waves have their nine
    // prefetch loads in flight at that point, and vmcnt(N < 9) in the middle of their MFMA stream waits for HBM on every item.
    __builtin_amdgcn_s_waitcnt(0x0F70);                   // vmcnt(0), lgkmcnt / expcnt untouched
    const bf16x8 ones = __builtin_bit_cast(bf16x8, (uint4){0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u});
    auto koffc = [](int m) { return ((m / 3) * C::P + (m % 3)) * C::S; };

    const int nwork = a.n * C::TPI;
    // Staging (global -> registers one item ahead -> LDS) is the weight-gradient waves' job: their half of st has room for the nine
    // prefetch words (st[22 .. 30]), the conv waves' half is full of filter fragments.  t2 = thread index among those 256 threads.
    const int t2 = tid - 256;
    auto load = [&](int work) {
        const long long img = work / C::TPI; const int ty0 = (work % C::TPI) * C::TH;
#pragma unroll
        for (int k = 0; k < KX2; ++k) {
            const int e = t2 + k * NT2;
            uint4 v = {0u, 0u, 0u, 0u};
            if (e < C::NX) { const int c8 = e & 3, px = (e >> 2) % C::HW, gy = ty0 - 2 + e / (4 * C::HW);
                             if (gy >= 0 && gy < C::HW) v = *(const uint4*)(a.dy + ((img * C::HW + gy) * C::HW + px) * C::C + c8 * 8); }
            st[22 + k] = __builtin_bit_cast(f32x4, v);
        }
#pragma unroll
        for (int k = 0; k < KA2; ++k) {
            const int e = t2 + k * NT2;
            uint4 va = {0u, 0u, 0u, 0u}, vp = {0u, 0u, 0u, 0u};
            if (e < C::NA) { const int c8 = e & 3, px = (e >> 2) % C::HW, gy = ty0 - 1 + e / (4 * C::HW);
                             if (gy >= 0 && gy < C::HW) { const long long o = ((img * C::HW + gy) * C::HW + px) * C::C + c8 * 8;
                                                          va = *(const uint4*)(a.a_fwd + o); vp = *(const uint4*)(a.x_fwd + o); } }
            st[22 + KX2 + k] = __builtin_bit_cast(f32x4, va); st[22 + KX2 + KA2 + k] = __builtin_bit_cast(f32x4, vp);
        }
    };
    // one pixel step (32 pixels) of a layer's weight gradient: d = the layer's output-gradient tile, src = its (ReLU'd) input tile
    auto wg_step = [&](int t, const unsigned short* s_d, int d_off, const unsigned short* s_src, const int a0, bool bias) {
        int orow[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int pl = 32 * t + 16 * (kq >> 1) + 8 * h + 4 * (kq & 1) + rq;
            orow[h] = ((pl / C::HW) * C::P + pl % C::HW) * C::S + 4 * cp;
        }
        auto tr = [&](const unsigned short* base, int off) {
            const rb_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((rb_lds_s16x4_ptr)(base + orow[0] + off));
            const rb_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((rb_lds_s16x4_ptr)(base + orow[1] + off));
            return (bf16x8)__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        };
        // all operand reads of the step go out before its first MFMA: with one column's read in flight the MFMA stream runs at the
        // LDS latency (200+ cycles with eight waves on the pipe) instead of 32 cycles per column
        bf16x8 d[2], b[QM];
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) d[cb] = tr(s_d, d_off + cb * 16);
#pragma unroll
        for (int qq = 0; qq < QM; ++qq) {
            const int q = rw + 4 * (qq < 4 ? qq : (qcnt > 4 ? 4 : 0)), tap = q >> 1, ib = q & 1;
            b[qq] = tr(s_src, ((tap / 3) * C::P + (tap % 3)) * C::S + ib * 16);
        }
        asm volatile("" ::: "memory");
        if (bias) { st[20] = MFMA_BF16(d[0], ones, st[20]); st[21] = MFMA_BF16(d[1], ones, st[21]); }
#pragma unroll
        for (int qq = 0; qq < QM; ++qq)
            if (qq < qcnt) {
#pragma unroll
                for (int cb = 0; cb < 2; ++cb) st[a0 + 2 * qq + cb] = MFMA_BF16(d[cb], b[qq], st[a0 + 2 * qq + cb]);
            }
    };

    if (!conv_role && (int)blockIdx.x < nwork) load(blockIdx.x);
    for (int work = blockIdx.x; work < nwork; work += gridDim.x) {
        const long long img = work / C::TPI; const int ty0 = (work % C::TPI) * C::TH;
        __syncthreads();
        RTCK(3);                                            // phase-2 work + wait at the top barrier
        if (!conv_role) {
#pragma unroll
            for (int k = 0; k < KX2; ++k) {
                const int e = t2 + k * NT2;
                if (e < C::NX) *(uint4*)(s_x + ((e / (4 * C::HW)) * C::P + (e >> 2) % C::HW + 1) * C::S + (e & 3) * 8) = __builtin_bit_cast(uint4, st[22 + k]);
            }
#pragma unroll
            for (int k = 0; k < KA2; ++k) {
                const int e = t2 + k * NT2;
                if (e < C::NA) {
                    const int o = ((e / (4 * C::HW)) * C::P + (e >> 2) % C::HW + 1) * C::S + (e & 3) * 8;
                    const uint4 va = __builtin_bit_cast(uint4, st[22 + KX2 + k]), vp = __builtin_bit_cast(uint4, st[22 + KX2 + KA2 + k]);
                    *(uint4*)(s_a + o) = (uint4){rb_relu2(va.x), rb_relu2(va.y), rb_relu2(va.z), rb_relu2(va.w)};
                    *(uint4*)(s_p + o) = (uint4){rb_relu2(vp.x), rb_relu2(vp.y), rb_relu2(vp.z), rb_relu2(vp.w)};
                }
            }
        }
        __syncthreads();
        RTCK(0);                                            // staging (+ the conv waves' wait for it)
        if (!conv_role && work + (int)gridDim.x < nwork) load(work + gridDim.x);

        if (conv_role) {
            // ---- da = convT2(dy) * (a > 0) on rows ty0-1 .. ty0+TH -> s_y: tiles rw, rw+4, rw+8 of the 10 (weights in registers), two
            // tiles at a time: four independent accumulator chains keep the matrix pipe busy through the MFMA latency
            for (int t = rw; t < C::NMT1; t += 4) {            // one tile at a time: its nine operand reads go out together, then 18 MFMAs
                const int pl = t * 16 + i, px = pl % C::HW, ry = pl / C::HW;
                const unsigned short* src = s_x + (ry * C::P + px) * C::S + kq * 8;
                const int yb = (ry * C::P + px + 1) * C::S + kq * 4;
                bf16x8 av[5];                                   // reads go out in batches of 5 + 4 taps (all nine at once spill)
#pragma unroll
                for (int m = 0; m < 5; ++m) av[m] = *(const bf16x8*)(src + koffc(m));
                const uint2 mk0 = *(const uint2*)(s_a + yb), mk1 = *(const uint2*)(s_a + yb + 16);
                asm volatile("" ::: "memory");
                f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
                for (int m = 0; m < 9; ++m) {
                    const bf16x8 cur = av[m % 5];
                    acc[0] = MFMA_BF16(__builtin_bit_cast(bf16x8, st[2 * m]), cur, acc[0]);
                    acc[1] = MFMA_BF16(__builtin_bit_cast(bf16x8, st[2 * m + 1]), cur, acc[1]);
                    if (m < 4) av[m] = *(const bf16x8*)(src + koffc(m + 5));        // slot m is free again: taps 5..8 follow behind
                }
#pragma unroll
                for (int nb = 0; nb < 2; ++nb) {
                    const uint2 mk = nb ? mk1 : mk0;
                    float v[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = rb_lane(mk, r) > 0.f ? acc[nb][r] : 0.f;
                    const uint2 raw = rb_pack(v);
                    *(uint2*)(s_y + yb + nb * 16) = raw;
                    if (a.da_out) {
                        const int gy = ty0 - 1 + ry;
                        if (ry >= 1 && ry <= C::TH) *(uint2*)(a.da_out + ((img * C::HW + gy) * C::HW + px) * C::C + nb * 16 + kq * 4) = raw;
                    }
                }
            }
        } else {
            // ---- conv2's weight / bias gradient from (dy, relu(a)): needs nothing the conv waves are producing
#pragma unroll 4                                             // (16-row items have 8 steps: unrolled by 8 the kernel spills -- hoisted operand addresses)
            for (int t = 0; t < C::NSTEP; ++t) wg_step(t, s_x, (2 * C::P + 1) * C::S, s_a, 0, rw == 2);
        }
        RTCK(1);                                            // phase-1 work
        __syncthreads();
        RTCK(2);                                            // wait for the other role
        if (conv_role) {
            // ---- dx = convT1(da) * (x > 0) + dy on rows ty0 .. ty0+TH-1 -> HBM: tiles rw, rw+4 of the 8, one at a time as above
            for (int t = rw; t < C::NMT2; t += 4) {
                const int pl = t * 16 + i, px = pl % C::HW, oy = pl / C::HW;
                const unsigned short* src = s_y + (oy * C::P + px) * C::S + kq * 8;
                bf16x8 av[5];
#pragma unroll
                for (int m = 0; m < 5; ++m) av[m] = *(const bf16x8*)(src + koffc(m));
                const int eo = ((oy + 1) * C::P + px + 1) * C::S + kq * 4;
                const uint2 mk0 = *(const uint2*)(s_p + eo), mk1 = *(const uint2*)(s_p + eo + 16);
                const uint2 sk0 = *(const uint2*)(s_x + eo + C::P * C::S), sk1 = *(const uint2*)(s_x + eo + C::P * C::S + 16);
                asm volatile("" ::: "memory");
                f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
                for (int m = 0; m < 9; ++m) {
                    const bf16x8 cur = av[m % 5];
                    acc[0] = MFMA_BF16(__builtin_bit_cast(bf16x8, st[18 + 2 * m]), cur, acc[0]);
                    acc[1] = MFMA_BF16(__builtin_bit_cast(bf16x8, st[18 + 2 * m + 1]), cur, acc[1]);
                    if (m < 4) av[m] = *(const bf16x8*)(src + koffc(m + 5));
                }
#pragma unroll
                for (int nb = 0; nb < 2; ++nb) {
                    const uint2 mk = nb ? mk1 : mk0, sk = nb ? sk1 : sk0;
                    float v[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = (rb_lane(mk, r) > 0.f ? acc[nb][r] : 0.f) + rb_lane(sk, r);
                    *(uint2*)(a.dx_out + ((img * C::HW + ty0 + oy) * C::HW + px) * C::C + nb * 16 + kq * 4) = rb_pack(v);
                }
            }
        } else {
            // ---- conv1's weight / bias gradient from (da, relu(x))
#pragma unroll 4
            for (int t = 0; t < C::NSTEP; ++t) wg_step(t, s_y, (C::P + 1) * C::S, s_p, 10, rw == 3);
        }
    }
#ifdef WG_TIMING
    if ((tid & 255) == 0) for (int q = 0; q < 4; ++q) atomicAdd(&g_rb_timing[(tid >> 8) * 4 + q], (unsigned long long)tacc_[q]);
#endif
    // ---- the weight-gradient waves own their columns: straight to the slabs (layout as the kernel above)
    if (!conv_role) {
        float* sl2 = a.slab2 + (long long)blockIdx.x * C::SLAB;
        float* sl1 = a.slab1 + (long long)blockIdx.x * C::SLAB;
#pragma unroll
        for (int qq = 0; qq < QM; ++qq)
            if (qq < qcnt) {
                const int q = rw + 4 * qq, tap = q >> 1, ib = q & 1;
#pragma unroll
                for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int o = ((cb * 16 + kq * 4 + r) * 9 + tap) * C::C + ib * 16 + i;
                        sl2[o] = st[2 * qq + cb][r]; sl1[o] = st[10 + 2 * qq + cb][r];
                    }
            }
        if (i == 0 && rw >= 2) {
            float* sl = rw == 2 ? sl2 : sl1;
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                for (int r = 0; r < 4; ++r) sl[C::WLEN + cb * 16 + kq * 4 + r] = st[20 + cb][r];
        }
    }
}

// ---- 8x8 images, FOUR per item, wave-specialised (round 3).  The kernel above handles ONE 8x8 image per item: 256 threads, one pixel tile
// per wave and conv -- every tile re-reads its conv's whole filter bank from LDS (18 KB of bank for 9 KB of pixels), three barriers per
// image, and the launch ran 67-73 us for 17 MB of tensors at 12 % matrix-pipe activity.  Here an item is four images (256 pixels, the
// pixel count of a 16x16 image) in the two-role scheme of resblock_bwd_full32s_bf16_kernel: conv waves 0-3 hold both transposed banks in
// registers and take four pixel tiles per conv, weight-gradient waves 4-7 stage the tiles (every staged row is inside its image: plain
// unconditional loads) and walk the item's eight 32-pixel steps with their columns.  LDS: per image four 10-row tiles (dy with its
// zero rows -1 and 8 -- da is only needed on the image's own rows --, d(conv1 output), relu(conv1 output), relu(block input)), 153.6 KB.
// Arithmetic per output element unchanged (same banks, same K order): dx is bit-identical to the kernel above; the slabs sum the same
// products in another order of partial sums.
struct RbFull32Q {
    static constexpr int C = 32, HW = 8, NIMG = 4, S = RB_S32, P = HW + 2, R = HW + 2;          // rows -1 .. 8 of every tile
    static constexpr int IMG_ELEMS = R * P * S, T_ELEMS = NIMG * IMG_ELEMS;                     // one image's / the item's share of a tile
    static constexpr int NK = 9, WS = NK * 32 + 16;
    static constexpr int NMT = NIMG * HW * HW / 16, MTW = NMT / 4;                              // 16 pixel tiles, 4 per conv wave
    static constexpr int NW16 = NIMG * HW * HW * 4, KW = NW16 / 256;                            // 16-byte words per tensor and item; per staging thread (4)
    static constexpr int NSTEP = NIMG * HW * HW / 32;                                           // 8 pixel steps of 32
    static constexpr int NQ = 18, WLEN = C * 9 * C, SLAB = WLEN + C;
    static constexpr size_t LDS_BYTES = (size_t)4 * T_ELEMS * 2;
    static_assert(LDS_BYTES <= 160 * 1024, "one workgroup per CU");
};
__global__ __launch_bounds__(512, 2) void resblock_bwd_full32q_bf16_kernel(RbFullArgs a) {
    using C = RbFull32Q;
    extern __shared__ __attribute__((aligned(16))) unsigned short smem_h[];
    unsigned short* s_x = smem_h;                         // dy, rows -1 .. 8 of each image (rows -1 and 8 stay zero)
    unsigned short* s_y = s_x + C::T_ELEMS;               // d(conv1 output)
    unsigned short* s_a = s_y + C::T_ELEMS;               // relu(conv1 output)
    unsigned short* s_p = s_a + C::T_ELEMS;               // relu(block input)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i = lane & 15, kq = lane >> 4, rq = (lane & 15) >> 2, cp = lane & 3;
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    const bool conv_role = wv < 4;
    const int rw = wv & 3;
#ifdef WG_TIMING
    long long tacc_[4] = {0, 0, 0, 0}, tlast_ = clock64();
#endif
    for (int e = tid; e < 4 * C::T_ELEMS / 8; e += 512) ((uint4*)smem_h)[e] = (uint4){0u, 0u, 0u, 0u};      // halo rows / columns stay zero
    // one register array for both roles, as in the 16x16 kernel: conv role st[2m + nb] / st[18 + 2m + nb] = bank fragments of conv2 / conv1;
    // weight-gradient role st[2qq + cb] / st[10 + 2qq + cb] accumulators, st[20 + cb] bias accumulators, st[22 .. 33] the 12 prefetch words
    constexpr int NST = 36, QM = 5;
    static_assert(22 + 3 * C::KW <= NST, "prefetch words fit behind the accumulators");
    f32x4 st[NST];
    const int qcnt = (C::NQ - rw + 3) / 4;
    if (conv_role) {
#pragma unroll
        for (int m = 0; m < 9; ++m)
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) {
                st[2 * m + nb] = __builtin_bit_cast(f32x4, *(const uint4*)(a.bank2_t + (nb * 16 + i) * C::WS + m * 32 + kq * 8));
                st[18 + 2 * m + nb] = __builtin_bit_cast(f32x4, *(const uint4*)(a.bank1_t + (nb * 16 + i) * C::WS + m * 32 + kq * 8));
            }
    } else {
#pragma unroll
        for (int q = 0; q < NST; ++q) st[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);                   // vmcnt(0): the fragments are in (see the 16x16 kernel)
    const bf16x8 ones = __builtin_bit_cast(bf16x8, (uint4){0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u});
    auto koffc = [](int m) { return ((m / 3) * C::P + (m % 3)) * C::S; };
    // pixel pl (0 .. 255) of the item: image pl / 64, row (pl % 64) / 8, column pl % 8; element offset of its WINDOW ORIGIN (row - 1, column - 1) in a tile
    auto porg = [](int pl) { return ((pl >> 6) * C::R + ((pl >> 3) & 7)) * C::P * C::S + (pl & 7) * C::S; };
    constexpr int CENTER = (C::P + 1) * C::S;

    const int nwork = (a.n + C::NIMG - 1) / C::NIMG;
    const int t2 = tid - 256;                             // staging thread: word e = t2 + 256 k of each tensor = (pixel e / 4, 8-channel chunk e % 4)
    auto load = [&](int work) {
        const int img0 = work * C::NIMG;
#pragma unroll
        for (int k = 0; k < C::KW; ++k) {
            const int e = t2 + k * 256, pl = e >> 2;
            const int n = img0 + (pl >> 6) < a.n ? img0 + (pl >> 6) : a.n - 1;        // images past the end: a valid one, replaced by zeros at the store
            const long long o = ((long long)n * 64 + (pl & 63)) * C::C + (e & 3) * 8;
            st[22 + k] = __builtin_bit_cast(f32x4, *(const uint4*)(a.dy + o));
            st[22 + C::KW + k] = __builtin_bit_cast(f32x4, *(const uint4*)(a.a_fwd + o));
            st[22 + 2 * C::KW + k] = __builtin_bit_cast(f32x4, *(const uint4*)(a.x_fwd + o));
        }
    };
    // one pixel step (32 pixels = half an image) of a layer's weight gradient, operands as in the 16x16 kernel.  The step loops below are
    // unrolled by 4, not 8: fully unrolled, the compiler hoists every step's operand addresses out of the item loop and 54 registers
    // spill (rolled 56.9, by 2 54.4, by 4 53.8 us per 8192 images).  Issuing the reads of step t + 1 before the MFMAs of step t by hand:
    // 59.6 us -- the phases do not wait for these reads.
    auto wg_step = [&](int t, const unsigned short* s_d, const unsigned short* s_src, const int a0, bool bias) {
        int orow[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) orow[h] = porg(32 * t + 16 * (kq >> 1) + 8 * h + 4 * (kq & 1) + rq) + 4 * cp;
        auto tr = [&](const unsigned short* base, int off) {
            const rb_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((rb_lds_s16x4_ptr)(base + orow[0] + off));
            const rb_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((rb_lds_s16x4_ptr)(base + orow[1] + off));
            return (bf16x8)__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        };
        bf16x8 d[2], b[QM];
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) d[cb] = tr(s_d, CENTER + cb * 16);
#pragma unroll
        for (int qq = 0; qq < QM; ++qq) {
            const int q = rw + 4 * (qq < 4 ? qq : (qcnt > 4 ? 4 : 0)), tap = q >> 1, ib = q & 1;
            b[qq] = tr(s_src, koffc(tap) + ib * 16);
        }
        asm volatile("" ::: "memory");
        if (bias) { st[20] = MFMA_BF16(d[0], ones, st[20]); st[21] = MFMA_BF16(d[1], ones, st[21]); }
#pragma unroll
        for (int qq = 0; qq < QM; ++qq)
            if (qq < qcnt) {
#pragma unroll
                for (int cb = 0; cb < 2; ++cb) st[a0 + 2 * qq + cb] = MFMA_BF16(d[cb], b[qq], st[a0 + 2 * qq + cb]);
            }
    };

    if (!conv_role && (int)blockIdx.x < nwork) load(blockIdx.x);
    for (int work = blockIdx.x; work < nwork; work += gridDim.x) {
        const int img0 = work * C::NIMG, left = a.n - img0;           // images of this item that exist
        __syncthreads();
        RTCK(3);
        if (!conv_role) {
#pragma unroll
            for (int k = 0; k < C::KW; ++k) {
                const int e = t2 + k * 256, pl = e >> 2;
                const int o = porg(pl) + CENTER + (e & 3) * 8;
                const bool on = (pl >> 6) < left;
                const uint4 z = {0u, 0u, 0u, 0u};
                const uint4 vx = on ? __builtin_bit_cast(uint4, st[22 + k]) : z, va = on ? __builtin_bit_cast(uint4, st[22 + C::KW + k]) : z,
                            vp = on ? __builtin_bit_cast(uint4, st[22 + 2 * C::KW + k]) : z;
                *(uint4*)(s_x + o) = vx;
                *(uint4*)(s_a + o) = (uint4){rb_relu2(va.x), rb_relu2(va.y), rb_relu2(va.z), rb_relu2(va.w)};
                *(uint4*)(s_p + o) = (uint4){rb_relu2(vp.x), rb_relu2(vp.y), rb_relu2(vp.z), rb_relu2(vp.w)};
            }
        }
        __syncthreads();
        RTCK(0);
        if (!conv_role && work + (int)gridDim.x < nwork) load(work + gridDim.x);

        if (conv_role) {
            // ---- da = convT2(dy) * (a > 0) on the images' own rows -> s_y: tiles rw, rw + 4, rw + 8, rw + 12, one at a time (weights in registers)
            for (int t = rw; t < C::NMT; t += 4) {
                const int org = porg(t * 16 + i);
                const unsigned short* src = s_x + org + kq * 8;
                const int yb = org + CENTER + kq * 4;
                bf16x8 av[5];
#pragma unroll
                for (int m = 0; m < 5; ++m) av[m] = *(const bf16x8*)(src + koffc(m));
                const uint2 mk0 = *(const uint2*)(s_a + yb), mk1 = *(const uint2*)(s_a + yb + 16);
                asm volatile("" ::: "memory");
                f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
                for (int m = 0; m < 9; ++m) {
                    const bf16x8 cur = av[m % 5];
                    acc[0] = MFMA_BF16(__builtin_bit_cast(bf16x8, st[2 * m]), cur, acc[0]);
                    acc[1] = MFMA_BF16(__builtin_bit_cast(bf16x8, st[2 * m + 1]), cur, acc[1]);
                    if (m < 4) av[m] = *(const bf16x8*)(src + koffc(m + 5));
                }
#pragma unroll
                for (int nb = 0; nb < 2; ++nb) {
                    const uint2 mk = nb ? mk1 : mk0;
                    float v[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = rb_lane(mk, r) > 0.f ? acc[nb][r] : 0.f;
                    const uint2 raw = rb_pack(v);
                    *(uint2*)(s_y + yb + nb * 16) = raw;
                    if (a.da_out) {
                        const int pl = t * 16 + i;
                        if ((pl >> 6) < left) *(uint2*)(a.da_out + ((long long)img0 * 64 + pl) * C::C + nb * 16 + kq * 4) = raw;
                    }
                }
            }
        } else {
            // ---- conv2's weight / bias gradient from (dy, relu(a))
#pragma unroll 4
            for (int t = 0; t < C::NSTEP; ++t) wg_step(t, s_x, s_a, 0, rw == 2);
        }
        RTCK(1);
        __syncthreads();
        RTCK(2);
        if (conv_role) {
            // ---- dx = convT1(da) * (x > 0) + dy -> HBM
            for (int t = rw; t < C::NMT; t += 4) {
                const int pl = t * 16 + i, org = porg(pl);
                const unsigned short* src = s_y + org + kq * 8;
                const int eo = org + CENTER + kq * 4;
                bf16x8 av[5];
#pragma unroll
                for (int m = 0; m < 5; ++m) av[m] = *(const bf16x8*)(src + koffc(m));
                const uint2 mk0 = *(const uint2*)(s_p + eo), mk1 = *(const uint2*)(s_p + eo + 16);
                const uint2 sk0 = *(const uint2*)(s_x + eo), sk1 = *(const uint2*)(s_x + eo + 16);
                asm volatile("" ::: "memory");
                f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
                for (int m = 0; m < 9; ++m) {
                    const bf16x8 cur = av[m % 5];
                    acc[0] = MFMA_BF16(__builtin_bit_cast(bf16x8, st[18 + 2 * m]), cur, acc[0]);
                    acc[1] = MFMA_BF16(__builtin_bit_cast(bf16x8, st[18 + 2 * m + 1]), cur, acc[1]);
                    if (m < 4) av[m] = *(const bf16x8*)(src + koffc(m + 5));
                }
                if ((pl >> 6) < left) {
#pragma unroll
                    for (int nb = 0; nb < 2; ++nb) {
                        const uint2 mk = nb ? mk1 : mk0, sk = nb ? sk1 : sk0;
                        float v[4];
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = (rb_lane(mk, r) > 0.f ? acc[nb][r] : 0.f) + rb_lane(sk, r);
                        *(uint2*)(a.dx_out + ((long long)img0 * 64 + pl) * C::C + nb * 16 + kq * 4) = rb_pack(v);
                    }
                }
            }
        } else {
            // ---- conv1's weight / bias gradient from (da, relu(x))
#pragma unroll 4
            for (int t = 0; t < C::NSTEP; ++t) wg_step(t, s_y, s_p, 10, rw == 3);
        }
    }
#ifdef WG_TIMING
    if ((tid & 255) == 0) for (int q = 0; q < 4; ++q) atomicAdd(&g_rb_timing[(tid >> 8) * 4 + q], (unsigned long long)tacc_[q]);
#endif
    if (!conv_role) {
        float* sl2 = a.slab2 + (long long)blockIdx.x * C::SLAB;
        float* sl1 = a.slab1 + (long long)blockIdx.x * C::SLAB;
#pragma unroll
        for (int qq = 0; qq < QM; ++qq)
            if (qq < qcnt) {
                const int q = rw + 4 * qq, tap = q >> 1, ib = q & 1;
#pragma unroll
                for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int o = ((cb * 16 + kq * 4 + r) * 9 + tap) * C::C + ib * 16 + i;
                        sl2[o] = st[2 * qq + cb][r]; sl1[o] = st[10 + 2 * qq + cb][r];
                    }
            }
        if (i == 0 && rw >= 2) {
            float* sl = rw == 2 ? sl2 : sl1;
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                for (int r = 0; r < 4; ++r) sl[C::WLEN + cb * 16 + kq * 4 + r] = st[20 + cb][r];
        }
    }
}
#ifndef RB32_QUAD8
#define RB32_QUAD8 1               // 8x8 blocks: 1 = resblock_bwd_full32q_bf16_kernel (four images per item), 0 = one image per item (kernel above)
#endif
static int rb_full32q_grid(int n) { const int w = (n + RbFull32Q::NIMG - 1) / RbFull32Q::NIMG; return w > 256 ? 256 : w; }
static void launch_rb_full32q(const RbFullArgs& a, hipStream_t st) {
    static std::once_flag attr;
    std::call_once(attr, [] { hipFuncSetAttribute((const void*)resblock_bwd_full32q_bf16_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)RbFull32Q::LDS_BYTES); });
    const int grid = rb_full32q_grid(a.n);
    if (grid < 1) return;
    hipLaunchKernelGGL(resblock_bwd_full32q_bf16_kernel, dim3(grid), dim3(512), RbFull32Q::LDS_BYTES, st, a);
}

template <class C>
static int rb_full32_grid_t(int n) {
    int bpc = (int)((160 * 1024) / C::LDS_BYTES);
    bpc = bpc < 1 ? 1 : (bpc > 4 ? 4 : bpc);
    const int w = n * C::TPI;
    return w > 256 * bpc ? 256 * bpc : w;
}
#ifndef RB32_SPECIALISED
#define RB32_SPECIALISED 1
#endif
static int rb_full32s_grid(int n);
int resblock_bwd_full32_grid(ConvShape s, int n) { return s == CS_32_32_16 ? (RB32_SPECIALISED ? rb_full32s_grid(n) : rb_full32_grid_t<RbFull32>(n)) : s == CS_32_32_8 ? (RB32_QUAD8 ? rb_full32q_grid(n) : rb_full32_grid_t<RbFull32S>(n)) : -1; }
template <class C>
static void launch_rb_full32_t(const RbFullArgs& a, hipStream_t st) {
    static std::once_flag attr;          // (launchers run on up to 4 group worker threads)
    std::call_once(attr, [] { hipFuncSetAttribute((const void*)resblock_bwd_full32_bf16_kernel<C>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::LDS_BYTES); });
    const int grid = rb_full32_grid_t<C>(a.n);
    if (grid < 1) return;
    hipLaunchKernelGGL(resblock_bwd_full32_bf16_kernel<C>, dim3(grid), dim3(C::NT), C::LDS_BYTES, st, a);
}
#ifndef RB32_SPECIALISED
#define RB32_SPECIALISED 1
#endif
static int rb_full32s_grid(int n) { const int w = n * RbFull32W::TPI; return w > 256 ? 256 : w; }      // one 512-thread workgroup per CU
static void launch_rb_full32s(const RbFullArgs& a, hipStream_t st) {
    using C = RbFull32W;
    constexpr size_t LDS = (size_t)(C::X_ELEMS + 3 * C::Y_ELEMS) * 2;          // no bank copies in LDS
    static std::once_flag attr;          // (launchers run on up to 4 group worker threads)
    std::call_once(attr, [] { hipFuncSetAttribute((const void*)resblock_bwd_full32s_bf16_kernel<C>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS); });
    const int grid = rb_full32s_grid(a.n);
    if (grid < 1) return;
    hipLaunchKernelGGL(resblock_bwd_full32s_bf16_kernel<C>, dim3(grid), dim3(512), LDS, st, a);
}
// 32-channel residual blocks @16x16 (CS_32_32_16) and @8x8 (CS_32_32_8).  slab2 / slab1: [grid][9248] floats each.
void launch_resblock_bwd_full32_bf16(ConvShape s, const void* dy, const void* a_fwd, const void* x_fwd, void* dx_out, void* da_out, int n,
                                     const unsigned short* bank2_t, const unsigned short* bank1_t, float* slab2, float* slab1, hipStream_t st) {
    RbFullArgs a{(const unsigned short*)dy, (const unsigned short*)a_fwd, (const unsigned short*)x_fwd, (unsigned short*)dx_out, (unsigned short*)da_out,
                 bank2_t, bank1_t, slab2, slab1, n};
    if (s == CS_32_32_16) { if (RB32_SPECIALISED) launch_rb_full32s(a, st); else launch_rb_full32_t<RbFull32>(a, st); }
    else if (s == CS_32_32_8) { if (RB32_QUAD8) launch_rb_full32q(a, st); else launch_rb_full32_t<RbFull32S>(a, st); }
}

// res1 + res2 of a block forward in one launch.  b / bank: res1.conv1, res1.conv2, res2.conv1, res2.conv2 (forward banks).
void launch_resblock_pair_bf16(ConvShape s, const void* x, const float* const* b, void* a1_out, void* y1_out, void* a2_out, void* y2_out, int n,
                               const unsigned short* const* bank, hipStream_t st) {
    ResblockPairArgs a{(const unsigned short*)x, {b[0], b[1], b[2], b[3]}, (unsigned short*)a1_out, (unsigned short*)y1_out, (unsigned short*)a2_out,
                       (unsigned short*)y2_out, n, {bank[0], bank[1], bank[2], bank[3]}};
    switch (s) {
        case CS_16_16_32: launch_rbp_t<RB_16_32>(a, st); break;
        case CS_32_32_16: if (RB32_PAIR_ROLES && n >= 1024) launch_rbp32r<RBR_32_16>(a, st); else launch_rbp_t<RB_32_16>(a, st); break;
        case CS_32_32_8:  if (n <= 1024) launch_rbp_t<RB_32_8S>(a, st); else if (RB32_PAIR_ROLES & 2) launch_rbp32r<RBR_32_8>(a, st); else launch_rbp_t<RB_32_8P>(a, st); break;
        default: break;
    }
}
