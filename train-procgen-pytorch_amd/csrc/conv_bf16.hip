// bf16 storage / bf16 matrix-core variant of the 3x3 convolutions (forward + data gradient) for the 16- and
// 32-channel layers of the IMPALA-CNN (BASELINE config 3: "IMPALA-CNN bf16").  Activations and activation
// gradients live in HBM as bf16 NHWC (half the bytes of the fp32 path); the fp32 master weights are rounded to
// bf16 while the filter bank is staged into LDS; products accumulate in fp32 (v_mfma_f32_16x16x32_bf16);
// bias / ReLU-mask / residual are applied in fp32 and the result is rounded to bf16 once, at the store.
//
// Same decomposition as conv.hip (persistent workgroups, register-prefetched haloed tiles, epilogue operands
// requested before the MFMA phase).  One MFMA consumes K = 32: for 32 input channels that is one filter tap
// (lane quarter kq owns channels 8kq..8kq+7); for 16 input channels it is two taps (kq>>1 selects the tap,
// kq&1 the 8-channel half), the 10th "tap" being a zero column of the filter bank.
#include "common.h"
#include <mutex>
#include <type_traits>

static int c1_grid(int n);          // persistent grid of the block1.conv weight-gradient kernel (defined with it)
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef s16x4 __attribute__((address_space(3))) * lds_s16x4_ptr;      // ds_read_b64_tr_b16 operand
#define MFMA_BF16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16((a), (b), (c), 0, 0, 0)

__device__ __forceinline__ unsigned short f2bf(float x) {
    __bf16 h = (__bf16)x;                                   // v_cvt_pk_bf16_f32: round to nearest even, NaN stays NaN
    return __builtin_bit_cast(unsigned short, h);
}
__device__ __forceinline__ float bf2f(unsigned short h) { return __uint_as_float(((unsigned)h) << 16); }

// ------------------------------------------------------------------------------------------ max-pool backward as an operand stage
// The gradient of a block's first conv output is never stored: its consumers (the conv's data-gradient and weight-gradient
// kernels) rebuild the window they need in LDS from the POOLED gradient and the arg-max bytes MaxPool2d(3,2,1) left.
// Staged: NPR pooled rows (first = oyb) x HO cols x CH channels, gradient as bf16 + arg-max as bytes; rows outside the
// pooled map carry arg 0xff (matches no position).  A thread then owns a 2x2 block of conv pixels x 4 channels: the 4
// windows (oy, ox) in {a, a+1} x {b, b+1} touching the block are read once each, their 9 (window, pixel) incidences -- one
// per pool position -- use compile-time (ky, kx); windows are added in (oy, ox) order, the order of the stand-alone kernel.
// Conv rows outside the image come out as zeros by themselves (no window's arg-max points outside the image).
template <int CH, int HO, int NPR>
struct PoolStage {
    static constexpr int NW = NPR * HO * (CH / 8), K = (NW + 255) / 256;      // 16-byte gradient words (+ 8-byte arg words) per item
    static constexpr int PD_ELEMS = NPR * HO * CH;                          // bf16 elements; the arg bytes follow
    static constexpr size_t BYTES = (size_t)PD_ELEMS * 3;
    uint4 rd[K]; uint2 ra[K];
    __device__ __forceinline__ void load(const unsigned short* g_dp, const uint8_t* g_arg, long long img, int oyb) {
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int e = threadIdx.x + k * 256;
            rd[k] = (uint4){0u, 0u, 0u, 0u}; ra[k] = (uint2){0xffffffffu, 0xffffffffu};
            if (e < NW) {
                const int c8 = e % (CH / 8), ox = (e / (CH / 8)) % HO, oy = oyb + e / ((CH / 8) * HO);
                if (oy >= 0 && oy < HO) {
                    const long long o = ((img * HO + oy) * HO + ox) * CH + c8 * 8;
                    rd[k] = *(const uint4*)(g_dp + o); ra[k] = *(const uint2*)(g_arg + o);
                }
            }
        }
    }
    __device__ __forceinline__ void store(unsigned short* s_pd) const {
        uint8_t* s_pa = (uint8_t*)(s_pd + PD_ELEMS);
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int e = threadIdx.x + k * 256;
            if (e < NW) { *(uint4*)(s_pd + e * 8) = rd[k]; *(uint2*)(s_pa + e * 8) = ra[k]; }
        }
    }
    // block rows a0 .. a0+NPR-2 (staged pooled row 0 = a0); writes conv rows y in [y_lo, y_hi) to
    // s_dst[((y - y_base) * dst_pw + x + xoff) * S + channel]
    static __device__ __forceinline__ void gather(const unsigned short* s_pd, int a0, int y_lo, int y_hi, unsigned short* s_dst, int y_base,
                                                  int dst_pw, int xoff, int S) {
        for (int task = threadIdx.x; task < NTASK; task += 256) gather_task(s_pd, task, a0, y_lo, y_hi, s_dst, y_base, dst_pw, xoff, S);
    }
    static constexpr int NTASK = (NPR - 1) * HO * (CH / 4);               // 2x2-pixel blocks x 4-channel groups of one staged item
    static __device__ __forceinline__ void gather_task(const unsigned short* s_pd, int task, int a0, int y_lo, int y_hi, unsigned short* s_dst,
                                                       int y_base, int dst_pw, int xoff, int S) {
        const uint8_t* s_pa = (const uint8_t*)(s_pd + PD_ELEMS);
        constexpr int Q = CH / 4;
        {
            const int cq = task % Q, bx = (task / Q) % HO, bl = task / (Q * HO);
            float sm[4][4];
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int c = 0; c < 4; ++c) sm[q][c] = 0.f;
#pragma unroll
            for (int wy = 0; wy < 2; ++wy)
#pragma unroll
                for (int wx = 0; wx < 2; ++wx) {
                    const int ox = bx + wx;
                    const int o = ((bl + wy) * HO + (ox < HO ? ox : HO - 1)) * CH + cq * 4;
                    const uint2 d = *(const uint2*)(s_pd + o);
                    unsigned ag = *(const unsigned*)(s_pa + o);
                    if (ox >= HO) ag = 0xffffffffu;                            // window column HO does not exist
                    const float v[4] = {__uint_as_float(d.x << 16), __uint_as_float(d.x & 0xffff0000u), __uint_as_float(d.y << 16), __uint_as_float(d.y & 0xffff0000u)};
#pragma unroll
                    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                        for (int dx = 0; dx < 2; ++dx) {
                            const int ky = dy + 1 - 2 * wy, kx = dx + 1 - 2 * wx;
                            if (ky < 0 || kx < 0) continue;
                            const unsigned pos = (unsigned)(ky * 3 + kx);
#pragma unroll
                            for (int c = 0; c < 4; ++c)
                                sm[dy * 2 + dx][c] += (((ag >> (8 * c)) & 0xffu) == pos) ? v[c] : 0.f;
                        }
                }
#pragma unroll
            for (int dy = 0; dy < 2; ++dy) {
                const int y = 2 * (a0 + bl) + dy;
                if (y < y_lo || y >= y_hi) continue;
#pragma unroll
                for (int dx = 0; dx < 2; ++dx) {
                    const float* q = sm[dy * 2 + dx];
                    *(uint2*)(s_dst + ((y - y_base) * dst_pw + 2 * bx + dx + xoff) * S + cq * 4) =
                        (uint2){mi_pk_bf16(q[0], q[1]), mi_pk_bf16(q[2], q[3])};
                }
            }
        }
    }
    // the same sums for a 2x2-pixel block x 2 channels (twice the tasks: an even deal over 512 threads, fewer live registers)
    static constexpr int NTASK2 = (NPR - 1) * HO * (CH / 2);
    static __device__ __forceinline__ void gather_task2(const unsigned short* s_pd, int task, int a0, int y_lo, int y_hi, unsigned short* s_dst,
                                                        int y_base, int dst_pw, int xoff, int S) {
        const uint8_t* s_pa = (const uint8_t*)(s_pd + PD_ELEMS);
        constexpr int Q = CH / 2;
        const int cq = task % Q, bx = (task / Q) % HO, bl = task / (Q * HO);
        float sm[4][2];
#pragma unroll
        for (int q = 0; q < 4; ++q) { sm[q][0] = 0.f; sm[q][1] = 0.f; }
#pragma unroll
        for (int wy = 0; wy < 2; ++wy)
#pragma unroll
            for (int wx = 0; wx < 2; ++wx) {
                const int ox = bx + wx;
                const int o = ((bl + wy) * HO + (ox < HO ? ox : HO - 1)) * CH + cq * 2;
                const unsigned d = *(const unsigned*)(s_pd + o);
                unsigned ag = *(const unsigned short*)(s_pa + o);
                if (ox >= HO) ag = 0xffffu;                                    // window column HO does not exist
                const float v[2] = {__uint_as_float(d << 16), __uint_as_float(d & 0xffff0000u)};
#pragma unroll
                for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                    for (int dx = 0; dx < 2; ++dx) {
                        const int ky = dy + 1 - 2 * wy, kx = dx + 1 - 2 * wx;
                        if (ky < 0 || kx < 0) continue;
                        const unsigned pos = (unsigned)(ky * 3 + kx);
#pragma unroll
                        for (int c = 0; c < 2; ++c)
                            sm[dy * 2 + dx][c] += (((ag >> (8 * c)) & 0xffu) == pos) ? v[c] : 0.f;
                    }
            }
#pragma unroll
        for (int dy = 0; dy < 2; ++dy) {
            const int y = 2 * (a0 + bl) + dy;
            if (y < y_lo || y >= y_hi) continue;
#pragma unroll
            for (int dx = 0; dx < 2; ++dx)
                *(unsigned*)(s_dst + ((y - y_base) * dst_pw + 2 * bx + dx + xoff) * S + cq * 2) = mi_pk_bf16(sm[dy * 2 + dx][0], sm[dy * 2 + dx][1]);
        }
    }
};

template <int CIN_, int COUT_, int HW_, int TH_, int TW_, int NIMG_, bool TRANSW_>
struct BfCfg {
    static constexpr int CIN = CIN_, COUT = COUT_, HW = HW_, TH = TH_, TW = TW_, NIMG = NIMG_;
    static constexpr bool TRANSW = TRANSW_;
    // bf16 elements per staged pixel / per filter row: strides for which every ds_read_b128 lane group is
    // bank-conflict free (brute-forced over the gfx950 lane-group / 64-bank rule; CIN+8 gave 2-way conflicts)
    static constexpr int S = (CIN == 16) ? 16 : 48;
    static constexpr int PH = TH + 2, PW = TW + 2;
    static constexpr int NPIX = NIMG * PH * PW;
    static constexpr int IN_ELEMS = ((NPIX * S + 7) / 8) * 8;
    static constexpr int NK = (CIN == 32) ? 9 : 5;           // MFMAs (K = 32 each) per (pixel tile, channel block)
    static constexpr int WS = NK * 32 + 16;                  // bf16 elements per output channel of the filter bank
    static constexpr int W_ELEMS = COUT * WS;
    static constexpr int NMT = NIMG * TH * TW / 16, MT = NMT / 4, NB = COUT / 16;
    static constexpr int TPI_X = HW / TW, TPI = (HW / TH) * (HW / TW);
    static constexpr int C8 = CIN / 8;
    static constexpr int NLD = (NPIX * C8 + 255) / 256;
    static constexpr size_t LDS_BYTES = (size_t)(IN_ELEMS + W_ELEMS) * 2;
#ifndef BF_WPE_32_16
#define BF_WPE_32_16 1
#endif
    // waves per SIMD the register allocation must leave room for.  The 32-channel 16x16 tiles fit 3 workgroups per CU by LDS
    // (50.6 KB) but compile to 196 registers = 2 waves per SIMD; forcing 3 spills 76 bytes and measured slower (12.3 vs 10.2 ms)
    static constexpr int WPE = (CIN == 32 && COUT == 32 && HW == 16 && NIMG == 1) ? BF_WPE_32_16 : 1;
    static_assert(NMT % 4 == 0, "M tiles must split over 4 waves");
};

template <class C>
__device__ __forceinline__ void bf_coords(int work, int& img0, int& ty0, int& tx0) {
    if (C::NIMG > 1) { img0 = work * C::NIMG; ty0 = 0; tx0 = 0; }
    else { img0 = work / C::TPI; const int t = work % C::TPI; ty0 = (t / C::TPI_X) * C::TH; tx0 = (t % C::TPI_X) * C::TW; }
}

template <class C>
__device__ __forceinline__ void bf_tile_load(uint4 (&r)[C::NLD], const unsigned short* in, int n_img, int img0, int ty0, int tx0) {
#pragma unroll
    for (int k = 0; k < C::NLD; ++k) {
        const int e = threadIdx.x + k * 256;
        uint4 v = {0u, 0u, 0u, 0u};
        if (e < C::NPIX * C::C8) {
            const int pix = e / C::C8, c8 = e % C::C8;
            const int img = pix / (C::PH * C::PW), q = pix % (C::PH * C::PW);
            const int gy = ty0 + q / C::PW - 1, gx = tx0 + q % C::PW - 1, n = img0 + img;
            if (n < n_img && gy >= 0 && gy < C::HW && gx >= 0 && gx < C::HW)
                v = *(const uint4*)(in + (((long long)n * C::HW + gy) * C::HW + gx) * C::CIN + c8 * 8);
        }
        r[k] = v;
    }
}
__device__ __forceinline__ unsigned relu_bf16x2(unsigned w) {         // zero every 16-bit lane whose sign bit is set
    const unsigned neg = (w >> 15) & 0x00010001u;
    return w & ~(neg * 0xFFFFu);
}
template <class C>
__device__ __forceinline__ void bf_tile_store(const uint4 (&r)[C::NLD], unsigned short* s_in, int relu_in) {
#pragma unroll
    for (int k = 0; k < C::NLD; ++k) {
        const int e = threadIdx.x + k * 256;
        if (e < C::NPIX * C::C8) {
            uint4 v = r[k];
            if (relu_in) { v.x = relu_bf16x2(v.x); v.y = relu_bf16x2(v.y); v.z = relu_bf16x2(v.z); v.w = relu_bf16x2(v.w); }
            *(uint4*)(s_in + (e / C::C8) * C::S + (e % C::C8) * 8) = v;
        }
    }
}

// POOLIN (data gradient of a block's first conv): a.in is the POOLED gradient, a.pool_arg the arg-max bytes; the haloed
// tile of the conv-output gradient is rebuilt in LDS by PoolStage::gather instead of being loaded.
// (block2.conv, whose weight gradient shares this gather, has its own kernel below: block2_conv_bwd_bf16_kernel)
template <class C, bool POOLIN = false>
__global__ __launch_bounds__(256, C::WPE) void conv3x3_bf16_kernel(ConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned short smem_h[];
    unsigned short* s_in = smem_h;
    unsigned short* s_w = smem_h + C::IN_ELEMS;
    using PS = PoolStage<C::CIN, C::HW / 2, C::TH / 2 + 3>;                // pooled rows ty0/2 - 1 .. ty0/2 + TH/2 + 1
    unsigned short* s_pd = s_w + C::W_ELEMS;
    static_assert(!POOLIN || (C::NIMG == 1 && C::TW == C::HW && C::TH % 2 == 0), "pooled input: full-width row tiles of one image");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 15, kq = lane >> 4;
    const unsigned short* g_in = (const unsigned short*)a.in;
    const unsigned short* g_mask = (const unsigned short*)a.mask;
    const unsigned short* g_res = (const unsigned short*)a.res;
    unsigned short* g_out = (unsigned short*)a.out;

    // filter bank -> LDS [co][tap-major K][ci] in bf16 (dgrad: co<->ci swapped, taps mirrored); K padded with zeros.
    // Normally a straight 16-byte copy of the image pack_banks_kernel refreshed after the last optimizer step (the
    // per-element fp32 gather + convert below costs ~5 us per launch, which dominated the n = n_envs rollout launches).
    if (a.wbank) {
        for (int e = tid; e < C::COUT * C::WS / 8; e += 256) ((uint4*)s_w)[e] = ((const uint4*)a.wbank)[e];
    } else
    for (int e = tid; e < C::COUT * C::WS; e += 256) {
        const int j = e / C::WS, k = e % C::WS;
        float v = 0.f;
        const int tap = k / C::CIN, ci = k % C::CIN;
        if (k < C::NK * 32 && tap < 9) v = C::TRANSW ? a.w[(ci * 9 + (8 - tap)) * C::COUT + j] : a.w[(j * 9 + tap) * C::CIN + ci];
        s_w[e] = f2bf(v);
    }

    // The MFMA runs with the FILTER rows as its A operand and the pixels as B: a lane's 4 accumulator registers are then 4
    // CONSECUTIVE output channels (4*kq .. 4*kq+3 of the block) of ONE pixel (lane & 15), so mask / residual / output move
    // as 8-byte words -- a wave's store covers whole 32- or 64-byte pixels back to back.
    float bias_r[C::NB][4];
#pragma unroll
    for (int nb = 0; nb < C::NB; ++nb)
#pragma unroll
        for (int r = 0; r < 4; ++r) bias_r[nb][r] = a.bias ? a.bias[nb * 16 + kq * 4 + r] : 0.f;
    // per-lane A offsets of the NK K-steps: tap offset + 8-channel chunk
    int koff[C::NK];
#pragma unroll
    for (int m = 0; m < C::NK; ++m) {
        int tap, chunk;
        if (C::CIN == 32) { tap = m; chunk = kq; } else { tap = 2 * m + (kq >> 1); chunk = kq & 1; if (tap > 8) tap = 8; }
        koff[m] = ((tap / 3) * C::PW + (tap % 3)) * C::S + chunk * 8;
    }

    const int nwork = (C::NIMG > 1) ? (a.n + C::NIMG - 1) / C::NIMG : a.n * C::TPI;
    uint4 regs[POOLIN ? 1 : C::NLD];
    PS ps;
    int img0, ty0, tx0;
    if (POOLIN) for (int e = tid; e < C::IN_ELEMS / 8; e += 256) ((uint4*)s_in)[e] = (uint4){0u, 0u, 0u, 0u};      // halo columns stay zero
    if ((int)blockIdx.x < nwork) {
        bf_coords<C>(blockIdx.x, img0, ty0, tx0);
        if constexpr (POOLIN) ps.load(g_in, a.pool_arg, img0, ty0 / 2 - 1); else bf_tile_load<C>(regs, g_in, a.n, img0, ty0, tx0);
    }
    for (int work = blockIdx.x; work < nwork; work += gridDim.x) {
        bf_coords<C>(work, img0, ty0, tx0);
        __syncthreads();
        if constexpr (POOLIN) {
            ps.store(s_pd);
            __syncthreads();
            PS::gather(s_pd, ty0 / 2 - 1, ty0 - 1, ty0 + C::TH + 1, s_in, ty0 - 1, C::PW, 1, C::S);
        } else bf_tile_store<C>(regs, s_in, a.relu_in);
        __syncthreads();
        if (work + (int)gridDim.x < nwork) {
            int i2, y2, x2;
            bf_coords<C>(work + gridDim.x, i2, y2, x2);
            if constexpr (POOLIN) ps.load(g_in, a.pool_arg, i2, y2 / 2 - 1); else bf_tile_load<C>(regs, g_in, a.n, i2, y2, x2);
        }
        // epilogue operands requested before the MFMA phase (one wave-uniform branch per block of loads)
        uint2 e_mask[C::MT][C::NB], e_res[C::MT][C::NB];
        long long e_off[C::MT];
        bool e_on[C::MT];
#pragma unroll
        for (int mt = 0; mt < C::MT; ++mt) {
            const int pl = (wave * C::MT + mt) * 16 + i, y = pl / C::TW, x = pl % C::TW;
            int n = img0 + y / C::TH;
            e_on[mt] = n < a.n;
            n = n < a.n ? n : a.n - 1;
            e_off[mt] = (((long long)n * C::HW + ty0 + (y % C::TH)) * C::HW + tx0 + x) * C::COUT + kq * 4;
        }
        if (g_mask) {
#pragma unroll
            for (int mt = 0; mt < C::MT; ++mt)
#pragma unroll
                for (int nb = 0; nb < C::NB; ++nb) e_mask[mt][nb] = *(const uint2*)(g_mask + e_off[mt] + nb * 16);
        } else {
#pragma unroll
            for (int mt = 0; mt < C::MT; ++mt)
#pragma unroll
                for (int nb = 0; nb < C::NB; ++nb) e_mask[mt][nb] = (uint2){0x3f803f80u, 0x3f803f80u};      // 1.0
        }
        if (g_res) {
#pragma unroll
            for (int mt = 0; mt < C::MT; ++mt)
#pragma unroll
                for (int nb = 0; nb < C::NB; ++nb) e_res[mt][nb] = *(const uint2*)(g_res + e_off[mt] + nb * 16);
        } else {
#pragma unroll
            for (int mt = 0; mt < C::MT; ++mt)
#pragma unroll
                for (int nb = 0; nb < C::NB; ++nb) e_res[mt][nb] = (uint2){0u, 0u};
        }

        f32x4 acc[C::MT][C::NB];
#pragma unroll
        for (int mt = 0; mt < C::MT; ++mt)
#pragma unroll
            for (int nb = 0; nb < C::NB; ++nb) acc[mt][nb] = (f32x4){0.f, 0.f, 0.f, 0.f};
        int abase[C::MT];
#pragma unroll
        for (int mt = 0; mt < C::MT; ++mt) {
            const int pl = (wave * C::MT + mt) * 16 + i, y = pl / C::TW, x = pl % C::TW;
            abase[mt] = (((y / C::TH) * C::PH + (y % C::TH)) * C::PW + x) * C::S;
        }
        const int bbase = i * C::WS + kq * 8;
#pragma unroll
        for (int m = 0; m < C::NK; ++m) {
            bf16x8 av[C::MT], bv[C::NB];
#pragma unroll
            for (int mt = 0; mt < C::MT; ++mt) av[mt] = *(const bf16x8*)(s_in + abase[mt] + koff[m]);
#pragma unroll
            for (int nb = 0; nb < C::NB; ++nb) bv[nb] = *(const bf16x8*)(s_w + bbase + nb * 16 * C::WS + m * 32);
#pragma unroll
            for (int mt = 0; mt < C::MT; ++mt)
#pragma unroll
                for (int nb = 0; nb < C::NB; ++nb) acc[mt][nb] = MFMA_BF16(bv[nb], av[mt], acc[mt][nb]);
        }
#pragma unroll
        for (int mt = 0; mt < C::MT; ++mt)
            if (e_on[mt]) {
#pragma unroll
                for (int nb = 0; nb < C::NB; ++nb) {
                    const unsigned mw[2] = {e_mask[mt][nb].x, e_mask[mt][nb].y}, rw[2] = {e_res[mt][nb].x, e_res[mt][nb].y};
                    float v[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float mk = (r & 1) ? __uint_as_float(mw[r >> 1] & 0xffff0000u) : __uint_as_float(mw[r >> 1] << 16);
                        const float rs = (r & 1) ? __uint_as_float(rw[r >> 1] & 0xffff0000u) : __uint_as_float(rw[r >> 1] << 16);
                        v[r] = acc[mt][nb][r] + bias_r[nb][r];
                        v[r] = mk > 0.f ? v[r] : 0.f;
                        v[r] += rs;
                    }
                    *(uint2*)(g_out + e_off[mt] + nb * 16) = (uint2){mi_pk_bf16(v[0], v[1]), mi_pk_bf16(v[2], v[3])};
                }
            }
    }
}

// ------------------------------------------------------------------------------------------ block2.conv: whole backward in one launch
// Data gradient AND weight / bias gradient of block2.conv (16 -> 32 channels @32x32) from the POOLED output gradient + arg-max
// bytes, 512 threads per workgroup and two workgroups per CU = 4 waves per SIMD.  The 256-thread fused variant of the generic
// kernel above measured 326 us per 8192 samples with every pipe under 55 % busy (scratch/kbench_cb.hip phase clocks: gather 32 %,
// weight-gradient MFMAs 25 %, data-gradient MFMAs 20 % of an item, each phase latency-bound at 2 waves per SIMD), so this kernel
// keeps the same LDS tiles and arithmetic but (a) doubles the waves per tile, (b) gives the staging / next-item loads to waves 4-7,
// which own one gather task against two for waves 0-3, (c) addresses every global access as item base (scalar) + per-thread
// constant offset, (d) splits the 20 weight-gradient accumulator tiles of a pixel step over a wave PAIR (taps 0-4 | taps 5-8 + bias)
// so a wave holds 10 tiles instead of 20 (128-register budget).
// Item = 8 conv rows x 32 columns of one image.  LDS: s_in [10][34][48] gathered conv-output gradient, s_w data-gradient filter
// bank, s_pd 7 pooled rows (bf16 + arg bytes), s_xi [10][34][16] forward input.  Data gradient: same operand layout and K order as
// the generic kernel -> bit-identical dX.  Weight gradient: pixel steps {2p, 2p+1} belong to the wave pair (p, p+4); partial sums
// leave through LDS in fixed wave order, one slab per workgroup.
struct B2Bwd {
    using C = BfCfg<32, 16, 32, 8, 32, 1, true>;
    using PS = PoolStage<32, 16, 6>;                                        // pooled rows 4k .. 4k+5 of item k
    static constexpr int NT = 512;
    static constexpr int IN_ROWS = C::PH + 1, IN_ELEMS = IN_ROWS * C::PW * C::S;       // conv rows ty0-1 .. ty0+9
    static constexpr int ROW_WORDS = C::PW * C::S / 8;                      // 16-byte words of one staged gradient row
    static constexpr int SX = 16, XI_ELEMS = C::PH * C::PW * SX;
    static constexpr int NXI = C::PH * C::TW * 2, KXI = (NXI + NT - 1) / NT;           // 16-byte words of the forward-input tile (interior columns)
    static constexpr int WLEN = 32 * 9 * 16, SLAB = WLEN + 32;
    static constexpr int ONES_ELEMS = (C::PW + 8 + 1) * SX + 16;           // bf16 1.0s covering every offset a slot's four operand reads add
    static constexpr size_t LDS = (size_t)(IN_ELEMS + C::W_ELEMS) * 2 + PS::BYTES + (size_t)(XI_ELEMS + ONES_ELEMS) * 2;
    static_assert(LDS >= (size_t)(WLEN + 4 * 32) * 4, "the final reduction reuses the tile memory");
    static_assert(2 * LDS <= 160 * 1024, "two workgroups per CU");
    static_assert(PS::NW <= NT && PS::NTASK2 == 5 * 256, "one pooled-stage word per thread; 256 two-channel gather tasks per block row");
};
#ifdef BF_TIMING      // scratch/kbench_cb.hip: per-phase shader-clock totals of wave 0 (slots 0-7) and wave 4 (8-15) of every workgroup
__device__ unsigned long long g_bf_timing[16];
#define B2CK(k) do { if ((threadIdx.x & 255) == 0) { const long long now_ = clock64(); tacc_[k] += now_ - tlast_; tlast_ = now_; } } while (0)
#else
#define B2CK(k) do { } while (0)
#endif
__global__ __launch_bounds__(512, 4) void block2_conv_bwd_bf16_kernel(ConvArgs a) {
    using K = B2Bwd; using C = K::C; using PS = K::PS;
    extern __shared__ __attribute__((aligned(16))) unsigned short smem_h[];
    unsigned short* s_in = smem_h;
    unsigned short* s_w = s_in + K::IN_ELEMS;
    unsigned short* s_pd = s_w + C::W_ELEMS;
    unsigned short* s_xi = s_pd + PS::PD_ELEMS + (PS::PD_ELEMS + 1) / 2;
    unsigned short* s_ones = s_xi + K::XI_ELEMS;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 15, kq = lane >> 4;
    const int pair = wave & 3, role = wave >> 2;                             // weight gradient: role 0 taps 0-4, role 1 taps 5-8 + bias (wave-uniform)
#ifdef BF_TIMING
    long long tacc_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast_ = clock64();
#endif
    const unsigned short* g_dp = (const unsigned short*)a.in;
    const unsigned short* g_xi = (const unsigned short*)a.wg_in;
    unsigned short* g_out = (unsigned short*)a.out;

    for (int e = tid; e < C::COUT * C::WS / 8; e += K::NT) ((uint4*)s_w)[e] = ((const uint4*)a.wbank)[e];
    for (int e = tid; e < K::IN_ELEMS / 8; e += K::NT) ((uint4*)s_in)[e] = (uint4){0u, 0u, 0u, 0u};            // halo columns stay zero
    for (int e = tid; e < K::XI_ELEMS / 8; e += K::NT) ((uint4*)s_xi)[e] = (uint4){0u, 0u, 0u, 0u};
    for (int e = tid; e < K::ONES_ELEMS / 2; e += K::NT) ((unsigned*)s_ones)[e] = 0x3f803f80u;

    // ---- staging registers: 16-byte word e of the pooled stage / of the input tile is contiguous from the item's first row
    uint4 rd, rxi[K::KXI]; uint2 ra;
    // A workgroup takes whole images, item k = conv rows 8k .. 8k+7 (work = 4 * local image + k; image = blockIdx.x + local * gridDim.x).
    // Every load is unconditional from a row clamped into the image; rows outside it are replaced when the registers are stored
    // (a load under a condition made the compiler wait for it right where it was issued).
    auto item_loads = [&](int work) {                                         // pooled rows 4k .. 4k+5, input rows ty0-1 .. ty0+8
        const int img = blockIdx.x + (work >> 2) * gridDim.x, k = work & 3, ty0 = k * 8;
        const int oy = 4 * k + tid / 64, oyc = oy < 16 ? oy : 15;
        const long long po = (((long long)img * 16 + oyc) * 16 * 4 + tid % 64) * 8;
        rd = *(const uint4*)(g_dp + po); ra = *(const uint2*)(a.pool_arg + po);
#pragma unroll
        for (int q = 0; q < K::KXI; ++q) {
            const int e = tid + q * K::NT, gy = ty0 - 1 + e / 64, gyc = gy < 0 ? 0 : (gy > 31 ? 31 : gy);
            rxi[q] = *(const uint4*)(g_xi + (((long long)img * 32 + gyc) * 64 + e % 64) * 8);
        }
    };
    // ---- weight-gradient state: 10 accumulator tiles (slot s, channel block cb); role 0: slot = tap, role 1: slots 0-3 = taps 5-8, slot 4 = bias
    f32x4 wacc[10];
#pragma unroll
    for (int k = 0; k < 10; ++k) wacc[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int cp = lane & 3;
    // this lane's first source pixel of the pair's first step (tile coordinates); the second half is +8 columns, the second step +1 row
    const int pl0 = 64 * pair + 16 * (kq >> 1) + 4 * (kq & 1) + ((lane & 15) >> 2);
    const int prow0 = (pl0 / 32) * C::PW + pl0 % 32;
    const unsigned short* dv_p = s_in + (prow0 + C::PW + 1) * C::S + 4 * cp;      // d(conv output) at the pixel: s_in row 0 = ty0-1, col 0 = -1
    const unsigned short* xv_p = s_xi + prow0 * K::SX + 4 * cp;                   // forward input at pixel + tap - (1, 1): same origin
    // ---- data-gradient constants: two 16-pixel tiles per wave (tile = half a row: both tiles of a wave are in row `wave`)
    const unsigned short* av_p = s_in + (wave * C::PW + i) * C::S + kq * 8;
    const unsigned short* bv_p = s_w + i * C::WS + kq * 8;
    const int ooff = (wave * 32 + i) * 16 + kq * 4;
    // One code path for both roles (two paths cost a second copy of the accumulators): the role only changes where the five slots
    // read their forward-input operand; the bias slot of role 1 reads a block of 1.0s, so its "tap" sums d(conv output) itself.
    const unsigned short* xvp[5];
#pragma unroll
    for (int sl = 0; sl < 5; ++sl) {
        const int tap = role ? 5 + sl : sl;
        xvp[sl] = (role && sl == 4) ? s_ones : xv_p + ((tap / 3) * C::PW + tap % 3) * K::SX;
    }
    auto wg_steps = [&]() {
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            bf16x8 dv[2];
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) {
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(dv_p + st * C::PW * C::S + cb * 16));
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(dv_p + (st * C::PW + 8) * C::S + cb * 16));
                dv[cb] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
            }
#pragma unroll
            for (int sl = 0; sl < 5; ++sl) {
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(xvp[sl] + st * C::PW * K::SX));
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(xvp[sl] + (st * C::PW + 8) * K::SX));
                const bf16x8 xv = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
                for (int cb = 0; cb < 2; ++cb) wacc[sl * 2 + cb] = MFMA_BF16(dv[cb], xv, wacc[sl * 2 + cb]);
            }
        }
    };

    const int nwork = 4 * ((a.n - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x);       // grid <= n: at least one image
    item_loads(0);
    B2CK(7);                                                                 // prologue: LDS fills, constants, first loads issued
    uint2 ost[2] = {(uint2){0u, 0u}, (uint2){0u, 0u}}; unsigned short* ob = g_out;
    for (int work = 0; work < nwork; ++work) {
        const int img = blockIdx.x + (work >> 2) * gridDim.x, k = work & 3, ty0 = k * 8;
        __syncthreads();                                                     // the previous item's LDS reads are done
        __builtin_amdgcn_s_waitcnt(0x0F70);                                  // vmcnt(0): the staged loads, once, on every path (else the compiler waits behind the stores below)
        B2CK(0);
        // Rolling gradient rows: s_in row r = conv row ty0-1+r.  Rows 0-2 of this item are rows 8-10 of the previous one (the gather
        // always runs one block row ahead); the first item of an image has a zero row 0 (conv row -1) and gathers rows 1-2 itself.
        if (k) { for (int e = tid; e < 3 * K::ROW_WORDS; e += K::NT) ((uint4*)s_in)[e] = ((const uint4*)s_in)[8 * K::ROW_WORDS + e]; }
        else if (tid < K::ROW_WORDS) ((uint4*)s_in)[tid] = (uint4){0u, 0u, 0u, 0u};
        if (tid < PS::NW) {
            const bool in = 4 * k + tid / 64 < 16;                           // pooled rows past the image: zero gradient, arg-max that matches no position
            *(uint4*)(s_pd + tid * 8) = in ? rd : (uint4){0u, 0u, 0u, 0u};
            *(uint2*)((uint8_t*)(s_pd + PS::PD_ELEMS) + tid * 8) = in ? ra : (uint2){0xffffffffu, 0xffffffffu};
        }
#pragma unroll
        for (int q = 0; q < K::KXI; ++q) {
            const int e = tid + q * K::NT, gy = ty0 - 1 + e / 64;
            if (e < K::NXI) *(uint4*)(s_xi + ((e / 64) * C::PW + (e / 2) % 32 + 1) * K::SX + (e % 2) * 8) = (gy >= 0 && gy < 32) ? rxi[q] : (uint4){0u, 0u, 0u, 0u};
        }
        // the previous item's output leaves here, behind the wait for the staged loads: its stores have a whole item to complete
        // before the next such wait (vmcnt counts loads and stores in order)
        if (work) { *(uint2*)(ob + ooff) = ost[0]; *(uint2*)(ob + ooff + 256) = ost[1]; }
        __syncthreads();
        B2CK(1);
        // block rows 4k+1 .. 4k+4 (first item: 4k .. 4k+4) -> conv rows up to 8k+9, every one of them stored (s_in row = 2 * bl + dy + 1)
#pragma unroll 1
        for (int task = tid + (k ? 256 : 0); task < PS::NTASK2; task += K::NT) PS::gather_task2(s_pd, task, 4 * k, -1000, 1000, s_in, ty0 - 1, C::PW, 1, C::S);
        item_loads(work + 1 < nwork ? work + 1 : work);                     // unconditional (the last item is simply fetched again): no merge, no copies behind the loads
        B2CK(2);
        __syncthreads();
        B2CK(3);
        // ---- weight gradient: the pair's two pixel steps, this role's five slots
        wg_steps();
        B2CK(4);
        // ---- data gradient of the wave's two pixel tiles
        f32x4 acc[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int m = 0; m < 9; ++m) {
            const bf16x8 bv = *(const bf16x8*)(bv_p + m * 32);
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) acc[mt] = MFMA_BF16(bv, *(const bf16x8*)(av_p + ((m / 3) * C::PW + (m % 3) + 16 * mt) * C::S), acc[mt]);
        }
        B2CK(5);
        ob = g_out + ((long long)img * 32 + ty0) * 32 * 16;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) ost[mt] = (uint2){mi_pk_bf16(acc[mt][0], acc[mt][1]), mi_pk_bf16(acc[mt][2], acc[mt][3])};
    }
    if (nwork > 0) { *(uint2*)(ob + ooff) = ost[0]; *(uint2*)(ob + ooff + 256) = ost[1]; }
    // ---- waves summed through LDS in fixed order (role 0: waves 0-3 over taps 0-4; role 1: waves 4-7 over taps 5-8 and the bias sums)
    __syncthreads();
    float* red = (float*)smem_h;
    float* redb = red + K::WLEN;
    for (int w = 0; w < 4; ++w) {
        if (pair == w) {
#pragma unroll
            for (int sl = 0; sl < 5; ++sl)
#pragma unroll
                for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float v = wacc[sl * 2 + cb][r];
                        if (role && sl == 4) { if (i == 0) { const int o = cb * 16 + kq * 4 + r; redb[o] = (w == 0) ? v : redb[o] + v; } }
                        else {
                            const int tap = role ? 5 + sl : sl;
                            const int o = ((cb * 16 + kq * 4 + r) * 9 + tap) * 16 + i;
                            red[o] = (w == 0) ? v : red[o] + v;
                        }
                    }
        }
        __syncthreads();
    }
    float* slab = a.wg_partial + (long long)blockIdx.x * K::SLAB;
    for (int e = tid; e < K::SLAB; e += K::NT) slab[e] = red[e];           // the bias sums follow the weights in both layouts
#ifdef BF_TIMING
    B2CK(6);                                                                 // last output stores, wave reduction, slab
    if ((threadIdx.x & 255) == 0) for (int q = 0; q < 8; ++q) atomicAdd(&g_bf_timing[(threadIdx.x >> 8) * 8 + q], (unsigned long long)tacc_[q]);
#endif
}
static int b2bwd_grid(int n) { return n > 512 ? 512 : n; }      // whole images per workgroup, two workgroups per CU
static void launch_block2_conv_bwd(const ConvArgs& a, hipStream_t st) {
    static std::once_flag attr;          // (launchers run on up to 4 group worker threads)
    std::call_once(attr, [] { hipFuncSetAttribute((const void*)block2_conv_bwd_bf16_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)B2Bwd::LDS); });
    const int grid = b2bwd_grid(a.n);
    if (grid < 1) return;
    hipLaunchKernelGGL(block2_conv_bwd_bf16_kernel, dim3(grid), dim3(B2Bwd::NT), B2Bwd::LDS, st, a);
}

// ------------------------------------------------------------------------------------------ block3.conv: whole backward in one launch
// Data gradient AND weight / bias gradient of block3.conv (32 -> 32 channels @16x16) from the POOLED output gradient (8x8) + arg-max
// bytes.  Until round 3 two generic launches did this (weight gradient 80 us + data gradient 77 us per 8192 samples), each with its own
// max-pool-backward gather and one small image per item.  Here: two images per item, 512 threads in the two-role scheme of the residual
// blocks' whole-backward kernels -- all eight waves stage and gather (PoolStage::gather_task2, the arithmetic of the generic kernels: the
// conv-output gradient tile is bit-identical), then waves 0-3 run the transposed conv with the bank in registers (18 fragments, 8 pixel
// tiles per wave) while waves 4-7 run the weight gradient (16 pixel steps of 32, (tap, input block) columns dealt to the waves, bias
// through a tile of 1.0s).  dX is bit-identical to the generic data-gradient kernel (same bank, same tap order).
struct B3Bwd {
    using PS = PoolStage<32, 8, 9>;                                         // pooled rows 0..7 + one row of "no window" (arg 0xff)
    static constexpr int NT = 512, NIMG = 2, HW = 16, P = HW + 2, S = 48;  // S: pixel stride (bf16 elements) of the 32-channel tiles
    static constexpr int T_ELEMS = P * P * S;                               // a haloed 16x16x32 tile
    static constexpr int PD_ALL = PS::PD_ELEMS + (PS::PD_ELEMS + 1) / 2;    // pooled stage of one image in bf16 elements (gradient + arg bytes)
    static constexpr int WS = 9 * 32 + 16, WLEN = 32 * 9 * 32, SLAB = WLEN + 32;
    static constexpr int NMT = NIMG * HW * HW / 16, NSTEP = NIMG * HW * HW / 32;      // 32 pixel tiles, 16 pixel steps
    static constexpr size_t LDS = (size_t)NIMG * (2 * T_ELEMS + PD_ALL) * 2;
    static_assert(LDS <= 160 * 1024 && PD_ALL % 8 == 0, "one workgroup per CU; 16-byte aligned stages");
    static_assert(PS::NTASK2 == 1024, "two gather tasks per thread and image");
};
__global__ __launch_bounds__(512, 2) void block3_conv_bwd_bf16_kernel(ConvArgs a) {
    using K = B3Bwd; using PS = K::PS;
    extern __shared__ __attribute__((aligned(16))) unsigned short smem_h[];
    unsigned short* s_dc = smem_h;                                          // [NIMG] gathered conv-output gradient, haloed
    unsigned short* s_xi = s_dc + K::NIMG * K::T_ELEMS;                     // [NIMG] forward input, haloed
    unsigned short* s_pd = s_xi + K::NIMG * K::T_ELEMS;                     // [NIMG] pooled gradient (bf16) + arg-max bytes
    const int tid = threadIdx.x, lane = tid & 63, i = lane & 15, kq = lane >> 4, rq = (lane & 15) >> 2, cp = lane & 3;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool conv_role = wv < 4;
    const int rw = wv & 3;
    const unsigned short* g_dp = (const unsigned short*)a.in;
    const unsigned short* g_xi = (const unsigned short*)a.wg_in;
    unsigned short* g_out = (unsigned short*)a.out;
    for (int e = tid; e < (int)(K::LDS / 16); e += K::NT) ((uint4*)smem_h)[e] = (uint4){0u, 0u, 0u, 0u};      // tile borders stay zero
    __syncthreads();
    for (int e = tid; e < K::NIMG * 8 * 32; e += K::NT)                     // pooled row 8 does not exist: an arg-max that matches no position
        ((uint8_t*)(s_pd + (e >> 8) * K::PD_ALL + PS::PD_ELEMS))[8 * 8 * 32 + (e & 255)] = 0xff;
    // one register array for both roles: conv role st[2m + nb] = fragment (tap m, output block nb) of the transposed bank;
    // weight-gradient role st[2qq + cb] = accumulator tile of this wave's column qq and output block cb, st[20 + cb] = bias accumulators
    constexpr int NST = 22, QM = 5;
    f32x4 st[NST];
    const int qcnt = (18 - rw + 3) / 4;                                     // columns q = rw + 4 qq of the 18 (tap, input block) columns: 5, 5, 4, 4
    if (conv_role) {
#pragma unroll
        for (int m = 0; m < 9; ++m)
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) st[2 * m + nb] = __builtin_bit_cast(f32x4, *(const uint4*)(a.wbank + (nb * 16 + i) * K::WS + m * 32 + kq * 8));
#pragma unroll
        for (int q = 18; q < NST; ++q) st[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
    } else {
#pragma unroll
        for (int q = 0; q < NST; ++q) st[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);                                     // vmcnt(0): the fragments are in
    const bf16x8 ones = __builtin_bit_cast(bf16x8, (uint4){0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u});
    auto koffc = [](int m) { return ((m / 3) * K::P + (m % 3)) * K::S; };
    // pixel pl (0 .. 511) of the item: image pl / 256, row (pl / 16) % 16, column pl % 16 -> element offset of its window origin in a tile pair
    auto porg = [](int pl) { return (pl >> 8) * K::T_ELEMS + (((pl >> 4) & 15) * K::P + (pl & 15)) * K::S; };
    constexpr int CENTER = (K::P + 1) * K::S;

    const int nwork = (a.n + K::NIMG - 1) / K::NIMG;
    typedef unsigned b3_u32x4 __attribute__((ext_vector_type(4)));            // (ext-vector typed: an array of the HIP uint4 STRUCT copied global -> local -> LDS goes through scratch memory)
    typedef unsigned b3_u32x2 __attribute__((ext_vector_type(2)));
    b3_u32x4 rx[4], rg; b3_u32x2 ra;                                        // staging registers of both roles: 4 input words, one pooled-gradient word + its arg bytes
    auto load = [&](int work) {
        const int img0 = work * K::NIMG;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int e = tid + k * K::NT, pl = e >> 2;
            const int n = img0 + (pl >> 8) < a.n ? img0 + (pl >> 8) : a.n - 1;       // an image past the end: a valid one, zeroed at the store
            rx[k] = *(const b3_u32x4*)(g_xi + ((long long)n * 256 + (pl & 255)) * 32 + (e & 3) * 8);
        }
        const int n = img0 + (tid >> 8) < a.n ? img0 + (tid >> 8) : a.n - 1;
        const long long o = (long long)n * 2048 + (tid & 255) * 8;
        rg = *(const b3_u32x4*)(g_dp + o); ra = *(const b3_u32x2*)(a.pool_arg + o);
    };
    auto wg_step = [&](int t) {
        int orow[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) orow[h] = porg(32 * t + 16 * (kq >> 1) + 8 * h + 4 * (kq & 1) + rq) + 4 * cp;
        auto tr = [&](const unsigned short* base, int off) {
            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(base + orow[0] + off));
            const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(base + orow[1] + off));
            return (bf16x8)__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        };
        bf16x8 d[2], b[QM];
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) d[cb] = tr(s_dc, CENTER + cb * 16);
#pragma unroll
        for (int qq = 0; qq < QM; ++qq) {
            const int q = rw + 4 * (qq < 4 ? qq : (qcnt > 4 ? 4 : 0)), tap = q >> 1, ib = q & 1;
            b[qq] = tr(s_xi, koffc(tap) + ib * 16);
        }
        asm volatile("" ::: "memory");
        if (rw == 2) { st[20] = MFMA_BF16(d[0], ones, st[20]); st[21] = MFMA_BF16(d[1], ones, st[21]); }
#pragma unroll
        for (int qq = 0; qq < QM; ++qq)
            if (qq < qcnt) {
#pragma unroll
                for (int cb = 0; cb < 2; ++cb) st[2 * qq + cb] = MFMA_BF16(d[cb], b[qq], st[2 * qq + cb]);
            }
    };

    if ((int)blockIdx.x < nwork) load(blockIdx.x);
    for (int work = blockIdx.x; work < nwork; work += gridDim.x) {
        const int img0 = work * K::NIMG, left = a.n - img0;                  // images of this item that exist
        __syncthreads();                                                     // the previous item's readers are done
        {
            const b3_u32x4 z = {0u, 0u, 0u, 0u};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int e = tid + k * K::NT, pl = e >> 2;
                *(b3_u32x4*)(s_xi + porg(pl) + CENTER + (e & 3) * 8) = (pl >> 8) < left ? rx[k] : z;
            }
            unsigned short* pd = s_pd + (tid >> 8) * K::PD_ALL;
            *(b3_u32x4*)(pd + (tid & 255) * 8) = (tid >> 8) < left ? rg : z;
            *(b3_u32x2*)((uint8_t*)(pd + PS::PD_ELEMS) + (tid & 255) * 8) = ra;
        }
        __syncthreads();
        // max-pool backward of both images into the haloed tiles (conv rows 0..15 -> tile rows 1..16), two tasks per thread and image
#pragma unroll
        for (int im = 0; im < K::NIMG; ++im)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                PS::gather_task2(s_pd + im * K::PD_ALL, tid + j * K::NT, 0, -1000, 1000, s_dc + im * K::T_ELEMS, -1, K::P, 1, K::S);
        if (work + (int)gridDim.x < nwork) load(work + gridDim.x);
        __syncthreads();
        if (conv_role) {
            // ---- dX = convT(dC): tiles rw, rw + 4, ... of the 32, one at a time (weights in registers)
            for (int t = rw; t < K::NMT; t += 4) {
                const int pl = t * 16 + i;
                const unsigned short* src = s_dc + porg(pl) + kq * 8;
                bf16x8 av[5];
#pragma unroll
                for (int m = 0; m < 5; ++m) av[m] = *(const bf16x8*)(src + koffc(m));
                asm volatile("" ::: "memory");
                f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
                for (int m = 0; m < 9; ++m) {
                    const bf16x8 cur = av[m % 5];
                    acc[0] = MFMA_BF16(__builtin_bit_cast(bf16x8, st[2 * m]), cur, acc[0]);
                    acc[1] = MFMA_BF16(__builtin_bit_cast(bf16x8, st[2 * m + 1]), cur, acc[1]);
                    if (m < 4) av[m] = *(const bf16x8*)(src + koffc(m + 5));
                }
                if ((pl >> 8) < left) {
#pragma unroll
                    for (int nb = 0; nb < 2; ++nb)
                        *(uint2*)(g_out + ((long long)img0 * 256 + pl) * 32 + nb * 16 + kq * 4) =
                            (uint2){mi_pk_bf16(acc[nb][0], acc[nb][1]), mi_pk_bf16(acc[nb][2], acc[nb][3])};
                }
            }
        } else {
            // ---- weight / bias gradient from (dC, forward input)
#pragma unroll 4
            for (int t = 0; t < K::NSTEP; ++t) wg_step(t);
        }
    }
    if (!conv_role) {                                                        // every weight-gradient wave owns its columns: straight to the slab
        float* sl = a.wg_partial + (long long)blockIdx.x * K::SLAB;
#pragma unroll
        for (int qq = 0; qq < QM; ++qq)
            if (qq < qcnt) {
                const int q = rw + 4 * qq, tap = q >> 1, ib = q & 1;
#pragma unroll
                for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) sl[((cb * 16 + kq * 4 + r) * 9 + tap) * 32 + ib * 16 + i] = st[2 * qq + cb][r];
            }
        if (i == 0 && rw == 2) {
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                for (int r = 0; r < 4; ++r) sl[K::WLEN + cb * 16 + kq * 4 + r] = st[20 + cb][r];
        }
    }
}
#ifndef B3BWD_FUSED
#define B3BWD_FUSED 1              // 0: block3.conv's two gradients stay two generic launches
#endif
static int b3bwd_grid(int n) { const int w = (n + B3Bwd::NIMG - 1) / B3Bwd::NIMG; return w > 256 ? 256 : w; }
static void launch_block3_conv_bwd(const ConvArgs& a, hipStream_t st) {
    static std::once_flag attr;
    std::call_once(attr, [] { hipFuncSetAttribute((const void*)block3_conv_bwd_bf16_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)B3Bwd::LDS); });
    const int grid = b3bwd_grid(a.n);
    if (grid < 1) return;
    hipLaunchKernelGGL(block3_conv_bwd_bf16_kernel, dim3(grid), dim3(B3Bwd::NT), B3Bwd::LDS, st, a);
}

// ------------------------------------------------------------------------------------------ filter-bank packing
int bank_ws(int cin_pass) { return (cin_pass == 32 ? 9 : 5) * 32 + 16; }
__global__ void pack_banks_kernel(const float* __restrict__ params, unsigned short* __restrict__ banks, const BankDesc* __restrict__ desc) {
    const BankDesc d = desc[blockIdx.y];
    const float* w = params + d.w_off;                    // forward layout [co_f][9][ci_f]
    unsigned short* out = banks + d.out_off;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < d.rows * d.ws; e += gridDim.x * blockDim.x) {
        const int j = e / d.ws, k = e % d.ws, tap = k / d.cin_pass, ci = k % d.cin_pass;
        float v = 0.f;
        if (k < d.nk * 32 && tap < 9) v = d.transw ? w[(ci * 9 + (8 - tap)) * d.ci_f + j] : w[(j * 9 + tap) * d.ci_f + ci];
        out[e] = f2bf(v);
    }
}
void launch_pack_banks(const float* params, unsigned short* banks, const BankDesc* d_desc, int n_desc, hipStream_t st) {
    if (n_desc <= 0) return;
    hipLaunchKernelGGL(pack_banks_kernel, dim3(8, n_desc), dim3(256), 0, st, params, banks, d_desc);
}

// ------------------------------------------------------------------------------------------ weight gradient (bf16)
// dW[co][tap][ci] = sum over pixels dOut[p][co] * relu?(in[p + tap][ci]) on v_mfma_f32_16x16x32_bf16:
// M = 16 output channels, N = 16 input channels, K = 32 pixels per instruction.  Both operands are K(=pixel)-major
// while the staged tiles are [pixel][channel]: the fragments come through ds_read_b64_tr_b16 (the LDS transpose
// read), two per operand per MFMA.  Accumulators (one 16x16 tile per tap / co-block / ci-block) stay in registers
// across the persistent loop; waves are summed through LDS in fixed order; one slab per workgroup.
// tuning knobs of the 32 -> 32 weight-gradient kernels (scratch/kbench.hip sweeps them): min waves per SIMD the
// compiler must leave room for, tile height at 16x16, images per item at 8x8
#ifndef WG_WPE
#define WG_WPE 2
#endif
#ifndef WG16_TH          // tile height of the 16x16 / images per item of the 8x8 weight-gradient kernels (scratch/kbench.hip sweeps them)
#define WG16_TH 16
#endif
#ifndef WG8_NIMG
#define WG8_NIMG 4
#endif

template <int CIN_, int COUT_, int HW_, int TH_, int TW_, int NIMG_>
struct WbCfg {
    static constexpr int CIN = CIN_, COUT = COUT_, HW = HW_, TH = TH_, TW = TW_, NIMG = NIMG_;
    // bf16 elements per staged pixel (input tile / dOut tile): with the pixel order below the 8 rows a half-wave's
    // ds_read_b64_tr_b16 touches are 8 consecutive pixels, conflict-free when the row stride is 32 B x odd
    static constexpr int S = (CIN == 16) ? 16 : 48, SO = (COUT == 16) ? 16 : 48;
    static constexpr int PH = TH + 2, PW = TW + 2, NPIX = NIMG * PH * PW, NT = NIMG * TH * TW;
    static constexpr int IN_ELEMS = ((NPIX * S + 7) / 8) * 8, DO_ELEMS = ((NT * SO + 7) / 8) * 8;
    static constexpr int NCB = COUT / 16, NIB = CIN / 16, NSTEP = NT / 32;
    static constexpr int WLEN = COUT * 9 * CIN, SLAB = WLEN + COUT;
    static constexpr int C8 = CIN / 8, OC8 = COUT / 8;
    static constexpr int NLD = (NPIX * C8 + 255) / 256, NDO = (NT * OC8 + 255) / 256;
    static constexpr int TPI_X = HW / TW, TPI = (HW / TH) * (HW / TW);
    // SPLIT (32 -> 32 channels): the 18 (tap, input block) column tiles are dealt round-robin to the 4 waves, every wave
    // walks ALL pixel steps with 10 accumulator tiles instead of 36 (312 -> ~110 registers: 1 -> 2+ waves per SIMD) and
    // owns its outputs outright, so the cross-wave reduction through LDS disappears as well.
    static constexpr bool SPLIT = (CIN == 32 && COUT == 32);
    static constexpr int NQ = 9 * NIB, QMAX = (NQ + 3) / 4;
    static constexpr size_t TILE_BYTES = (size_t)(IN_ELEMS + DO_ELEMS) * 2, RED_BYTES = (size_t)((SPLIT ? 0 : WLEN) + 256) * 4;
    static constexpr size_t LDS_BYTES = TILE_BYTES > RED_BYTES ? TILE_BYTES : RED_BYTES;
    static_assert(NT % 32 == 0, "pixel steps of 32");
};

#ifdef WG_TIMING      // scratch/kbench.hip: per-phase shader-clock totals of wave 0, summed over workgroups
__device__ unsigned long long g_wg_timing[8];
#define TCK(k) do { if (tid == 0) { const long long now_ = clock64(); tacc_[k] += now_ - tlast_; tlast_ = now_; } } while (0)
#else
#define TCK(k) do { } while (0)
#endif

// POOLED (a block's first conv): a.dout is the POOLED gradient, a.pool_arg the arg-max bytes; the dOut tile is rebuilt in
// LDS by PoolStage::gather (max-pool backward fused into the operand staging).
template <class C, bool POOLED = false>
__global__ __launch_bounds__(256, C::SPLIT ? WG_WPE : 1) void conv3x3_wgrad_bf16_kernel(WgradArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned short smem_h[];
    using PS = PoolStage<C::COUT, C::HW / 2, C::TH / 2 + 1>;              // pooled rows ty0/2 .. ty0/2 + TH/2
    static_assert(!POOLED || (C::NIMG == 1 && C::TW == C::HW && C::TH % 2 == 0), "pooled dOut: full-width row tiles of one image");
    PS ps;
#ifdef WG_TIMING
    long long tacc_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast_ = clock64();
#endif
    unsigned short* s_in = smem_h;
    unsigned short* s_do = smem_h + C::IN_ELEMS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 15, kq = lane >> 4, rq = (lane & 15) >> 2, cp = lane & 3;
    const unsigned short* g_in = (const unsigned short*)a.in;
    const unsigned short* g_do = (const unsigned short*)a.dout;
    constexpr int NACC = C::SPLIT ? C::QMAX * C::NCB : 9 * C::NCB * C::NIB;
    f32x4 acc[NACC];
#pragma unroll
    for (int k = 0; k < NACC; ++k) acc[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int wv = __builtin_amdgcn_readfirstlane(wave);          // wave-uniform copy (SGPR): SPLIT column ownership
    const int qcnt = (C::NQ - wv + 3) / 4;                         // columns q = wv + 4*qq, qq < qcnt
    // bias gradient = row sums of the dOut operand: one extra MFMA against an all-ones B tile (exact: x * 1.0, fp32 adds)
    f32x4 accb[C::NCB];
#pragma unroll
    for (int cb = 0; cb < C::NCB; ++cb) accb[cb] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const bf16x8 ones = __builtin_bit_cast(bf16x8, (uint4){0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u});

    const int nwork = (C::NIMG > 1) ? (a.n + C::NIMG - 1) / C::NIMG : a.n * C::TPI;
    uint4 rin[C::NLD], rdo[C::NDO];
    // whole-image tiles: a thread's source offsets inside the item never change -> computed once (-1 = halo outside the image)
    int ioff[C::TPI == 1 ? C::NLD : 1], doff[C::TPI == 1 ? C::NDO : 1];
    if constexpr (C::TPI == 1) {
#pragma unroll
        for (int k = 0; k < C::NLD; ++k) {
            const int e = tid + k * 256;
            ioff[k] = -1;
            if (e < C::NPIX * C::C8) {
                const int pix = e / C::C8, c8 = e % C::C8, img = pix / (C::PH * C::PW), q = pix % (C::PH * C::PW);
                const int gy = q / C::PW - 1, gx = q % C::PW - 1;
                if (gy >= 0 && gy < C::HW && gx >= 0 && gx < C::HW) ioff[k] = ((img * C::HW + gy) * C::HW + gx) * C::CIN + c8 * 8;
            }
        }
#pragma unroll
        for (int k = 0; k < C::NDO; ++k) {
            const int e = tid + k * 256;
            doff[k] = -1;
            if (e < C::NT * C::OC8) {
                const int pl = e / C::OC8, c8 = e % C::OC8, y = pl / C::TW, x = pl % C::TW;
                doff[k] = (((y / C::TH) * C::HW + (y % C::TH)) * C::HW + x) * C::COUT + c8 * 8;
            }
        }
    }
    auto coords = [&](int work, int& img0, int& ty0, int& tx0) {
        if (C::NIMG > 1) { img0 = work * C::NIMG; ty0 = 0; tx0 = 0; }
        else { img0 = work / C::TPI; const int t = work % C::TPI; ty0 = (t / C::TPI_X) * C::TH; tx0 = (t % C::TPI_X) * C::TW; }
    };
    auto load = [&](int img0, int ty0, int tx0) {
        if constexpr (C::TPI == 1) {
            const int left = a.n - img0;                                      // images of this item that exist
            const unsigned short* bi = g_in + (long long)img0 * (C::HW * C::HW * C::CIN);
            const unsigned short* bd = g_do + (long long)img0 * (C::HW * C::HW * C::COUT);
#pragma unroll
            for (int k = 0; k < C::NLD; ++k) {
                uint4 v = {0u, 0u, 0u, 0u};
                if (ioff[k] >= 0 && (C::NIMG == 1 || ioff[k] / (C::HW * C::HW * C::CIN) < left)) v = *(const uint4*)(bi + ioff[k]);
                rin[k] = v;
            }
            if constexpr (POOLED) ps.load(g_do, a.pool_arg, img0, ty0 / 2);
            else {
#pragma unroll
                for (int k = 0; k < C::NDO; ++k) {
                    uint4 v = {0u, 0u, 0u, 0u};
                    if (doff[k] >= 0 && (C::NIMG == 1 || doff[k] / (C::HW * C::HW * C::COUT) < left)) v = *(const uint4*)(bd + doff[k]);
                    rdo[k] = v;
                }
            }
            return;
        }
#pragma unroll
        for (int k = 0; k < C::NLD; ++k) {
            const int e = tid + k * 256;
            uint4 v = {0u, 0u, 0u, 0u};
            if (e < C::NPIX * C::C8) {
                const int pix = e / C::C8, c8 = e % C::C8, img = pix / (C::PH * C::PW), q = pix % (C::PH * C::PW);
                const int gy = ty0 + q / C::PW - 1, gx = tx0 + q % C::PW - 1, n = img0 + img;
                if (n < a.n && gy >= 0 && gy < C::HW && gx >= 0 && gx < C::HW)
                    v = *(const uint4*)(g_in + (((long long)n * C::HW + gy) * C::HW + gx) * C::CIN + c8 * 8);
            }
            rin[k] = v;
        }
        if constexpr (POOLED) { ps.load(g_do, a.pool_arg, img0, ty0 / 2); return; }
#pragma unroll
        for (int k = 0; k < C::NDO; ++k) {
            const int e = tid + k * 256;
            uint4 v = {0u, 0u, 0u, 0u};
            if (e < C::NT * C::OC8) {
                const int pl = e / C::OC8, c8 = e % C::OC8, y = pl / C::TW, x = pl % C::TW, n = img0 + y / C::TH;
                if (n < a.n)
                    v = *(const uint4*)(g_do + (((long long)n * C::HW + ty0 + (y % C::TH)) * C::HW + tx0 + x) * C::COUT + c8 * 8);
            }
            rdo[k] = v;
        }
    };
    int img0, ty0, tx0;
    if ((int)blockIdx.x < nwork) { coords(blockIdx.x, img0, ty0, tx0); load(img0, ty0, tx0); }
    TCK(0);
    for (int work = blockIdx.x; work < nwork; work += gridDim.x) {
        __syncthreads();
        TCK(1);
        int ty_cur = 0;
        if constexpr (POOLED) { int i_, x_; coords(work, i_, ty_cur, x_); }
#pragma unroll
        for (int k = 0; k < C::NLD; ++k) {
            const int e = tid + k * 256;
            if (e < C::NPIX * C::C8) {
                uint4 v = rin[k];
                if (a.relu_in) { v.x = relu_bf16x2(v.x); v.y = relu_bf16x2(v.y); v.z = relu_bf16x2(v.z); v.w = relu_bf16x2(v.w); }
                *(uint4*)(s_in + (e / C::C8) * C::S + (e % C::C8) * 8) = v;
            }
        }
        if constexpr (POOLED) {
            unsigned short* s_pd = s_do + C::DO_ELEMS;
            ps.store(s_pd);
            __syncthreads();
            PS::gather(s_pd, ty_cur / 2, ty_cur, ty_cur + C::TH, s_do, ty_cur, C::TW, 0, C::SO);
        } else {
#pragma unroll
            for (int k = 0; k < C::NDO; ++k) {
                const int e = tid + k * 256;
                if (e < C::NT * C::OC8) *(uint4*)(s_do + (e / C::OC8) * C::SO + (e % C::OC8) * 8) = rdo[k];
            }
        }
        TCK(2);
        __syncthreads();
        TCK(3);
        if (work + (int)gridDim.x < nwork) { coords(work + gridDim.x, img0, ty0, tx0); load(img0, ty0, tx0); }
        TCK(4);

        if constexpr (C::SPLIT) {
#pragma unroll
            for (int t = 0; t < C::NSTEP; ++t) {
                int drow[2], irow[2];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int pl = 32 * t + 16 * (kq >> 1) + 8 * h + 4 * (kq & 1) + rq, y = pl / C::TW, x = pl % C::TW;
                    drow[h] = pl * C::SO + 4 * cp;
                    irow[h] = (((y / C::TH) * C::PH + (y % C::TH)) * C::PW + x) * C::S + 4 * cp;
                }
                bf16x8 av[C::NCB];
#pragma unroll
                for (int cb = 0; cb < C::NCB; ++cb) {
                    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(s_do + drow[0] + cb * 16));
                    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(s_do + drow[1] + cb * 16));
                    av[cb] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                }
#pragma unroll
                for (int cb = 0; cb < C::NCB; ++cb)
                    if (wv == 2 + cb) accb[cb] = MFMA_BF16(av[cb], ones, accb[cb]);       // waves 2, 3 own one column fewer
#pragma unroll
                for (int qq = 0; qq < C::QMAX; ++qq) {
                    if (qq < qcnt) {                                  // wave-uniform: EXEC stays full for the transpose reads
                        const int q = wv + 4 * qq, tap = q / C::NIB, ib = q % C::NIB;
                        const int toff = ((tap / 3) * C::PW + (tap % 3)) * C::S + ib * 16;
                        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(s_in + irow[0] + toff));
                        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(s_in + irow[1] + toff));
                        const bf16x8 bv = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
                        for (int cb = 0; cb < C::NCB; ++cb) acc[qq * C::NCB + cb] = MFMA_BF16(av[cb], bv, acc[qq * C::NCB + cb]);
                    }
                }
            }
        } else
        for (int t = wave; t < C::NSTEP; t += 4) {
            // this lane's two source rows (pixels) of the 4x16 transpose blocks.  MFMA k index = 8*kq + 4*h + rq; the
            // pixel it stands for is a free (A/B-consistent) permutation: 16*(kq>>1) + 8*h + 4*(kq&1) + rq
            int drow[2], irow[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int pl = 32 * t + 16 * (kq >> 1) + 8 * h + 4 * (kq & 1) + rq, y = pl / C::TW, x = pl % C::TW;
                drow[h] = pl * C::SO + 4 * cp;
                irow[h] = (((y / C::TH) * C::PH + (y % C::TH)) * C::PW + x) * C::S + 4 * cp;
            }
            bf16x8 av[C::NCB];
#pragma unroll
            for (int cb = 0; cb < C::NCB; ++cb) {
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(s_do + drow[0] + cb * 16));
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(s_do + drow[1] + cb * 16));
                av[cb] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                accb[cb] = MFMA_BF16(av[cb], ones, accb[cb]);
            }
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int toff = ((tap / 3) * C::PW + (tap % 3)) * C::S;
#pragma unroll
                for (int ib = 0; ib < C::NIB; ++ib) {
                    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(s_in + irow[0] + toff + ib * 16));
                    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(s_in + irow[1] + toff + ib * 16));
                    const bf16x8 bv = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
                    for (int cb = 0; cb < C::NCB; ++cb)
                        acc[(tap * C::NCB + cb) * C::NIB + ib] = MFMA_BF16(av[cb], bv, acc[(tap * C::NCB + cb) * C::NIB + ib]);
                }
            }
        }
        TCK(5);
    }

    __syncthreads();
    if constexpr (C::SPLIT) {
        float* slab = a.partial + (long long)blockIdx.x * C::SLAB;
#pragma unroll
        for (int qq = 0; qq < C::QMAX; ++qq)
            if (qq < qcnt) {
                const int q = wv + 4 * qq, tap = q / C::NIB, ib = q % C::NIB;
#pragma unroll
                for (int cb = 0; cb < C::NCB; ++cb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) slab[((cb * 16 + kq * 4 + r) * 9 + tap) * C::CIN + ib * 16 + i] = acc[qq * C::NCB + cb][r];
            }
#pragma unroll
        for (int cb = 0; cb < C::NCB; ++cb)
            if (wv == 2 + cb && i == 0) {
#pragma unroll
                for (int r = 0; r < 4; ++r) slab[C::WLEN + cb * 16 + kq * 4 + r] = accb[cb][r];
            }
        TCK(7);
#ifdef WG_TIMING
        if (tid == 0) for (int k = 0; k < 8; ++k) atomicAdd(&g_wg_timing[k], (unsigned long long)tacc_[k]);
#endif
        return;
    }
    float* red = (float*)smem_h;
    float* redb = red + C::WLEN;
    for (int w = 0; w < 4; ++w) {
        if (wave == w) {
#pragma unroll
            for (int tap = 0; tap < 9; ++tap)
#pragma unroll
                for (int cb = 0; cb < C::NCB; ++cb)
#pragma unroll
                    for (int ib = 0; ib < C::NIB; ++ib)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int o = ((cb * 16 + kq * 4 + r) * 9 + tap) * C::CIN + ib * 16 + i;
                            const float v = acc[(tap * C::NCB + cb) * C::NIB + ib][r];
                            red[o] = (w == 0) ? v : red[o] + v;
                        }
        }
        __syncthreads();
    }
    if (i == 0) {
#pragma unroll
        for (int cb = 0; cb < C::NCB; ++cb)
#pragma unroll
            for (int r = 0; r < 4; ++r) redb[wave * C::COUT + cb * 16 + kq * 4 + r] = accb[cb][r];
    }
    __syncthreads();
    float* slab = a.partial + (long long)blockIdx.x * C::SLAB;
    for (int e = tid; e < C::WLEN; e += 256) slab[e] = red[e];
    if (tid < C::COUT) slab[C::WLEN + tid] = (redb[tid] + redb[C::COUT + tid]) + (redb[2 * C::COUT + tid] + redb[3 * C::COUT + tid]);
}

//                          CIN COUT HW  TH  TW NIMG
using WT_16_16_32 = WbCfg<16, 16, 32,  8, 32, 1>;
using WT_16_32_32 = WbCfg<16, 32, 32,  8, 32, 1>;
using WT_32_32_16 = WbCfg<32, 32, 16, WG16_TH, 16, 1>;
using WT_32_32_8  = WbCfg<32, 32,  8,  8,  8, WG8_NIMG>;

template <class C>
static int wb_grid(int n) {
    int bpc = (int)((160 * 1024) / C::LDS_BYTES);
    bpc = bpc < 1 ? 1 : (bpc > 4 ? 4 : bpc);
    const int w = (C::NIMG > 1) ? (n + C::NIMG - 1) / C::NIMG : n * C::TPI;
    return w > 256 * bpc ? 256 * bpc : w;
}
template <class C, bool POOLED = false>
static void launch_wb_t(const WgradArgs& a, hipStream_t st) {
    static std::once_flag attr;
    constexpr size_t TILE = C::TILE_BYTES + (POOLED ? PoolStage<C::COUT, C::HW / 2, C::TH / 2 + 1>::BYTES : 0);
    constexpr size_t LDS = TILE > C::RED_BYTES ? TILE : C::RED_BYTES;
    std::call_once(attr, [] { hipFuncSetAttribute((const void*)conv3x3_wgrad_bf16_kernel<C, POOLED>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS); });
    const int grid = wb_grid<C>(a.n);
    if (grid < 1) return;
    hipLaunchKernelGGL((conv3x3_wgrad_bf16_kernel<C, POOLED>), dim3(grid), dim3(256), LDS, st, a);
}
int wgrad_grid_bf16(ConvShape s, int n) {
    switch (s) {
        case CS_16_16_32: return wb_grid<WT_16_16_32>(n);
        case CS_16_32_32: return wb_grid<WT_16_32_32>(n);
        case CS_32_32_16: return wb_grid<WT_32_32_16>(n);
        case CS_32_32_8:  return wb_grid<WT_32_32_8>(n);
        case CS_3_16_64:  return c1_grid(n);
        default: return -1;
    }
}
void launch_conv_wgrad_bf16(ConvShape s, const WgradArgs& a, hipStream_t st) {
    switch (s) {
        case CS_16_16_32: launch_wb_t<WT_16_16_32>(a, st); break;
        case CS_16_32_32: if (a.pool_arg) launch_wb_t<WT_16_32_32, true>(a, st); else launch_wb_t<WT_16_32_32>(a, st); break;
        case CS_32_32_16: if (a.pool_arg) launch_wb_t<WT_32_32_16, true>(a, st); else launch_wb_t<WT_32_32_16>(a, st); break;
        case CS_32_32_8:  launch_wb_t<WT_32_32_8>(a, st); break;
        default: break;
    }
}

// ------------------------------------------------------------------------------------------ block1.conv (3 -> 16 @ 64x64, uint8 frames in)
// Forward: the haloed frame tile is staged as bf16 [pixel][r,g,b,0] (8 bytes per pixel, uint8 -> bf16 through a
// 256-entry table); K = 9 taps x 4 = 36 -> two K=32 MFMAs: the first covers taps 0..7 (lane quarter kq owns taps
// 2kq, 2kq+1: two 8-byte LDS reads), the second tap 8 (+ zero columns).  Weight gradient: M = 16 output channels,
// N = (tap, ci) = 27 -> 32 columns, K = pixels; the B operand needs the 27 values of a pixel contiguous, so an
// im2col image [pixel][32] is built in LDS from the staged tile (9 8-byte reads + 4 16-byte writes per pixel) and
// both operands come through ds_read_b64_tr_b16.
template <int TH_>
struct C1T {
    static constexpr int HW = 64, TH = TH_, TW = 64, PH = TH + 2, PW = 66, NPIX = PH * PW, NT = TH * TW;
    static constexpr int NTASK = PH * 16, NLD = 3;      // a task = 12 frame bytes = 4 pixels (3 dwords in, 4 x 8 bytes out)
    static constexpr int TPI = HW / TH;
    static_assert(NTASK <= 256, "one staging task per thread");
};
using C1 = C1T<8>;          // forward tile
using C1W = C1T<4>;         // weight-gradient tile (smaller LDS footprint -> 4 workgroups per CU)
// The frame index of a minibatch sample is a load of its own (idx[img]) in front of the pixel loads: it is resolved one item
// earlier than the pixels (c1_frame for item k+2 is issued with the pixel loads of item k+1), and the pixel loads are unconditional
// from a row clamped into the image -- c1_store zeroes the rows outside it.  (A conditional load made the compiler wait for it where
// it was issued, and the dependent pair exposed a full memory round trip per item.)
__device__ __forceinline__ long long c1_frame(const int32_t* idx, long long in_base, int img) { return idx ? (long long)idx[img] : in_base + img; }
// The early fetch goes through the constant address space with a uniform address, i.e. a scalar load (lgkmcnt): kept out of the
// in-order vmcnt stream of the pixel loads, where -- being the youngest load and needed first -- it forced the older pixel loads to
// be drained.  With idx == null a word of the frames is fetched and ignored (one path: no merge of load histories).
typedef const __attribute__((address_space(4))) int32_t* c1_const_i32p;
__device__ __forceinline__ int c1_frame_fetch(const int32_t* idx, const void* any_valid, int img) {
    const int32_t* p = idx ? idx + img : (const int32_t*)any_valid;
    return *(c1_const_i32p)p;
}
__device__ __forceinline__ long long c1_frame_of(const int32_t* idx, long long in_base, int fetched, int img) {
    return idx ? (long long)fetched : in_base + img;
}
template <class C1>
__device__ __forceinline__ void c1_load(uint32_t (&r)[C1::NLD], const uint8_t* frames, long long frame, int ty0) {
    // every thread loads (no branch: the compiler's in-order vmcnt bookkeeping stays exact); threads past the tile repeat its last row
    const int e = threadIdx.x, row = (e >> 4) < C1::PH ? (e >> 4) : C1::PH - 1, g = e & 15, gy = ty0 + row - 1;
    const int gyc = gy < 0 ? 0 : (gy > C1::HW - 1 ? C1::HW - 1 : gy);
    const uint32_t* p = (const uint32_t*)(frames + frame * (C1::HW * C1::HW * 3) + gyc * (C1::HW * 3) + g * 12);
    r[0] = p[0]; r[1] = p[1]; r[2] = p[2];
}
// uint8 -> bf16(k/255): k * (1/255) rounded to bf16 equals the bf16 of the exact quotient for all 256 values (checked
// exhaustively on the host, tests/test_host_logic.py), so no table: 12 converts + 4 8-byte LDS stores per task.
template <class C1, bool ONE4 = false>      // ONE4: the padding channel holds 1.0 (weight gradient: its centre-tap column sums the bias gradient)
__device__ __forceinline__ void c1_store(const uint32_t (&r)[C1::NLD], unsigned short* s_in, int ty0) {
    const int e = threadIdx.x, row = e >> 4, g = e & 15, gy = ty0 + row - 1;
    if (e < C1::NTASK) {
        const bool in = gy >= 0 && gy < C1::HW;
        const uint32_t px[3] = {in ? r[0] : 0u, in ? r[1] : 0u, in ? r[2] : 0u};      // rows outside the image: zeros (3 selects instead of 12)
        mi_f32x2 f[6];                                                               // 6 packed multiplies instead of 12
#pragma unroll
        for (int b = 0; b < 6; ++b)
            f[b] = (mi_f32x2){(float)((px[(2 * b) >> 2] >> (8 * ((2 * b) & 3))) & 0xffu), (float)((px[(2 * b + 1) >> 2] >> (8 * ((2 * b + 1) & 3))) & 0xffu)}
                   * (mi_f32x2){1.0f / 255.0f, 1.0f / 255.0f};
        auto fv = [&](int b) { return (b & 1) ? f[b >> 1].y : f[b >> 1].x; };
#pragma unroll
        for (int j = 0; j < 4; ++j)
            *(uint2*)(s_in + (row * C1::PW + 1 + g * 4 + j) * 4) =
                (uint2){mi_pk_bf16(fv(3 * j), fv(3 * j + 1)), mi_pk_bf16(fv(3 * j + 2), ONE4 ? 1.f : 0.f)};
    }
}

constexpr int C1_WS = 80;          // bf16 elements per output channel of the 64-column filter bank (conflict-free rows)
__global__ __launch_bounds__(256) void conv1_fwd_bf16_kernel(ConvArgs a, const unsigned short* lut16) {
    __shared__ __attribute__((aligned(16))) unsigned short s_in[C1::NPIX * 4];
    __shared__ __attribute__((aligned(16))) unsigned short s_w[16 * C1_WS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i = lane & 15, kq = lane >> 4;
    unsigned short* g_out = (unsigned short*)a.out;
    for (int e = tid; e < 16 * C1_WS; e += 256) {        // K index k = tap*4 + ci (ci == 3 and k >= 36 are zero columns)
        const int j = e / C1_WS, k = e % C1_WS, tap = k / 4, ci = k % 4;
        s_w[e] = f2bf((k < 36 && ci < 3) ? a.w[(j * 9 + tap) * 3 + ci] : 0.f);
    }
    for (int e = tid; e < C1::NPIX * 4; e += 256) s_in[e] = 0;
    const float bias = a.bias ? a.bias[i] : 0.f;
    int off0, off1;                                        // this lane's two taps of the first MFMA
    { const int t0 = 2 * kq, t1 = 2 * kq + 1; off0 = ((t0 / 3) * C1::PW + t0 % 3) * 4; off1 = ((t1 / 3) * C1::PW + t1 % 3) * 4; }
    constexpr int off8 = (2 * C1::PW + 2) * 4;
    const int nwork = a.n * C1::TPI;
    uint32_t regs[C1::NLD];
    auto item = [&](int w) { return w < nwork ? w : nwork - 1; };          // past the end: the last item again (loads stay unconditional)
    int frame_next = 0;
    if ((int)blockIdx.x < nwork) {
        c1_load<C1>(regs, (const uint8_t*)a.in, c1_frame(a.idx, a.in_base, blockIdx.x / C1::TPI), (blockIdx.x % C1::TPI) * C1::TH);
        frame_next = c1_frame_fetch(a.idx, a.in, item(blockIdx.x + gridDim.x) / C1::TPI);
    }
    for (int work = blockIdx.x; work < nwork; work += gridDim.x) {
        const int img = work / C1::TPI, ty0 = (work % C1::TPI) * C1::TH;
        __syncthreads();
        c1_store<C1>(regs, s_in, ty0);
        __syncthreads();
        { const int w2 = item(work + gridDim.x); c1_load<C1>(regs, (const uint8_t*)a.in, c1_frame_of(a.idx, a.in_base, frame_next, w2 / C1::TPI), (w2 % C1::TPI) * C1::TH);
          frame_next = c1_frame_fetch(a.idx, a.in, item(work + 2 * gridDim.x) / C1::TPI); }
        const bf16x8 bw1 = *(const bf16x8*)(s_w + i * C1_WS + kq * 8);
        const bf16x8 bw2 = *(const bf16x8*)(s_w + i * C1_WS + 32 + kq * 8);
#pragma unroll
        for (int mt = 0; mt < 8; ++mt) {
            const int pl = (wave * 8 + mt) * 16 + i, y = pl / C1::TW, x = pl % C1::TW;
            const unsigned short* p = s_in + (y * C1::PW + x) * 4;
            const uint2 lo = *(const uint2*)(p + off0), hi = *(const uint2*)(p + off1), t8 = *(const uint2*)(p + off8);
            const bf16x8 a1 = __builtin_bit_cast(bf16x8, (uint4){lo.x, lo.y, hi.x, hi.y});
            const bf16x8 a2 = __builtin_bit_cast(bf16x8, (uint4){t8.x, t8.y, 0u, 0u});
            f32x4 acc = {bias, bias, bias, bias};                 // the bias is the accumulator's initial value (as in the fused conv + pool kernel)
            acc = MFMA_BF16(a1, bw1, acc);
            acc = MFMA_BF16(a2, bw2, acc);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int po = (wave * 8 + mt) * 16 + kq * 4 + r;
                g_out[(((long long)img * C1::HW + ty0 + po / C1::TW) * C1::HW + po % C1::TW) * 16 + i] = f2bf(acc[r]);
            }
        }
    }
}

// block1.conv + MaxPool2d(3,2,1) in one launch: a work item produces 4 pooled rows (x 32 x 16 ch) from 9 conv rows
// (one halo row recomputed), which only ever exist in LDS -- the 64x64x16 conv output, the largest tensor of the
// network, is neither written nor re-read (the backward pass needs the pooled arg-max, not the conv output).
using C1P = C1T<9>;
#ifndef C1P_SC
#define C1P_SC 24         // conv-output tile pixel stride in LDS (bf16 elements)
#endif
// block1.conv forward bank in the LDS layout of the conv1 kernels: [16 filters][C1_WS], K index k = tap*4 + ci (ci == 3 and k >= 36 zero)
__global__ void pack_conv1_bank_kernel(const float* __restrict__ w, unsigned short* __restrict__ bank) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= 16 * C1_WS) return;
    const int j = e / C1_WS, k = e % C1_WS, tap = k / 4, ci = k % 4;
    bank[e] = f2bf((k < 36 && ci < 3) ? w[(j * 9 + tap) * 3 + ci] : 0.f);
}
void launch_pack_conv1_bank(const float* w, unsigned short* bank, hipStream_t st) {
    hipLaunchKernelGGL(pack_conv1_bank_kernel, dim3((16 * C1_WS + 255) / 256), dim3(256), 0, st, w, bank);
}
int conv1_bank_elems() { return 16 * C1_WS; }
// Every bf16 image of the parameters in ONE launch after an optimizer step (three launches before): blocks [0, 8 n_desc) the conv filter
// banks (pack_banks_kernel's element loop, 8 blocks per bank), the next few block1.conv's bank, the rest embedder.fc's two packed
// images ([256][2048] for the forward, [2048][256] for the data gradient; fc_bf16.hip).
__global__ void repack_all_kernel(const float* __restrict__ params, unsigned short* __restrict__ banks, const BankDesc* __restrict__ desc, int n_desc,
                                  const float* __restrict__ c1_w, unsigned short* __restrict__ c1_bank,
                                  const float* __restrict__ fc_w, unsigned short* __restrict__ fc_wp, unsigned short* __restrict__ fc_wt) {
    constexpr int C1B = (16 * C1_WS + 255) / 256;
    int b = blockIdx.x;
    if (b < 8 * n_desc) {
        const BankDesc d = desc[b >> 3];
        const float* w = params + d.w_off;                // forward layout [co_f][9][ci_f]
        unsigned short* out = banks + d.out_off;
        for (int e = (b & 7) * 256 + threadIdx.x; e < d.rows * d.ws; e += 8 * 256) {
            const int j = e / d.ws, k = e % d.ws, tap = k / d.cin_pass, ci = k % d.cin_pass;
            float v = 0.f;
            if (k < d.nk * 32 && tap < 9) v = d.transw ? w[(ci * 9 + (8 - tap)) * d.ci_f + j] : w[(j * 9 + tap) * d.ci_f + ci];
            out[e] = f2bf(v);
        }
        return;
    }
    b -= 8 * n_desc;
    if (b < C1B) {
        const int e = b * 256 + threadIdx.x;
        if (c1_bank && e < 16 * C1_WS) {
            const int j = e / C1_WS, k = e % C1_WS, tap = k / 4, ci = k % 4;
            c1_bank[e] = f2bf((k < 36 && ci < 3) ? c1_w[(j * 9 + tap) * 3 + ci] : 0.f);
        }
        return;
    }
    b -= C1B;
    const int e = b * 256 + threadIdx.x;                   // embedder.fc: N = 256 rows of K = 2048
    if (e < 256 * 2048) {
        const unsigned short h = f2bf(fc_w[e]);
        fc_wp[e] = h;                                      // [n][k]
        fc_wt[(long long)(e % 2048) * 256 + e / 2048] = h; // [k][n]
    }
}
void launch_repack_all(const float* params, unsigned short* banks, const BankDesc* d_desc, int n_desc, const float* c1_w, unsigned short* c1_bank,
                       const float* fc_w, unsigned short* fc_wp, unsigned short* fc_wt, hipStream_t st) {
    const int blocks = 8 * n_desc + (16 * C1_WS + 255) / 256 + 256 * 2048 / 256;
    hipLaunchKernelGGL(repack_all_kernel, dim3(blocks), dim3(256), 0, st, params, banks, d_desc, n_desc, c1_w, c1_bank, fc_w, fc_wp, fc_wt);
}
__global__ __launch_bounds__(256, 4) void conv1_pool_fwd_bf16_kernel(ConvArgs a, const unsigned short* lut16, unsigned short* p_out, uint8_t* p_arg) {      // 4 waves per SIMD: <= 128 registers (140 cost a workgroup per CU)
    __shared__ __attribute__((aligned(16))) unsigned short s_in[C1P::NPIX * 4];
    __shared__ __attribute__((aligned(16))) unsigned short s_w[16 * C1_WS];
    // conv-output tile as ORDER-PRESERVING KEYS (common.h), [9 rows][1 pad + 64 px][16 ch]: the pad cell is column -1 of its row
    // (minimal keys, written once).  Pixel stride 24 elements = 1.5 pixels: the pooling reads (lanes two pixels apart) then hit every
    // bank once -- 5.9 vs 6.1 ms per iteration with the key-based pooling (with the float compare chain it had measured slower).
    __shared__ __attribute__((aligned(16))) unsigned short s_c[9 * 65 * C1P_SC];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i = lane & 15, kq = lane >> 4;
#ifdef WG_TIMING
    long long tacc_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast_ = clock64();
#endif
    if (a.wbank) {                                         // pre-packed by pack_conv1_bank_kernel: one 16-byte load per thread (the rollout-sized
        if (tid < 16 * C1_WS / 8) ((uint4*)s_w)[tid] = ((const uint4*)a.wbank)[tid];      // launch runs this prologue for 2 items only)
    } else {
        for (int e = tid; e < 16 * C1_WS; e += 256) {
            const int j = e / C1_WS, k = e % C1_WS, tap = k / 4, ci = k % 4;
            s_w[e] = f2bf((k < 36 && ci < 3) ? a.w[(j * 9 + tap) * 3 + ci] : 0.f);
        }
    }
    static_assert((C1P::NPIX * 4) % 8 == 0, "16-byte zero fill");
    for (int e = tid; e < C1P::NPIX * 4 / 8; e += 256) ((uint4*)s_in)[e] = (uint4){0u, 0u, 0u, 0u};
    if (tid < 9 * 8) ((unsigned*)s_c)[(tid >> 3) * 65 * (C1P_SC / 2) + (tid & 7)] = MI_KEY_MIN2;
    f32x4 bias4;                                           // output channels 4*kq .. 4*kq+3 (the MFMA's row quad): the accumulators' initial value
#pragma unroll
    for (int r = 0; r < 4; ++r) bias4[r] = a.bias ? a.bias[kq * 4 + r] : 0.f;
    int off0, off1;
    { const int t0 = 2 * kq, t1 = 2 * kq + 1; off0 = ((t0 / 3) * C1P::PW + t0 % 3) * 4; off1 = ((t1 / 3) * C1P::PW + t1 % 3) * 4; }
    constexpr int off8 = (2 * C1P::PW + 2) * 4;
    const int nwork = a.n * 8;                             // 8 groups of 4 pooled rows per image
    // Two staging register sets: the frame rows of item k + 2 are requested while item k is processed.  An item lasts about one loaded
    // memory round trip, so with a single set the staging phase waited for its own data (20 % of the phase clocks).
    uint32_t regs0[C1P::NLD], regs1[C1P::NLD];
    auto item = [&](int w) { return w < nwork ? w : nwork - 1; };          // past the end: the last item again (loads stay unconditional)
    const int G = gridDim.x;
    int frame_next = 0;
    if ((int)blockIdx.x < nwork) {
        const int w0 = blockIdx.x, w1 = item(w0 + G);
        c1_load<C1P>(regs0, (const uint8_t*)a.in, c1_frame(a.idx, a.in_base, w0 / 8), (w0 % 8) * 8 - 1);
        c1_load<C1P>(regs1, (const uint8_t*)a.in, c1_frame(a.idx, a.in_base, w1 / 8), (w1 % 8) * 8 - 1);
        frame_next = c1_frame_fetch(a.idx, a.in, item(w0 + 2 * G) / 8);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                   // the prologue's index loads are the youngest: drain once here, not in the loop
    }
    auto step = [&](int work_, uint32_t (&regs)[C1P::NLD]) {
        const int work = __builtin_amdgcn_readfirstlane(work_);               // uniform (a lambda parameter would live in a vector register)
        const int img = work / 8, oy0 = (work % 8) * 4, cy0 = 2 * oy0 - 1;      // first conv row of this item (may be -1)
        TCK(0);
        __syncthreads();
        TCK(1);
        c1_store<C1P>(regs, s_in, cy0);
        TCK(2);
        __syncthreads();
        TCK(3);
        { const int w2 = item(work + 2 * G); c1_load<C1P>(regs, (const uint8_t*)a.in, c1_frame_of(a.idx, a.in_base, frame_next, w2 / 8), (w2 % 8) * 8 - 1);
          frame_next = c1_frame_fetch(a.idx, a.in, item(work + 3 * G) / 8); }
        TCK(4);
        const bf16x8 bw1 = *(const bf16x8*)(s_w + i * C1_WS + kq * 8);
        const bf16x8 bw2 = *(const bf16x8*)(s_w + i * C1_WS + 32 + kq * 8);
        // operands swapped (A = filter rows, B = pixels): the lane then holds 4 CONSECUTIVE channels of one pixel -> one 8-byte
        // LDS store per tile.  All 9 tiles accumulate into their own registers first (18 independent MFMAs back to back), the
        // epilogues follow: with a single accumulator every tile waited for the previous tile's read-out.
#ifndef C1P_GROUP
#define C1P_GROUP 3          // tiles accumulated together before their epilogues (scratch/kbench.hip, per 8192 frames: 1 -> 258-262 us, 3 -> 252-259,
                             // 9 -> 279-282 with the key-based pooling; with the float compare chain it was 328 / 403 / 383)
#endif
#pragma unroll
        for (int g0 = 0; g0 < 9; g0 += C1P_GROUP) {
            f32x4 acc[C1P_GROUP];
#pragma unroll
            for (int m = 0; m < C1P_GROUP; ++m) {
                const int mt = g0 + m;
                const int pl = (wave * 9 + mt) * 16 + i, y = pl / 64, x = pl % 64;
                const unsigned short* p = s_in + (y * C1P::PW + x) * 4;
                const uint2 lo = *(const uint2*)(p + off0), hi = *(const uint2*)(p + off1), t8 = *(const uint2*)(p + off8);
                acc[m] = MFMA_BF16(bw1, __builtin_bit_cast(bf16x8, (uint4){lo.x, lo.y, hi.x, hi.y}), bias4);
                acc[m] = MFMA_BF16(bw2, __builtin_bit_cast(bf16x8, (uint4){t8.x, t8.y, 0u, 0u}), acc[m]);
            }
#pragma unroll
            for (int m = 0; m < C1P_GROUP; ++m) {
                const int t = wave * 9 + g0 + m;
                const unsigned k0 = mi_bf16x2_to_keys(mi_pk_bf16(acc[m][0], acc[m][1]));
                const unsigned k1 = mi_bf16x2_to_keys(mi_pk_bf16(acc[m][2], acc[m][3]));
                *(uint2*)(s_c + (t * 16 + i + (t >> 2) + 1) * C1P_SC + kq * 4) = (uint2){k0, k1};
            }
        }
        // tiles 0..3 = conv row 0 of the item, outside the image when cy0 == -1 (an image's first item): minimal keys over what wave 0 has
        // just written there (its own LDS stores, in order) -- one branch per item instead of two selects per tile
        if (cy0 < 0 && wave == 0) {
            unsigned* q = (unsigned*)(s_c + (lane + 1) * C1P_SC);
#pragma unroll
            for (int w = 0; w < 8; ++w) q[w] = MI_KEY_MIN2;
        }
        TCK(5);
        __syncthreads();
        TCK(6);
        {   // pooling: thread = (pooled row 0..3, pooled col 0..31, 8-channel half); 9 window reads, 4 v_max3_i32 per channel
            const int c8 = tid & 1, ox = (tid >> 1) & 31, oyl = tid >> 6;
            uint4 u[9];
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx)
                    u[ky * 3 + kx] = *(const uint4*)(s_c + ((2 * oyl + ky) * 65 + 2 * ox + kx) * C1P_SC + c8 * 8);
            const size_t o = ((((size_t)img * 32 + oy0 + oyl) * 32 + ox) * 2 + c8) * 8;
            uint4 pk;
            uint2 ar;
            mi_pool9_keys(u, pk, ar);
            *(uint4*)(p_out + o) = pk;
            *(uint2*)(p_arg + o) = ar;
        }
        TCK(7);
    };
    int work = blockIdx.x;
    for (; work + G < nwork; work += 2 * G) { step(work, regs0); step(work + G, regs1); }      // (a conditional second step would merge two load histories)
    if (work < nwork) step(work, regs0);
#ifdef WG_TIMING
    if (tid == 0) for (int k = 0; k < 8; ++k) atomicAdd(&g_wg_timing[k], (unsigned long long)tacc_[k]);
#endif
}
void launch_conv1_pool_fwd_bf16(const ConvArgs& a, const unsigned short* lut16, void* p_out, uint8_t* p_arg, hipStream_t st) {
    const int w = a.n * 8, grid = w > 1024 ? 1024 : w;
    if (grid < 1) return;
    hipLaunchKernelGGL(conv1_pool_fwd_bf16_kernel, dim3(grid), dim3(256), 0, st, a, lut16, (unsigned short*)p_out, p_arg);
}

// POOLED: a.dout is the gradient of the POOLED map (32x32x16) and a.pool_arg its arg-max bytes; the conv-output
// gradient tile is rebuilt in LDS from the 3 pooled rows that touch the 4 conv rows of the item (max-pool backward
// fused into the staging: no 64x64x16 gradient tensor in HBM).
template <bool POOLED>
__global__ __launch_bounds__(256) void conv1_wgrad_bf16_kernel(WgradArgs a, const unsigned short* lut16) {
    extern __shared__ __attribute__((aligned(16))) unsigned short smem_h[];
    unsigned short* s_in = smem_h;                                   // [396][4]: 3 channels + 1.0
    unsigned short* s_do = smem_h + ((C1W::NPIX * 4 + 7) / 8) * 8;    // [256][16]
    unsigned short* s_pd = s_do + C1W::NT * 16;                       // POOLED: [3][32][16] pooled gradient
    uint8_t* s_pa = (uint8_t*)(s_pd + 3 * 32 * 16);                   // POOLED: [3][32][16] arg-max bytes
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i = lane & 15, kq = lane >> 4, rq = (lane & 15) >> 2, cp = lane & 3;
    const unsigned short* g_do = (const unsigned short*)a.dout;
#ifdef WG_TIMING
    long long tacc_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast_ = clock64();
#endif
    f32x4 acc[3] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    for (int e = tid; e < C1W::NPIX * 4; e += 256) s_in[e] = 0;
    const int nwork = a.n * C1W::TPI;
    // Two staging register sets: the loads of item k+2 are issued while item k is processed.  One item lasts ~2.5 us, about one
    // loaded-HBM round trip, so with a single set every item waited for its own data (phases "store" + "issue" = 33 % of the clocks).
    constexpr int NDO = C1W::NT * 2 / 256;                            // NT px x 2 chunks of 8 channels
    struct Stage { uint32_t regs[C1W::NLD]; uint4 rdo[NDO]; uint2 rpa; };
    Stage S0, S1;
    auto load_do = [&](Stage& S, int img, int ty0) {
        if (POOLED) {                                               // 3 pooled rows x 32 x 2 halves = 192 (uint4, uint2) pairs; unconditional,
            const int pr = tid >> 6, oy = ty0 / 2 + (pr < 2 ? pr : 2);   // clamped: rows past the map are replaced when the registers are stored
            const size_t o = (((size_t)img * 32 + (oy < 32 ? oy : 31)) * 32) * 16 + (size_t)(tid & 63) * 8;
            S.rdo[0] = *(const uint4*)(g_do + o);
            S.rpa = *(const uint2*)(a.pool_arg + o);
        } else {
#pragma unroll
            for (int k = 0; k < NDO; ++k) {
                const int e = tid + k * 256, pl = e >> 1, c8 = e & 1;
                S.rdo[k] = *(const uint4*)(g_do + (((long long)img * C1W::HW + ty0 + pl / C1W::TW) * C1W::HW + pl % C1W::TW) * 16 + c8 * 8);
            }
        }
    };
    auto item = [&](int w) { return w < nwork ? w : nwork - 1; };          // past the end: the last item again (loads stay unconditional)
    const int G = gridDim.x;
    int frame_next = 0;                                                     // fetched index of item (current + 2G), see c1_frame_fetch
    {
        const int w0 = blockIdx.x, w1 = item(w0 + G);
        c1_load<C1W>(S0.regs, (const uint8_t*)a.in, c1_frame(a.idx, a.in_base, w0 / C1W::TPI), (w0 % C1W::TPI) * C1W::TH); load_do(S0, w0 / C1W::TPI, (w0 % C1W::TPI) * C1W::TH);
        c1_load<C1W>(S1.regs, (const uint8_t*)a.in, c1_frame(a.idx, a.in_base, w1 / C1W::TPI), (w1 % C1W::TPI) * C1W::TH); load_do(S1, w1 / C1W::TPI, (w1 % C1W::TPI) * C1W::TH);
        frame_next = c1_frame_fetch(a.idx, a.in, item(w0 + 2 * G) / C1W::TPI);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                   // the prologue's index fetch is the youngest load: drain once here, not in the loop
    }
    auto step = [&](int work, Stage& S) {
        TCK(0);
        __syncthreads();
        TCK(1);
        c1_store<C1W, true>(S.regs, s_in, (work % C1W::TPI) * C1W::TH);
        if (POOLED) {
            if (tid < 192) {
                const bool in = (work % C1W::TPI) * C1W::TH / 2 + (tid >> 6) < 32;      // the third pooled row of an image's last item does not exist
                *(uint4*)(s_pd + tid * 8) = in ? S.rdo[0] : (uint4){0u, 0u, 0u, 0u}; *(uint2*)(s_pa + tid * 8) = in ? S.rpa : (uint2){0xffffffffu, 0xffffffffu};
            }
        } else {
#pragma unroll
            for (int k = 0; k < NDO; ++k) { const int e = tid + k * 256; *(uint4*)(s_do + (e >> 1) * 16 + (e & 1) * 8) = S.rdo[k]; }
        }
        TCK(2);
        __syncthreads();
        {
            const int w2 = item(work + 2 * G), img = w2 / C1W::TPI, ty0 = (w2 % C1W::TPI) * C1W::TH;
            const long long frame = c1_frame_of(a.idx, a.in_base, frame_next, img);
            frame_next = c1_frame_fetch(a.idx, a.in, item(work + 3 * G) / C1W::TPI);       // ahead of the pixel loads: vmcnt retires in order, and the
            c1_load<C1W>(S.regs, (const uint8_t*)a.in, frame, ty0); load_do(S, img, ty0);   // next step needs this index but not these pixels
        }
        TCK(3);
        if (POOLED) {       // max-pool backward into the LDS tile.  A thread owns a 2x2 block of conv pixels x 4 channels: the
            // 4 windows (oy, ox) in {a, a+1} x {b, b+1} that touch the block are read once each and their 9 (window, pixel)
            // incidences -- one per pool position -- are resolved with compile-time (ky, kx); windows are added in
            // (oy, ox) order, the order of the stand-alone max-pool backward kernel.
            const int cq = tid & 3, bx = (tid >> 2) & 31, by = tid >> 7;
            float sm[4][4];
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int c = 0; c < 4; ++c) sm[q][c] = 0.f;
#pragma unroll
            for (int wy = 0; wy < 2; ++wy)
#pragma unroll
                for (int wx = 0; wx < 2; ++wx) {
                    const int o = ((by + wy) * 32 + bx + wx) * 16 + cq * 4;
                    const uint2 d = *(const uint2*)(s_pd + o);
                    unsigned ag = *(const unsigned*)(s_pa + o);
                    if (wx == 1 && bx == 31) ag = 0xffffffffu;                 // window column 32 does not exist
                    const float v[4] = {__uint_as_float(d.x << 16), __uint_as_float(d.x & 0xffff0000u), __uint_as_float(d.y << 16), __uint_as_float(d.y & 0xffff0000u)};
#pragma unroll
                    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                        for (int dx = 0; dx < 2; ++dx) {
                            const int ky = dy + 1 - 2 * wy, kx = dx + 1 - 2 * wx;
                            if (ky < 0 || kx < 0) continue;
                            const unsigned pos = (unsigned)(ky * 3 + kx);
#pragma unroll
                            for (int c = 0; c < 4; ++c)
                                sm[dy * 2 + dx][c] += (((ag >> (8 * c)) & 0xffu) == pos) ? v[c] : 0.f;
                        }
                }
#pragma unroll
            for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                for (int dx = 0; dx < 2; ++dx) {
                    const float* q = sm[dy * 2 + dx];
                    *(uint2*)(s_do + ((2 * by + dy) * 64 + 2 * bx + dx) * 16 + cq * 4) =
                        (uint2){mi_pk_bf16(q[0], q[1]), mi_pk_bf16(q[2], q[3])};
                }
        }
        TCK(4);
        TCK(5);
        __syncthreads();
        // B operand straight from the staged tile: the 16 columns of an MFMA are 4 taps x (3 channels + the 1.0), column group cp of
        // MFMA m reading pixel + tap(4m + cp) -- no im2col copy.  Three MFMAs cover the 9 taps (the last one's groups 1-3 repeat tap 8).
        for (int t = wave; t < C1W::NT / 32; t += 4) {
            int row[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) row[h] = 32 * t + 16 * (kq >> 1) + 8 * h + 4 * (kq & 1) + rq;
            const s16x4 alo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(s_do + row[0] * 16 + 4 * cp));
            const s16x4 ahi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(s_do + row[1] * 16 + 4 * cp));
            const bf16x8 av = __builtin_shufflevector(alo, ahi, 0, 1, 2, 3, 4, 5, 6, 7);
            int prow[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) prow[h] = (row[h] / C1W::TW) * C1W::PW + row[h] % C1W::TW;      // tile coordinates: tap (0, 0) = this cell
#pragma unroll
            for (int m = 0; m < 3; ++m) {
                const int tap = (4 * m + cp) < 9 ? 4 * m + cp : 8, toff = (tap / 3) * C1W::PW + tap % 3;
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(s_in + (prow[0] + toff) * 4));
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(s_in + (prow[1] + toff) * 4));
                acc[m] = MFMA_BF16(av, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7), acc[m]);
            }
        }
        TCK(6);
    };
    int work = blockIdx.x;
    for (; work + G < nwork; work += 2 * G) { step(work, S0); step(work + G, S1); }      // (a conditional second step would merge two load histories:
    if (work < nwork) step(work, S0);                                                   //  the compiler then drains vmcnt to 0 at every wait)
#ifdef WG_TIMING
    if (tid == 0) for (int k = 0; k < 8; ++k) atomicAdd(&g_wg_timing[k], (unsigned long long)tacc_[k]);
#endif
    __syncthreads();
    float* red = (float*)smem_h;                 // 432 weights + 16 bias sums (the ones column)
    for (int w = 0; w < 4; ++w) {
        if (wave == w) {
#pragma unroll
            for (int m = 0; m < 3; ++m)
#pragma unroll
                for (int r = 0; r < 4; ++r) {          // column i of MFMA m = (tap 4m + i/4, channel i%4); channel 3 of the centre tap = bias sum
                    const int tap = 4 * m + (i >> 2), ci = i & 3;
                    if (tap < 9 && (ci < 3 || tap == 4)) {
                        const int o = (ci < 3) ? (kq * 4 + r) * 27 + tap * 3 + ci : 432 + kq * 4 + r;
                        red[o] = (w == 0) ? acc[m][r] : red[o] + acc[m][r];
                    }
                }
        }
        __syncthreads();
    }
    float* slab = a.partial + (long long)blockIdx.x * 448;
    for (int e = tid; e < 448; e += 256) slab[e] = red[e];
}
// ---- block1.conv weight gradient from the POOLED gradient in ONE-HOT form (round 3; replaces the gather of conv1_wgrad_bf16_kernel<true>)
// dW[co][tap][ci] = sum_p dC[p][co] x[p + tap][ci], and the max-pool backward is dC[p][co] = sum of g[w][co] over the windows w whose
// arg-max is p.  Substituted:  dW = sum_{w, pos} T[w][pos][co] x[pixel(w, pos) + tap][ci]  with  T[w][pos][co] = (arg[w][co] == pos) ?
// g[w][co] : 0  -- the pool backward becomes a one-hot EXPANSION of the contraction index (9 K-rows per window: 9216 per image instead
// of 4096 pixels) whose operand addresses are regular: pixel(w, pos) = (2 oy - 1 + ky, 2 ox - 1 + kx) does not depend on the data.  No
// gathered dC tile, no compare / select / add per (window position, channel): the old kernel spent ~75 vector instructions per conv
// pixel there and was VALU-issue-bound at 2.6 TB/s with 8 % of the matrix pipe busy.  Numerically the products g * x are summed in fp32
// directly (the old form rounded the <= 4 coinciding contributions of a pixel to one bf16 first): closer to the fp32 reference.
//   * item = 4 pooled rows of one image (8 items per image); wave r owns pooled row r: 8 K-steps of 4 windows x positions 0..7 and one
//     K-step of 32 windows x position 8;
//   * positions 0..7 -- the A operand (16 channels x 32 K-rows) is built in REGISTERS: lane (co, kq) reads g and arg of window 4 s + kq
//     (two small LDS reads) and places g in the slot arg of its 8; position 8 -- a second one-hot image T8[w][co], written once per item;
//   * B operand = frame pixels through ds_read_b64_tr_b16 as in the old kernel: lane (rq, cp) of quarter kq supplies the 8-byte LDS row of
//     pixel(window, position rq | 4 + rq) + tap(4 m + cp); the 16 columns of MFMA m are 4 taps x (3 channels + the 1.0 that sums the bias);
//   * two LDS tile sets and two staging register sets: item k + 1 is stored while item k is multiplied, loads run two items ahead,
//     one barrier per item.
#ifndef C1H_TWAVE
#define C1H_TWAVE 0
#endif
#ifdef WG_TIMING
#define TCKH(k) do { if (tid == 64 * C1H_TWAVE) { const long long now_ = clock64(); tacc_[k] += now_ - tlast_; tlast_ = now_; } } while (0)
#else
#define TCKH(k) do { } while (0)
#endif
#ifndef C1H_PW
#define C1H_PW 72                  // pixels per tile row: 72 x 8 bytes = 16 banks more per row -- the <= 3 rows x 7 pixels a half-wave's B read touches fall on distinct banks
#endif
struct C1H {
    static constexpr int R = 4, TPI = 32 / R;                  // pooled rows per item, items per image
    static constexpr int PH = 2 * R + 3, PW = C1H_PW;           // tile row 0 = image row 2 oy0 - 2, tile column 0 = image column -2 (68 needed)
    static constexpr int NTASK = PH * 16;                      // 12-byte frame pieces (4 pixels) of an item
    static constexpr int IN_BYTES = PH * PW * 8, G_BYTES = R * 32 * 16 * 2, A_BYTES = R * 32 * 16, T8_BYTES = G_BYTES;
    static constexpr int SET_BYTES = IN_BYTES + G_BYTES + A_BYTES + T8_BYTES;
    static_assert(NTASK <= 256 && R * 32 * 2 == 256, "one frame piece and one 8-channel gradient chunk per thread");
    static_assert(IN_BYTES % 16 == 0 && SET_BYTES % 16 == 0, "16-byte aligned tile sets");
};
typedef unsigned c1h_u32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void conv1_wgrad_onehot_bf16_kernel(WgradArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_c1h[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i = lane & 15, kq = lane >> 4, rq = i >> 2, cp = i & 3;
    const unsigned short* g_dp = (const unsigned short*)a.dout;
    const int nwork = a.n * C1H::TPI, G = gridDim.x;
    if ((int)blockIdx.x >= nwork) {                             // (the slab count is the old kernel's: min(1024, 16 n) >= 8 n)
        float* slab = a.partial + (long long)blockIdx.x * 448;
        for (int e = tid; e < 448; e += 256) slab[e] = 0.f;
        return;
    }
#ifdef WG_TIMING
    long long tacc_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast_ = clock64();
#endif
    f32x4 acc[3] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    for (int e = tid; e < 2 * C1H::SET_BYTES / 16; e += 256) ((uint4*)smem_c1h)[e] = (uint4){0u, 0u, 0u, 0u};      // halo columns stay zero
    struct Stage { uint32_t px[3]; uint4 g; uint2 ar; };
    Stage S0, S1;
    const int ftid = tid;          // thread of frame piece ftid (rotating the three waves that hold the 176 pieces with the workgroup: no change, 184-198 us)
    auto item = [&](int w) { return w < nwork ? w : nwork - 1; };          // past the end: the last item again (loads stay unconditional)
    auto load = [&](Stage& S, int w, long long frame) {
        const int img = w / C1H::TPI, oy0 = (w % C1H::TPI) * C1H::R;
        const int row = (ftid >> 4) < C1H::PH ? (ftid >> 4) : C1H::PH - 1, gy = 2 * oy0 - 2 + row, gyc = gy < 0 ? 0 : (gy > 63 ? 63 : gy);
        const uint32_t* p = (const uint32_t*)((const uint8_t*)a.in + frame * (64 * 64 * 3) + gyc * 192 + (ftid & 15) * 12);
        S.px[0] = p[0]; S.px[1] = p[1]; S.px[2] = p[2];
        const size_t o = ((size_t)img * 32 + oy0) * 512 + (size_t)tid * 8;
        S.g = *(const uint4*)(g_dp + o); S.ar = *(const uint2*)(a.pool_arg + o);
    };
    auto store = [&](const Stage& S, unsigned char* set, int w) {
        const int oy0 = (w % C1H::TPI) * C1H::R;
        if (ftid < C1H::NTASK) {
            unsigned short* s_in = (unsigned short*)set;
            const int row = ftid >> 4, gy = 2 * oy0 - 2 + row;
            const bool in = gy >= 0 && gy < 64;
            const uint32_t px[3] = {in ? S.px[0] : 0u, in ? S.px[1] : 0u, in ? S.px[2] : 0u};      // rows outside the image: zeros
            mi_f32x2 f[6];
#pragma unroll
            for (int b = 0; b < 6; ++b)
                f[b] = (mi_f32x2){(float)((px[(2 * b) >> 2] >> (8 * ((2 * b) & 3))) & 0xffu), (float)((px[(2 * b + 1) >> 2] >> (8 * ((2 * b + 1) & 3))) & 0xffu)}
                       * (mi_f32x2){1.0f / 255.0f, 1.0f / 255.0f};
            auto fv = [&](int b) { return (b & 1) ? f[b >> 1].y : f[b >> 1].x; };
#pragma unroll
            for (int j = 0; j < 4; ++j)
                *(uint2*)(s_in + (row * C1H::PW + 2 + (ftid & 15) * 4 + j) * 4) = (uint2){mi_pk_bf16(fv(3 * j), fv(3 * j + 1)), mi_pk_bf16(fv(3 * j + 2), 1.f)};
        }
        *(uint4*)(set + C1H::IN_BYTES + tid * 16) = S.g;
        *(uint2*)(set + C1H::IN_BYTES + C1H::G_BYTES + tid * 8) = S.ar;
        // position 8 <=> bit 3 of the arg byte (values 0..8): byte masks 0x00 / 0xff, each spread over its channel's 16-bit half
        const unsigned m0 = (S.ar.x >> 3) & 0x01010101u, m1 = (S.ar.y >> 3) & 0x01010101u, mm0 = (m0 << 8) - m0, mm1 = (m1 << 8) - m1;
        const uint4 t8 = {S.g.x & __builtin_amdgcn_perm(mm0, mm0, 0x01010000u), S.g.y & __builtin_amdgcn_perm(mm0, mm0, 0x03030202u),
                          S.g.z & __builtin_amdgcn_perm(mm1, mm1, 0x01010000u), S.g.w & __builtin_amdgcn_perm(mm1, mm1, 0x03030202u)};
        *(uint4*)(set + C1H::IN_BYTES + C1H::G_BYTES + C1H::A_BYTES + tid * 16) = t8;
    };
    // lane constants (byte offsets inside a tile set); this wave's pooled row = wave
    int offB[2][3], off8B[3];
#pragma unroll
    for (int m = 0; m < 3; ++m) {
        const int t = (4 * m + cp) < 9 ? 4 * m + cp : 8, ty = t / 3, tx = t % 3;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int pos = 4 * h + rq, ky = pos / 3, kx = pos % 3;
            offB[h][m] = ((2 * wave + ky + ty) * C1H::PW + 2 * kq + kx + tx) * 8;
        }
        off8B[m] = ((2 * wave + 2 + ty) * C1H::PW + 2 * (8 * kq + rq) + 2 + tx) * 8;
    }
    const int offG = C1H::IN_BYTES + ((wave * 32 + kq) * 16 + i) * 2, offA = C1H::IN_BYTES + C1H::G_BYTES + (wave * 32 + kq) * 16 + i;
    const int off8A = C1H::IN_BYTES + C1H::G_BYTES + C1H::A_BYTES + ((wave * 32 + 8 * kq + rq) * 16 + 4 * cp) * 2;
    // operand fragments of one K-step, fetched one step ahead of their MFMAs (the LDS latency of a step hides behind the previous one)
    struct Frag { s16x4 lo[3], hi[3]; unsigned g16, ab; s16x4 alo, ahi; };
    auto fetch = [&](Frag& F, const unsigned char* set, int s) {
        if (s < 8) {                                            // windows 4 s + kq, positions 0..7
            F.g16 = *(const unsigned short*)(set + offG + s * 128); F.ab = *(const unsigned char*)(set + offA + s * 64);
#pragma unroll
            for (int m = 0; m < 3; ++m) {
                F.lo[m] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(set + offB[0][m] + s * 64));
                F.hi[m] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(set + offB[1][m] + s * 64));
            }
        } else {                                                // position 8 of the row's 32 windows: slots 0..3 = windows 8 kq + rq, 4..7 = + 4
            F.alo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(set + off8A));
            F.ahi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(set + off8A + 128));
#pragma unroll
            for (int m = 0; m < 3; ++m) {
                F.lo[m] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(set + off8B[m]));
                F.hi[m] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(set + off8B[m] + 64));
            }
        }
    };
    auto mul = [&](const Frag& F, int s) {
        bf16x8 av;
        if (s < 8) {                                            // slot j of the lane's 8 = position j: g goes to slot arg
            const unsigned val = F.g16 << ((F.ab & 1u) * 16u), d = F.ab >> 1;
            const c1h_u32x4 aw = {d == 0u ? val : 0u, d == 1u ? val : 0u, d == 2u ? val : 0u, d == 3u ? val : 0u};
            av = __builtin_bit_cast(bf16x8, aw);
        } else
            av = __builtin_shufflevector(F.alo, F.ahi, 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
        for (int m = 0; m < 3; ++m) acc[m] = MFMA_BF16(av, __builtin_shufflevector(F.lo[m], F.hi[m], 0, 1, 2, 3, 4, 5, 6, 7), acc[m]);
    };
    auto compute = [&](const unsigned char* set) {
        Frag F0, F1;
        fetch(F0, set, 0);
#pragma unroll
        for (int s = 0; s < 8; s += 2) {
            fetch(F1, set, s + 1);
            __builtin_amdgcn_sched_barrier(0);
            mul(F0, s);
            fetch(F0, set, s + 2);
            __builtin_amdgcn_sched_barrier(0);
            mul(F1, s + 1);
        }
        mul(F0, 8);
    };
    unsigned char* const set0 = smem_c1h;
    unsigned char* const set1 = smem_c1h + C1H::SET_BYTES;
    int work = blockIdx.x, frame_next;
    {
        const int w1 = item(work + G), w2 = item(work + 2 * G);
        load(S0, work, c1_frame(a.idx, a.in_base, work / C1H::TPI));
        load(S1, w1, c1_frame(a.idx, a.in_base, w1 / C1H::TPI));
        __syncthreads();                                        // the zero fill
        store(S0, set0, work);
        load(S0, w2, c1_frame(a.idx, a.in_base, w2 / C1H::TPI));
        frame_next = c1_frame_fetch(a.idx, a.in, item(work + 3 * G) / C1H::TPI);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // the prologue's index loads are the youngest: drain once here, not in the loop
        __syncthreads();
    }
    // `work` is multiplied from set_cur while S (item work + G) goes to set_nxt and is re-loaded with item work + 3 G
    auto iter = [&](int work_, Stage& S, unsigned char* set_cur, unsigned char* set_nxt) {
        const int work = __builtin_amdgcn_readfirstlane(work_);      // uniform: item / image / row arithmetic on the scalar unit, the index fetch a scalar load
        TCKH(0);
        store(S, set_nxt, item(work + G));
        TCKH(1);
        {
            const int w3 = item(work + 3 * G);
            const long long frame = c1_frame_of(a.idx, a.in_base, frame_next, w3 / C1H::TPI);
            frame_next = c1_frame_fetch(a.idx, a.in, item(work + 4 * G) / C1H::TPI);
            load(S, w3, frame);
        }
        TCKH(2);
        compute(set_cur);
        TCKH(3);
        __syncthreads();
    };
    for (; work + G < nwork; work += 2 * G) { iter(work, S1, set0, set1); iter(work + G, S0, set1, set0); }
    if (work < nwork) iter(work, S1, set0, set1);
#ifdef WG_TIMING
    if (tid == 64 * C1H_TWAVE) for (int k = 0; k < 8; ++k) atomicAdd(&g_wg_timing[k], (unsigned long long)tacc_[k]);
#endif
    float* red = (float*)smem_c1h;                 // 432 weights + 16 bias sums (the ones column)
    for (int w = 0; w < 4; ++w) {
        if (wave == w) {
#pragma unroll
            for (int m = 0; m < 3; ++m)
#pragma unroll
                for (int r = 0; r < 4; ++r) {          // column i of MFMA m = (tap 4m + i/4, channel i%4); channel 3 of the centre tap = bias sum
                    const int tap = 4 * m + (i >> 2), ci = i & 3;
                    if (tap < 9 && (ci < 3 || tap == 4)) {
                        const int o = (ci < 3) ? (kq * 4 + r) * 27 + tap * 3 + ci : 432 + kq * 4 + r;
                        red[o] = (w == 0) ? acc[m][r] : red[o] + acc[m][r];
                    }
                }
        }
        __syncthreads();
    }
    float* slab = a.partial + (long long)blockIdx.x * 448;
    for (int e = tid; e < 448; e += 256) slab[e] = red[e];
}
#ifndef C1_WG_ONEHOT
#define C1_WG_ONEHOT 1             // 0: the gathering kernel conv1_wgrad_bf16_kernel<true>
#endif
#ifndef C1_WG_PAD
#define C1_WG_PAD (36 * 1024)      // requested LDS: at most 4 workgroups of the 1024 land on one CU (an even spread; the tiles need 12-16 KB)
#endif
constexpr size_t C1_WG_TILES = (size_t)(((C1W::NPIX * 4 + 7) / 8) * 8 + C1W::NT * 16) * 2;
constexpr size_t C1_WG_LDS = C1_WG_TILES + 3 * 32 * 16 * 3 > C1_WG_PAD ? C1_WG_TILES : C1_WG_PAD - 3 * 32 * 16 * 3;
constexpr size_t C1_WGP_LDS = C1_WG_LDS + 3 * 32 * 16 * 3;
static int c1_grid(int n) { const int w = n * C1W::TPI; return w > 1024 ? 1024 : w; }
void launch_conv1_fwd_bf16(const ConvArgs& a, const unsigned short* lut16, hipStream_t st) {
    const int w = a.n * C1::TPI, grid = w > 1024 ? 1024 : w;
    if (grid < 1) return;
    hipLaunchKernelGGL(conv1_fwd_bf16_kernel, dim3(grid), dim3(256), 0, st, a, lut16);
}
void launch_conv1_wgrad_bf16(const WgradArgs& a, const unsigned short* lut16, hipStream_t st) {
    static std::once_flag attr;
    std::call_once(attr, [] {
        hipFuncSetAttribute((const void*)conv1_wgrad_bf16_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)C1_WG_LDS);
        hipFuncSetAttribute((const void*)conv1_wgrad_bf16_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)C1_WGP_LDS);
        hipFuncSetAttribute((const void*)conv1_wgrad_onehot_bf16_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * C1H::SET_BYTES);
    });
    const int grid = c1_grid(a.n);
    if (grid < 1) return;
    if (a.pool_arg && C1_WG_ONEHOT) hipLaunchKernelGGL(conv1_wgrad_onehot_bf16_kernel, dim3(grid), dim3(256), 2 * C1H::SET_BYTES, st, a);
    else if (a.pool_arg) hipLaunchKernelGGL(conv1_wgrad_bf16_kernel<true>, dim3(grid), dim3(256), C1_WGP_LDS, st, a, lut16);
    else hipLaunchKernelGGL(conv1_wgrad_bf16_kernel<false>, dim3(grid), dim3(256), C1_WG_LDS, st, a, lut16);
}

// ------------------------------------------------------------------------------------------ launchers
//                        CIN COUT HW  TH  TW NIMG transW       (CIN/COUT of the PASS)
using B_16_16_32  = BfCfg<16, 16, 32,  8, 32, 1, false>;
using B_16_32_32  = BfCfg<16, 32, 32,  8, 32, 1, false>;
using B_32_32_16  = BfCfg<32, 32, 16, 16, 16, 1, false>;
using B_32_32_8   = BfCfg<32, 32,  8,  8,  8, 4, false>;
using BD_16_16_32 = BfCfg<16, 16, 32,  8, 32, 1, true>;
using BD_16_32_32 = BfCfg<32, 16, 32,  8, 32, 1, true>;
using BD_32_32_16 = BfCfg<32, 32, 16, 16, 16, 1, true>;
using BD_32_32_8  = BfCfg<32, 32,  8,  8,  8, 4, true>;

template <class C, bool POOLIN>
static constexpr size_t bf_lds_bytes() { return C::LDS_BYTES + (POOLIN ? PoolStage<C::CIN, C::HW / 2, C::TH / 2 + 3>::BYTES : 0); }
template <class C, bool POOLIN>
static int bf_grid(int n) {
    int bpc = (int)((160 * 1024) / bf_lds_bytes<C, POOLIN>());
    bpc = bpc < 1 ? 1 : (bpc > 4 ? 4 : bpc);
    int grid = (C::NIMG > 1) ? (n + C::NIMG - 1) / C::NIMG : n * C::TPI;
    return grid > 256 * bpc ? 256 * bpc : grid;
}
template <class C, bool POOLIN = false>
static void launch_bf_t(const ConvArgs& a, hipStream_t st) {
    static std::once_flag attr;
    constexpr size_t LDS = bf_lds_bytes<C, POOLIN>();
    std::call_once(attr, [] { hipFuncSetAttribute((const void*)conv3x3_bf16_kernel<C, POOLIN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS); });
    int bpc = (int)((160 * 1024) / LDS);
    bpc = bpc < 1 ? 1 : (bpc > 4 ? 4 : bpc);
    int grid = (C::NIMG > 1) ? (a.n + C::NIMG - 1) / C::NIMG : a.n * C::TPI;
    if (grid > 256 * bpc) grid = 256 * bpc;
    if (grid < 1) return;
    hipLaunchKernelGGL((conv3x3_bf16_kernel<C, POOLIN>), dim3(grid), dim3(256), LDS, st, a);
}

void launch_conv_fwd_bf16(ConvShape s, const ConvArgs& a, hipStream_t st) {
    switch (s) {
        case CS_16_16_32: launch_bf_t<B_16_16_32>(a, st); break;
        case CS_16_32_32: launch_bf_t<B_16_32_32>(a, st); break;
        case CS_32_32_16: launch_bf_t<B_32_32_16>(a, st); break;
        case CS_32_32_8:  launch_bf_t<B_32_32_8>(a, st); break;
        default: break;          // block1.conv (uint8 frames in): fp32-MFMA kernel with bf16 output, conv.hip
    }
}
void launch_conv_dgrad_bf16(ConvShape s, const ConvArgs& a, hipStream_t st) {
    switch (s) {
        case CS_16_16_32: launch_bf_t<BD_16_16_32>(a, st); break;
        case CS_16_32_32: if (a.pool_arg && a.wg_partial) launch_block2_conv_bwd(a, st);
                          else if (a.pool_arg) launch_bf_t<BD_16_32_32, true>(a, st); else launch_bf_t<BD_16_32_32>(a, st); break;
        case CS_32_32_16: if (a.pool_arg && a.wg_partial) launch_block3_conv_bwd(a, st);
                          else if (a.pool_arg) launch_bf_t<BD_32_32_16, true>(a, st); else launch_bf_t<BD_32_32_16>(a, st); break;
        case CS_32_32_8:  launch_bf_t<BD_32_32_8>(a, st); break;
        default: break;
    }
}

int conv_bwd_fused_grid(ConvShape s, int n) { return s == CS_16_32_32 ? b2bwd_grid(n) : (s == CS_32_32_16 && B3BWD_FUSED) ? b3bwd_grid(n) : -1; }
