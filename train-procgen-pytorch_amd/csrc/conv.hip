// 3x3 / stride 1 / pad 1 convolutions of the IMPALA-CNN as implicit GEMMs on the gfx950 matrix
// cores (v_mfma_f32_16x16x4_f32: exact fp32, bitwise an fmaf chain), NHWC activations.
//
// Reference semantics: nn.Conv2d(k=3,s=1,p=1) in common/model.py:138-139,152 and its autograd
// (data gradient / weight gradient) -- these kernels replace the cuDNN calls behind them.
//
// Forward / dgrad kernel (one template):
//   * a workgroup (4 waves) owns NIMG x TH x TW output pixels x all COUT channels and walks the
//     work list persistently, so the 3x3xCINxCOUT filter bank is staged into LDS once per workgroup;
//   * the haloed input tile is staged into LDS as [pixel][CINP (+4 pad)] fp32 (uint8 frames are
//     expanded through a 256-entry table: exactly the reference's obs/255.0), ReLU-on-load for the
//     pre-activation residual convs;
//   * M = 16 pixels, N = 16 output channels, K = 9*CINP.  The K order is permuted so that lane
//     quarter q = lane>>4 owns input channels [q*CINP/4, (q+1)*CINP/4) of every tap: each lane then
//     reads its A and B fragments as contiguous 16-byte LDS words (one ds_read_b128 feeds 4 MFMAs);
//   * epilogue: + bias, * (mask > 0) (dgrad through a ReLU), + residual, store NHWC.
//   dgrad is the same kernel over dOut with the weight bank staged tap-flipped and
//   channel-transposed (TRANSW).
//
// Weight-gradient kernel: M = 16 output channels, N = 16 input channels, K = pixels; one
// accumulator tile per (tap, co-block, ci-block) lives in registers across the whole persistent
// loop; the four waves are summed through LDS in a fixed order and every workgroup writes one
// slab, which reduce_slabs adds up in a fixed order (bitwise reproducible, no float atomics).
#include "common.h"
#include <mutex>

#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

template <int CIN_, int CINP_, int COUT_, int HW_, int TH_, int TW_, int NIMG_, bool IN_U8_, bool TRANSW_, bool BFIO_ = false>
struct FwdCfg {
    static constexpr int CIN = CIN_, CINP = CINP_, COUT = COUT_, HW = HW_, TH = TH_, TW = TW_, NIMG = NIMG_;
    static constexpr bool IN_U8 = IN_U8_, TRANSW = TRANSW_;
    static constexpr bool BFIO = BFIO_;     // activations in HBM are bf16 (fp32 arithmetic inside the kernel)
    // LDS floats per input pixel / per filter row, chosen so that every ds_read_b128 lane group (and the conv1
    // ds_read_b32 half-waves) is bank-conflict free: pixel stride = 2*odd 16-B slots, lane quarter q at slot q
    // (+4 slots for the second half of a 32-channel pixel) -- measured: SQ_LDS_BANK_CONFLICT/SQ_LDS_IDX_ACTIVE
    // 0.46-0.50 with the naive CINP+4 stride.
    static constexpr int S = (CINP == 4) ? 6 : (CINP == 16 ? 20 : 40);   // (CINP 16: the conflict-free 24 costs a workgroup per CU and measured slower)
    static constexpr int PH = TH + 2, PW = TW + 2;
    static constexpr int NPIX = NIMG * PH * PW;
    static constexpr int IN_FLOATS = NPIX * S;
    static constexpr int WS = (CINP == 4) ? 38 : (CINP == 16 ? 148 : 296);   // LDS floats per output channel of the filter bank
    static constexpr int W_FLOATS = ((COUT * WS + 3) / 4) * 4;
    static constexpr int NMT = NIMG * TH * TW / 16;             // 16-pixel M tiles per work item
    static constexpr int MT = NMT / 4;                          // per wave
    static constexpr int NB = COUT / 16;
    static constexpr int TPI_X = HW / TW, TPI = (HW / TH) * (HW / TW);
    static constexpr size_t LDS_BYTES = (size_t)(IN_FLOATS + W_FLOATS) * 4;
    static_assert(NMT % 4 == 0, "M tiles must split over 4 waves");
    static_assert(NIMG == 1 || (TH == HW && TW == HW), "multi-image work items hold whole images");
};

// Input-tile staging is split in two so that the HBM latency of tile i+1 hides under the MFMAs of tile i
// (issue-early / write-late): tile_load puts the next tile's global loads in flight into registers before the
// compute phase, tile_store converts (uint8 -> fp32 table / ReLU) and writes them to LDS after it.
__device__ __forceinline__ unsigned short f2bf(float x) { __bf16 h = (__bf16)x; return __builtin_bit_cast(unsigned short, h); }
__device__ __forceinline__ float bf2f(unsigned short h) { return __uint_as_float(((unsigned)h) << 16); }
template <bool BF> __device__ __forceinline__ float ld_act(const void* p, long long o) {
    if constexpr (BF) return bf2f(((const unsigned short*)p)[o]); else return ((const float*)p)[o];
}
template <bool BF> __device__ __forceinline__ void st_act(void* p, long long o, float v) {
    if constexpr (BF) ((unsigned short*)p)[o] = f2bf(v); else ((float*)p)[o] = v;
}

template <class C>
struct TileRegs {
    // fp32: 16-byte chunks of 4 channels;  bf16: 16-byte chunks of 8 channels;
    // uint8: the tile's full-width rows as dwords (HW*3/4 per row)
    static constexpr int C4 = C::BFIO ? C::CINP / 8 : C::CINP / 4;      // 16-byte chunks per pixel
    static constexpr int ROW_DW = C::HW * 3 / 4;
    static constexpr int N = C::IN_U8 ? (C::NIMG * C::PH * ROW_DW + 255) / 256 : (C::NPIX * C4 + 255) / 256;
    f32x4 v[C::IN_U8 ? 1 : N];
    uint32_t w[C::IN_U8 ? N : 1];
};

template <class C>
__device__ __forceinline__ void tile_load(TileRegs<C>& r, const void* in, const int32_t* idx, long long in_base,
                                          int n_img, int img0, int ty0, int tx0) {
    const int tid = threadIdx.x;
    if constexpr (C::IN_U8) {
        static_assert(!C::IN_U8 || C::TW == C::HW, "uint8 tiles span the full frame width");
        constexpr int RD = TileRegs<C>::ROW_DW;
#pragma unroll
        for (int k = 0; k < TileRegs<C>::N; ++k) {
            const int e = tid + k * 256;
            uint32_t v = 0u;
            if (e < C::NIMG * C::PH * RD) {
                const int row = e / RD, dw = e % RD, img = row / C::PH, gy = ty0 + (row % C::PH) - 1, n = img0 + img;
                if (n < n_img && gy >= 0 && gy < C::HW) {
                    const long long frame = idx ? (long long)idx[n] : in_base + n;
                    v = *(const uint32_t*)((const uint8_t*)in + frame * (C::HW * C::HW * 3) + gy * (C::HW * 3) + dw * 4);
                }
            }
            r.w[k] = v;
        }
    } else {
        constexpr int C4 = TileRegs<C>::C4;
#pragma unroll
        for (int k = 0; k < TileRegs<C>::N; ++k) {
            const int e = tid + k * 256;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (e < C::NPIX * C4) {
                const int pix = e / C4, c4 = e % C4;
                const int img = pix / (C::PH * C::PW), q = pix % (C::PH * C::PW);
                const int gy = ty0 + q / C::PW - 1, gx = tx0 + q % C::PW - 1, n = img0 + img;
                if (n < n_img && gy >= 0 && gy < C::HW && gx >= 0 && gx < C::HW)
                    v = *(const f32x4*)((const char*)in + ((((long long)n * C::HW + gy) * C::HW + gx) * C::CIN * (C::BFIO ? 2 : 4) + c4 * 16));
            }
            r.v[k] = v;
        }
    }
}

template <class C>
__device__ __forceinline__ void tile_store(const TileRegs<C>& r, float* s_in, const float* lut, int relu_in) {
    const int tid = threadIdx.x;
    if constexpr (C::IN_U8) {
        constexpr int RD = TileRegs<C>::ROW_DW;
#pragma unroll
        for (int k = 0; k < TileRegs<C>::N; ++k) {
            const int e = tid + k * 256;
            if (e < C::NIMG * C::PH * RD) {
                const int row = e / RD, dw = e % RD;
                const uint32_t v = r.w[k];
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const int byte = dw * 4 + b, px = byte / 3, ch = byte % 3;
                    s_in[(row * C::PW + 1 + px) * C::S + ch] = lut[(v >> (8 * b)) & 0xffu];    // lut[0] == 0: padded rows
                }
            }
        }
    } else {
        constexpr int C4 = TileRegs<C>::C4;
#pragma unroll
        for (int k = 0; k < TileRegs<C>::N; ++k) {
            const int e = tid + k * 256;
            if (e < C::NPIX * C4) {
                if constexpr (C::BFIO) {                       // 8 bf16 -> 8 fp32 (two 4-channel chunks of the fp32 tile)
                    const uint4 u = __builtin_bit_cast(uint4, r.v[k]);
                    f32x4 lo = {__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xffff0000u)};
                    f32x4 hi = {__uint_as_float(u.z << 16), __uint_as_float(u.z & 0xffff0000u), __uint_as_float(u.w << 16), __uint_as_float(u.w & 0xffff0000u)};
                    if (relu_in) {
#pragma unroll
                        for (int c = 0; c < 4; ++c) { lo[c] = fmaxf(lo[c], 0.f); hi[c] = fmaxf(hi[c], 0.f); }
                    }
                    float* d = s_in + (e / C4) * C::S + (e % C4) * 8;
                    *(f32x4*)d = lo;
                    *(f32x4*)(d + 4) = hi;
                } else {
                    f32x4 v = r.v[k];
                    if (relu_in) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                    *(f32x4*)(s_in + (e / C4) * C::S + (e % C4) * 4) = v;
                }
            }
        }
    }
}

// uint8 tiles: the halo columns x = -1 / x = HW and the 4th (pad) channel are never written by tile_store
template <class C>
__device__ __forceinline__ void tile_clear(float* s_in) {
    if constexpr (C::IN_U8)
        for (int e = threadIdx.x; e < C::IN_FLOATS; e += 256) s_in[e] = 0.f;
}

template <class C>
__device__ __forceinline__ void work_coords(int work, int& img0, int& ty0, int& tx0) {
    if (C::NIMG > 1) { img0 = work * C::NIMG; ty0 = 0; tx0 = 0; }
    else { img0 = work / C::TPI; const int t = work % C::TPI; ty0 = (t / C::TPI_X) * C::TH; tx0 = (t % C::TPI_X) * C::TW; }
}

template <class C>
__global__ __launch_bounds__(256) void conv3x3_kernel(ConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* s_in = smem;
    float* s_w = smem + C::IN_FLOATS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 15, q = lane >> 4;

    // filter bank -> LDS as [co][tap][ci]; dgrad view: co<->ci swapped, taps mirrored
    for (int e = tid; e < C::COUT * 9 * C::CINP; e += 256) {
        const int j = e / (9 * C::CINP), r = e % (9 * C::CINP), tap = r / C::CINP, k = r % C::CINP;
        float v = 0.f;
        if (k < C::CIN) v = C::TRANSW ? a.w[(k * 9 + (8 - tap)) * C::COUT + j] : a.w[(j * 9 + tap) * C::CIN + k];
        s_w[j * C::WS + tap * C::CINP + k] = v;
    }

    const int nwork = (C::NIMG > 1) ? (a.n + C::NIMG - 1) / C::NIMG : a.n * C::TPI;
    tile_clear<C>(s_in);
    float bias_r[C::NB];
#pragma unroll
    for (int nb = 0; nb < C::NB; ++nb) bias_r[nb] = a.bias ? a.bias[nb * 16 + i] : 0.f;
    TileRegs<C> regs;
    int img0, ty0, tx0;
    if ((int)blockIdx.x < nwork) {
        work_coords<C>(blockIdx.x, img0, ty0, tx0);
        tile_load<C>(regs, a.in, a.idx, a.in_base, a.n, img0, ty0, tx0);
    }
    for (int work = blockIdx.x; work < nwork; work += gridDim.x) {
        work_coords<C>(work, img0, ty0, tx0);
        __syncthreads();                       // previous tile's LDS reads are complete
        tile_store<C>(regs, s_in, a.lut, a.relu_in);
        __syncthreads();
        if (work + (int)gridDim.x < nwork) {   // next tile's loads fly during this tile's MFMAs
            int i2, y2, x2;
            work_coords<C>(work + gridDim.x, i2, y2, x2);
            tile_load<C>(regs, a.in, a.idx, a.in_base, a.n, i2, y2, x2);
        }

        // epilogue operands (ReLU-mask source, residual) are requested BEFORE the MFMA phase so that their HBM
        // latency hides under it (issued right before use they cost a full round trip per tile)
        // (whole load blocks sit under ONE wave-uniform branch each: a per-element "load or constant" select makes
        //  hipcc branch and wait around every single load)
        float e_mask[C::MT][4][C::NB], e_res[C::MT][4][C::NB];
        long long e_off[C::MT][4];
#pragma unroll
        for (int mt = 0; mt < C::MT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int pl = (wave * C::MT + mt) * 16 + q * 4 + r, y = pl / C::TW, x = pl % C::TW;
                int n = img0 + y / C::TH;
                n = n < a.n ? n : a.n - 1;                           // tail of a multi-image work item: any valid address
                e_off[mt][r] = (((long long)n * C::HW + ty0 + (y % C::TH)) * C::HW + tx0 + x) * C::COUT + i;
            }
        if (a.mask) {
#pragma unroll
            for (int mt = 0; mt < C::MT; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int nb = 0; nb < C::NB; ++nb) e_mask[mt][r][nb] = ld_act<C::BFIO>(a.mask, e_off[mt][r] + nb * 16);
        } else {
#pragma unroll
            for (int mt = 0; mt < C::MT; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int nb = 0; nb < C::NB; ++nb) e_mask[mt][r][nb] = 1.f;
        }
        if (a.res) {
#pragma unroll
            for (int mt = 0; mt < C::MT; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int nb = 0; nb < C::NB; ++nb) e_res[mt][r][nb] = ld_act<C::BFIO>(a.res, e_off[mt][r] + nb * 16);
        } else {
#pragma unroll
            for (int mt = 0; mt < C::MT; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int nb = 0; nb < C::NB; ++nb) e_res[mt][r][nb] = 0.f;
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
            abase[mt] = (((y / C::TH) * C::PH + (y % C::TH)) * C::PW + x) * C::S + (C::CINP == 4 ? q : q * 4);
        }
        const int bbase = i * C::WS + (C::CINP == 4 ? q : q * 4);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int toff = ((tap / 3) * C::PW + (tap % 3)) * C::S;
            if constexpr (C::CINP == 4) {
                float av[C::MT], bv[C::NB];
#pragma unroll
                for (int mt = 0; mt < C::MT; ++mt) av[mt] = s_in[abase[mt] + toff];
#pragma unroll
                for (int nb = 0; nb < C::NB; ++nb) bv[nb] = s_w[bbase + nb * 16 * C::WS + tap * C::CINP];
#pragma unroll
                for (int mt = 0; mt < C::MT; ++mt)
#pragma unroll
                    for (int nb = 0; nb < C::NB; ++nb) acc[mt][nb] = MFMA16(av[mt], bv[nb], acc[mt][nb]);
            } else {
#pragma unroll
                for (int s4 = 0; s4 < C::CINP / 16; ++s4) {
                    f32x4 av[C::MT], bv[C::NB];
#pragma unroll
                    for (int mt = 0; mt < C::MT; ++mt) av[mt] = *(const f32x4*)(s_in + abase[mt] + toff + s4 * 16);
#pragma unroll
                    for (int nb = 0; nb < C::NB; ++nb)
                        bv[nb] = *(const f32x4*)(s_w + bbase + nb * 16 * C::WS + tap * C::CINP + s4 * 16);
#pragma unroll
                    for (int e = 0; e < 4; ++e)
#pragma unroll
                        for (int mt = 0; mt < C::MT; ++mt)
#pragma unroll
                            for (int nb = 0; nb < C::NB; ++nb) acc[mt][nb] = MFMA16(av[mt][e], bv[nb][e], acc[mt][nb]);
                }
            }
        }

        // epilogue: D[row = q*4 + r (pixel)][col = i (channel)]
#pragma unroll
        for (int mt = 0; mt < C::MT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int pl = (wave * C::MT + mt) * 16 + q * 4 + r, y = pl / C::TW, x = pl % C::TW;
                const int n = img0 + y / C::TH;
                if (n < a.n) {
                    const long long gp = ((long long)n * C::HW + ty0 + (y % C::TH)) * C::HW + tx0 + x;
#pragma unroll
                    for (int nb = 0; nb < C::NB; ++nb) {
                        const int co = nb * 16 + i;
                        const long long o = gp * C::COUT + co;
                        float v = acc[mt][nb][r] + bias_r[nb];
                        v = e_mask[mt][r][nb] > 0.f ? v : 0.f;
                        v += e_res[mt][r][nb];
                        st_act<C::BFIO>(a.out, o, v);
                    }
                }
            }
    }
}

// ------------------------------------------------------------------------------------------ wgrad
template <int CIN_, int CINP_, int COUT_, int HW_, int TH_, int TW_, int NIMG_, bool IN_U8_, bool BFIO_ = false>
struct WgCfg {
    static constexpr int CIN = CIN_, CINP = CINP_, COUT = COUT_, HW = HW_, TH = TH_, TW = TW_, NIMG = NIMG_;
    static constexpr bool IN_U8 = IN_U8_, TRANSW = false, BFIO = BFIO_;
    static constexpr int S = (CINP == 4) ? 4 : (CINP == 32 ? 48 : 16);   // input pixel stride in LDS
    static constexpr int SO = (COUT == 32) ? 48 : 16;                    // dOut pixel stride in LDS
    static constexpr int PH = TH + 2, PW = TW + 2;
    static constexpr int NPIX = NIMG * PH * PW;
    static constexpr int NT = NIMG * TH * TW;                            // output pixels per work item
    static constexpr int IN_FLOATS = NPIX * S, DO_FLOATS = NT * SO;
    static constexpr int NCB = COUT / 16, NIB = (CINP == 4) ? 2 : CINP / 16;
    static constexpr int WLEN = COUT * 9 * CIN;                          // slab: weights then COUT bias sums
    static constexpr int SLAB = WLEN + COUT;
    static constexpr int TPI_X = HW / TW, TPI = (HW / TH) * (HW / TW);
    static constexpr int RED_FLOATS = WLEN + 256;
    static constexpr size_t LDS_BYTES =
        (size_t)((IN_FLOATS + DO_FLOATS) > RED_FLOATS ? (IN_FLOATS + DO_FLOATS) : RED_FLOATS) * 4;
};

template <class C>
__global__ __launch_bounds__(256) void conv3x3_wgrad_kernel(WgradArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* s_in = smem;
    float* s_do = smem + C::IN_FLOATS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 15, q = lane >> 4;
    constexpr int NACC = (C::CINP == 4) ? 2 : 9 * C::NCB * C::NIB;
    f32x4 acc[NACC];
#pragma unroll
    for (int k = 0; k < NACC; ++k) acc[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float bsum = 0.f;
    const int bc = tid % C::COUT, bg = tid / C::COUT;
    constexpr int BG = 256 / C::COUT;

    // conv1 (CIN = 3): N index jj = (tap, ci) flattened, 27 real columns padded to 32
    int coloff[2] = {0, 0};
    float colval[2] = {0.f, 0.f};
    if constexpr (C::CINP == 4) {
#pragma unroll
        for (int ib = 0; ib < 2; ++ib) {
            const int jj = ib * 16 + i, tap = jj / 3, ci = jj % 3;
            if (jj < 27) { coloff[ib] = ((tap / 3) * C::PW + (tap % 3)) * C::S + ci; colval[ib] = 1.f; }
        }
    }

    const int nwork = (C::NIMG > 1) ? (a.n + C::NIMG - 1) / C::NIMG : a.n * C::TPI;
    tile_clear<C>(s_in);
    TileRegs<C> regs;
    constexpr int OC4 = C::BFIO ? C::COUT / 8 : C::COUT / 4, NDO = (C::NT * OC4 + 255) / 256;     // 16-byte chunks
    f32x4 dreg[NDO];
    auto dout_load = [&](int i0, int y0, int x0) {
#pragma unroll
        for (int k = 0; k < NDO; ++k) {
            const int e = tid + k * 256;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (e < C::NT * OC4) {
                const int pl = e / OC4, c4 = e % OC4, y = pl / C::TW, x = pl % C::TW, n = i0 + y / C::TH;
                if (n < a.n)
                    v = *(const f32x4*)((const char*)a.dout + ((((long long)n * C::HW + y0 + (y % C::TH)) * C::HW + x0 + x) * C::COUT * (C::BFIO ? 2 : 4) + c4 * 16));
            }
            dreg[k] = v;
        }
    };
    int img0, ty0, tx0;
    if ((int)blockIdx.x < nwork) {
        work_coords<C>(blockIdx.x, img0, ty0, tx0);
        tile_load<C>(regs, a.in, a.idx, a.in_base, a.n, img0, ty0, tx0);
        dout_load(img0, ty0, tx0);
    }
    for (int work = blockIdx.x; work < nwork; work += gridDim.x) {
        __syncthreads();
        tile_store<C>(regs, s_in, a.lut, a.relu_in);
#pragma unroll
        for (int k = 0; k < NDO; ++k) {
            const int e = tid + k * 256;
            if (e < C::NT * OC4) {
                if constexpr (C::BFIO) {
                    const uint4 u = __builtin_bit_cast(uint4, dreg[k]);
                    float* d = s_do + (e / OC4) * C::SO + (e % OC4) * 8;
                    *(f32x4*)d = (f32x4){__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xffff0000u)};
                    *(f32x4*)(d + 4) = (f32x4){__uint_as_float(u.z << 16), __uint_as_float(u.z & 0xffff0000u), __uint_as_float(u.w << 16), __uint_as_float(u.w & 0xffff0000u)};
                } else {
                    *(f32x4*)(s_do + (e / OC4) * C::SO + (e % OC4) * 4) = dreg[k];
                }
            }
        }
        __syncthreads();
        if (work + (int)gridDim.x < nwork) {
            work_coords<C>(work + gridDim.x, img0, ty0, tx0);
            tile_load<C>(regs, a.in, a.idx, a.in_base, a.n, img0, ty0, tx0);
            dout_load(img0, ty0, tx0);
        }

#pragma unroll 2
        for (int t = wave; t < C::NT / 4; t += 4) {
            const int pl = 4 * t + q, y = pl / C::TW, x = pl % C::TW;
            const int inb = (((y / C::TH) * C::PH + (y % C::TH)) * C::PW + x) * C::S;
            float av[C::NCB];
#pragma unroll
            for (int cb = 0; cb < C::NCB; ++cb) av[cb] = s_do[pl * C::SO + cb * 16 + i];
            if constexpr (C::CINP == 4) {
#pragma unroll
                for (int ib = 0; ib < 2; ++ib) {
                    const float bv = s_in[inb + coloff[ib]] * colval[ib];
                    acc[ib] = MFMA16(av[0], bv, acc[ib]);
                }
            } else {
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) {
                    const int toff = ((tap / 3) * C::PW + (tap % 3)) * C::S;
#pragma unroll
                    for (int ib = 0; ib < C::NIB; ++ib) {
                        const float bv = s_in[inb + toff + ib * 16 + i];
#pragma unroll
                        for (int cb = 0; cb < C::NCB; ++cb)
                            acc[(tap * C::NCB + cb) * C::NIB + ib] = MFMA16(av[cb], bv, acc[(tap * C::NCB + cb) * C::NIB + ib]);
                    }
                }
            }
        }
        for (int p = bg; p < C::NT; p += BG) bsum += s_do[p * C::SO + bc];
    }

    // fixed-order reduction over the four waves, then one slab per workgroup
    __syncthreads();
    float* red = smem;
    float* redb = smem + C::WLEN;
    for (int w = 0; w < 4; ++w) {
        if (wave == w) {
            if constexpr (C::CINP == 4) {
#pragma unroll
                for (int ib = 0; ib < 2; ++ib)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int jj = ib * 16 + i;
                        if (jj < 27) {
                            const int o = (q * 4 + r) * 27 + jj;
                            red[o] = (w == 0) ? acc[ib][r] : red[o] + acc[ib][r];
                        }
                    }
            } else {
#pragma unroll
                for (int tap = 0; tap < 9; ++tap)
#pragma unroll
                    for (int cb = 0; cb < C::NCB; ++cb)
#pragma unroll
                        for (int ib = 0; ib < C::NIB; ++ib)
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const int o = ((cb * 16 + q * 4 + r) * 9 + tap) * C::CIN + ib * 16 + i;
                                const float v = acc[(tap * C::NCB + cb) * C::NIB + ib][r];
                                red[o] = (w == 0) ? v : red[o] + v;
                            }
            }
        }
        __syncthreads();
    }
    redb[tid] = bsum;
    __syncthreads();
    float* slab = a.partial + (long long)blockIdx.x * C::SLAB;
    for (int e = tid; e < C::WLEN; e += 256) slab[e] = red[e];
    if (tid < C::COUT) {
        float s = 0.f;
        for (int g = 0; g < BG; ++g) s += redb[g * C::COUT + tid];
        slab[C::WLEN + tid] = s;
    }
}

// ------------------------------------------------------------------------------------------ launchers
//                         CIN CINP COUT HW TH TW NIMG  u8     transW
using F_3_16_64   = FwdCfg< 3,  4, 16, 64, 8, 64, 1, true,  false>;
using F_16_16_32  = FwdCfg<16, 16, 16, 32, 8, 32, 1, false, false>;
using F_16_32_32  = FwdCfg<16, 16, 32, 32, 8, 32, 1, false, false>;
using F_32_32_16  = FwdCfg<32, 32, 32, 16, 8, 16, 1, false, false>;
using F_32_32_8   = FwdCfg<32, 32, 32,  8, 8,  8, 2, false, false>;
// dgrad: pass input channels = forward COUT, pass output channels = forward CIN
using D_16_16_32  = FwdCfg<16, 16, 16, 32, 8, 32, 1, false, true>;
using D_16_32_32  = FwdCfg<32, 32, 16, 32, 8, 32, 1, false, true>;
using D_32_32_16  = FwdCfg<32, 32, 32, 16, 8, 16, 1, false, true>;
using D_32_32_8   = FwdCfg<32, 32, 32,  8, 8,  8, 2, false, true>;
using W_3_16_64   = WgCfg< 3,  4, 16, 64, 8, 64, 1, true>;
using W_16_16_32  = WgCfg<16, 16, 16, 32, 8, 32, 1, false>;
using W_16_32_32  = WgCfg<16, 16, 32, 32, 8, 32, 1, false>;
using W_32_32_16  = WgCfg<32, 32, 32, 16, 8, 16, 1, false>;
using W_32_32_8   = WgCfg<32, 32, 32,  8, 8,  8, 2, false>;
// bf16 activation storage, fp32 arithmetic: block1.conv forward (uint8 in, bf16 out) and every weight gradient
using FB_3_16_64  = FwdCfg< 3,  4, 16, 64, 8, 64, 1, true,  false, true>;
using WB_3_16_64  = WgCfg< 3,  4, 16, 64, 8, 64, 1, true,  true>;
using WB_16_16_32 = WgCfg<16, 16, 16, 32, 8, 32, 1, false, true>;
using WB_16_32_32 = WgCfg<16, 16, 32, 32, 8, 32, 1, false, true>;
using WB_32_32_16 = WgCfg<32, 32, 32, 16, 8, 16, 1, false, true>;
using WB_32_32_8  = WgCfg<32, 32, 32,  8, 8,  8, 2, false, true>;

// persistent grids: as many workgroups per CU as the LDS footprint admits (max 4), on a 256-CU part
template <class C>
static int max_blocks() { int b = (int)((160 * 1024) / C::LDS_BYTES); b = b < 1 ? 1 : (b > 4 ? 4 : b); return 256 * b; }

template <class C>
static int work_items(int n) { return (C::NIMG > 1) ? (n + C::NIMG - 1) / C::NIMG : n * C::TPI; }

template <class C>
static void launch_fwd_t(const ConvArgs& a, hipStream_t st) {
    static std::once_flag attr;          // (launchers run on up to 4 group worker threads)
    std::call_once(attr, [] { hipFuncSetAttribute((const void*)conv3x3_kernel<C>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::LDS_BYTES); });
    int grid = work_items<C>(a.n);
    if (grid > max_blocks<C>()) grid = max_blocks<C>();
    if (grid < 1) return;
    hipLaunchKernelGGL(conv3x3_kernel<C>, dim3(grid), dim3(256), C::LDS_BYTES, st, a);
}

template <class C>
static void launch_wg_t(const WgradArgs& a, hipStream_t st) {
    static std::once_flag attr;          // (launchers run on up to 4 group worker threads)
    std::call_once(attr, [] { hipFuncSetAttribute((const void*)conv3x3_wgrad_kernel<C>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::LDS_BYTES); });
    int grid = work_items<C>(a.n);
    if (grid > max_blocks<C>()) grid = max_blocks<C>();
    if (grid < 1) return;
    hipLaunchKernelGGL(conv3x3_wgrad_kernel<C>, dim3(grid), dim3(256), C::LDS_BYTES, st, a);
}

void conv_shape_dims(ConvShape s, int* cin, int* cout, int* hw) {
    static const int d[CS_COUNT][3] = {{3, 16, 64}, {16, 16, 32}, {16, 32, 32}, {32, 32, 16}, {32, 32, 8}};
    *cin = d[s][0]; *cout = d[s][1]; *hw = d[s][2];
}

void launch_conv_fwd(ConvShape s, const ConvArgs& a, hipStream_t st) {
    if (a.bf16) {
        if (s == CS_3_16_64) launch_conv1_fwd_bf16(a, a.lut16, st);       // lut16 is always set in bf16 mode
        else launch_conv_fwd_bf16(s, a, st);
        return;
    }
    switch (s) {
        case CS_3_16_64:  launch_fwd_t<F_3_16_64>(a, st); break;
        case CS_16_16_32: launch_fwd_t<F_16_16_32>(a, st); break;
        case CS_16_32_32: launch_fwd_t<F_16_32_32>(a, st); break;
        case CS_32_32_16: launch_fwd_t<F_32_32_16>(a, st); break;
        case CS_32_32_8:  launch_fwd_t<F_32_32_8>(a, st); break;
        default: break;
    }
}

void launch_conv_dgrad(ConvShape s, const ConvArgs& a, hipStream_t st) {
    if (a.bf16) { launch_conv_dgrad_bf16(s, a, st); return; }
    switch (s) {
        case CS_16_16_32: launch_fwd_t<D_16_16_32>(a, st); break;
        case CS_16_32_32: launch_fwd_t<D_16_32_32>(a, st); break;
        case CS_32_32_16: launch_fwd_t<D_32_32_16>(a, st); break;
        case CS_32_32_8:  launch_fwd_t<D_32_32_8>(a, st); break;
        default: break;   // block1.conv has no data gradient (input frames)
    }
}

template <class C>
static int wg_grid_t(int n) { const int w = work_items<C>(n), m = max_blocks<C>(); return w > m ? m : w; }
int wgrad_grid(ConvShape s, int n) {
    switch (s) {
        case CS_3_16_64:  return wg_grid_t<W_3_16_64>(n);
        case CS_16_16_32: return wg_grid_t<W_16_16_32>(n);
        case CS_16_32_32: return wg_grid_t<W_16_32_32>(n);
        case CS_32_32_16: return wg_grid_t<W_32_32_16>(n);
        case CS_32_32_8:  return wg_grid_t<W_32_32_8>(n);
        default: return 0;
    }
}

int wgrad_grid_for(ConvShape s, int n, int bf16) {
    if (bf16) { const int g = wgrad_grid_bf16(s, n); if (g >= 0) return g; return wg_grid_t<WB_3_16_64>(n); }   // (conv1 without lut16: see launch)
    return wgrad_grid(s, n);
}

void launch_conv_wgrad(ConvShape s, const WgradArgs& a, hipStream_t st) {
    if (a.bf16) {
        if (s == CS_3_16_64) { launch_conv1_wgrad_bf16(a, a.lut16, st); return; }
        if (wgrad_grid_bf16(s, a.n) >= 0) { launch_conv_wgrad_bf16(s, a, st); return; }
        switch (s) {
            case CS_3_16_64:  launch_wg_t<WB_3_16_64>(a, st); break;
            case CS_16_16_32: launch_wg_t<WB_16_16_32>(a, st); break;
            case CS_16_32_32: launch_wg_t<WB_16_32_32>(a, st); break;
            case CS_32_32_16: launch_wg_t<WB_32_32_16>(a, st); break;
            case CS_32_32_8:  launch_wg_t<WB_32_32_8>(a, st); break;
            default: break;
        }
        return;
    }
    switch (s) {
        case CS_3_16_64:  launch_wg_t<W_3_16_64>(a, st); break;
        case CS_16_16_32: launch_wg_t<W_16_16_32>(a, st); break;
        case CS_16_32_32: launch_wg_t<W_16_32_32>(a, st); break;
        case CS_32_32_16: launch_wg_t<W_32_32_16>(a, st); break;
        case CS_32_32_8:  launch_wg_t<W_32_32_8>(a, st); break;
        default: break;
    }
}
