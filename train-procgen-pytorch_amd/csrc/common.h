// Shared declarations for the MI355X PPO hot-path library (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define MI_WAVE 64

// ---------------------------------------------------------------- conv3x3 (NHWC, stride 1, pad 1)
// One descriptor serves forward and data-gradient launches (dgrad = forward form over dOut with
// the tap-flipped, channel-transposed weight view; see conv.hip).
#if defined(__HIPCC__)
// two fp32 -> one dword of two bf16 (lo in bits 0..15), round to nearest even: ONE v_cvt_pk_bf16_f32 (converting the halves
// separately costs two converts + and + shift + or)
typedef float mi_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 mi_bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned mi_pk_bf16(float lo, float hi) {
    return __builtin_bit_cast(unsigned, __builtin_convertvector((mi_f32x2){lo, hi}, mi_bf16x2));
}

// ---- MaxPool2d(3,2,1) on order-preserving integer keys (the fused conv+pool kernels are VALU-issue bound, and a float
// compare + two selects per (window position, channel) was most of their instruction count).
// key16(v): signed-int16 order == float order of the bf16 value, -0 and +0 share a key.  The conv epilogue stores keys, the
// pooling widens a key to the high half of a dword and puts (8 - window position) in the low bits: ONE v_max3_i32 per
// three positions then yields the maximum AND its first row-major position (ties: larger low bits = earlier position --
// the `v > best` rule of torch's max_pool2d and of the stand-alone kernel in misc.hip).  A positive NaN sorts above +inf
// and so propagates; cells outside the image hold MI_KEY_MIN.
typedef short mi_s16x2 __attribute__((ext_vector_type(2)));
constexpr unsigned MI_KEY_MIN2 = 0x80008000u;                 // two minimal keys
__device__ __forceinline__ unsigned mi_bf16x2_to_keys(unsigned w) {
    const mi_s16x2 v = __builtin_bit_cast(mi_s16x2, w), s = v >> (mi_s16x2){15, 15};          // 0 / -1 per half
    return __builtin_bit_cast(unsigned, (mi_s16x2)((v ^ (s & (mi_s16x2){0x7fff, 0x7fff})) - s));
}
__device__ __forceinline__ unsigned mi_keys_to_bf16x2(unsigned kk) {
    mi_s16x2 k = __builtin_bit_cast(mi_s16x2, kk);
    const mi_s16x2 s = k >> (mi_s16x2){15, 15};
    k = k + s;
    return __builtin_bit_cast(unsigned, (mi_s16x2)(k ^ (s & (mi_s16x2){0x7fff, 0x7fff})));
}
__device__ __forceinline__ int mi_max3i(int a, int b, int c) { return max(max(a, b), c); }
// u[p] = the 8 channel keys (4 dwords) of window position p = ky*3 + kx; val = 8 pooled bf16, arg = 8 position bytes
__device__ __forceinline__ void mi_pool9_keys(const uint4 (&u)[9], uint4& val, uint2& arg) {
    int best[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        int k[9];
#pragma unroll
        for (int p = 0; p < 9; ++p) {
            const unsigned w = (q >> 1) == 0 ? u[p].x : (q >> 1) == 1 ? u[p].y : (q >> 1) == 2 ? u[p].z : u[p].w;
            k[p] = (int)((q & 1) ? ((w & 0xffff0000u) | (unsigned)(8 - p)) : ((w << 16) | (unsigned)(8 - p)));
        }
        best[q] = mi_max3i(mi_max3i(k[0], k[1], k[2]), mi_max3i(k[3], k[4], k[5]), mi_max3i(k[6], k[7], k[8]));
    }
    unsigned kv[4], lo[2];
#pragma unroll
    for (int j = 0; j < 4; ++j) kv[j] = ((unsigned)best[2 * j] >> 16) | ((unsigned)best[2 * j + 1] & 0xffff0000u);
#pragma unroll
    for (int j = 0; j < 2; ++j)        // low bytes of four keys = 8 - position
        lo[j] = __builtin_amdgcn_perm(__builtin_amdgcn_perm((unsigned)best[4 * j + 3], (unsigned)best[4 * j + 2], 0x0c0c0400u),
                                      __builtin_amdgcn_perm((unsigned)best[4 * j + 1], (unsigned)best[4 * j], 0x0c0c0400u), 0x05040100u);
    val = (uint4){mi_keys_to_bf16x2(kv[0]), mi_keys_to_bf16x2(kv[1]), mi_keys_to_bf16x2(kv[2]), mi_keys_to_bf16x2(kv[3])};
    arg = (uint2){0x08080808u - lo[0], 0x08080808u - lo[1]};
}
#endif

struct ConvArgs {
    const void*    in;        // fp32 NHWC [n][HW][HW][CIN]  (or uint8 frames NHWC when the shape is the u8 one)
    const int32_t* idx;       // u8 input only: sample s reads frame idx[s] (minibatch gather); null -> in_base + s
    long long      in_base;   // u8 input only: first frame when idx == null (rollout step t -> t*E)
    const float*   w;         // device weight layout [CO][9][CI] (tap = ky*3+kx), CO/CI of the FORWARD conv
    const float*   bias;      // [COUT] or null
    const void*    res;       // optional residual, same shape / type as out: out += res
    const void*    mask;      // optional ReLU mask source, same shape / type as out: out *= (mask > 0)
    void*          out;       // NHWC [n][HW][HW][COUT], fp32 or bf16
    const float*   lut;       // 256-entry u8 -> fp32 table (k/255 correctly rounded)
    int            n;         // images
    int            relu_in;   // apply max(x,0) while staging the input tile
    int            bf16;      // activations (in/res/mask/out) are bf16 in HBM
    const unsigned short* wbank;   // bf16 mode: pre-packed filter bank in the kernel's LDS layout (conv_bf16.hip), or null
    const unsigned short* lut16;   // bf16 mode: 256-entry uint8 -> bf16 table (block1.conv)
    const uint8_t* pool_arg;       // bf16 data gradient of a block's first conv: `in` is the POOLED gradient [n][HW/2][HW/2][C] and these
                                   // are the max-pool arg-max bytes; the conv-output gradient is rebuilt in LDS (pool backward fused)
    const void*    wg_in;          // with pool_arg (block2.conv): the conv's FORWARD input (bf16 NHWC) -- the kernel then also accumulates
    float*         wg_partial;     // the weight / bias gradient from the same LDS tile: [grid][cout_f*9*cin_f + cout_f] slabs
};

struct WgradArgs {
    const void*    in;        // forward input of the conv (fp32 NHWC or u8 frames)
    const int32_t* idx;
    long long      in_base;
    const void*    dout;      // NHWC [n][HW][HW][COUT], fp32 or bf16
    float*         partial;   // [gridDim.x][COUT*9*CIN + COUT] per-workgroup slabs (weights then bias)
    const float*   lut;
    int            n;
    int            relu_in;
    int            bf16;      // in (unless uint8 frames) and dout are bf16 in HBM
    const unsigned short* lut16;   // bf16 mode: uint8 -> bf16 table (block1.conv)
    const uint8_t* pool_arg;       // block1.conv, bf16: dout is the POOLED gradient (32x32x16) + these arg-max bytes
};

enum ConvShape {              // (CIN, COUT, HW) of the FORWARD conv
    CS_3_16_64 = 0,           // block1.conv   (uint8 frames in)
    CS_16_16_32,              // block1.res*.conv*
    CS_16_32_32,              // block2.conv
    CS_32_32_16,              // block2.res*.conv*, block3.conv
    CS_32_32_8,               // block3.res*.conv*
    CS_COUNT
};

// launchers (conv.hip).  grid_blocks <= 0 -> library default.
void launch_conv_fwd(ConvShape s, const ConvArgs& a, hipStream_t st);
void launch_conv_dgrad(ConvShape s, const ConvArgs& a, hipStream_t st);   // a.in = dOut, a.out = dIn
int  wgrad_grid(ConvShape s, int n);                                       // workgroups a wgrad launch will use
void launch_conv_wgrad(ConvShape s, const WgradArgs& a, hipStream_t st);
void conv_shape_dims(ConvShape s, int* cin, int* cout, int* hw);
void launch_conv_fwd_bf16(ConvShape s, const ConvArgs& a, hipStream_t st);     // conv_bf16.hip (bf16 MFMA)
void launch_conv_dgrad_bf16(ConvShape s, const ConvArgs& a, hipStream_t st);
void launch_conv1_fwd_bf16(const ConvArgs& a, const unsigned short* lut16, hipStream_t st);
void launch_pack_conv1_bank(const float* w, unsigned short* bank, hipStream_t st);   // ConvArgs.wbank of launch_conv1_pool_fwd_bf16 (optional)
int  conv1_bank_elems();
void launch_conv1_pool_fwd_bf16(const ConvArgs& a, const unsigned short* lut16, void* p_out, uint8_t* p_arg, hipStream_t st);
void launch_conv1_wgrad_bf16(const WgradArgs& a, const unsigned short* lut16, hipStream_t st);
int  wgrad_grid_bf16(ConvShape s, int n);                                  // -1: shape handled by conv.hip
void launch_conv_wgrad_bf16(ConvShape s, const WgradArgs& a, hipStream_t st);
int  wgrad_grid_for(ConvShape s, int n, int bf16);                          // slabs a wgrad launch writes

// ---------------------------------------------------------------- generic MFMA GEMM (linear layers)
// C[M][N] (op)= sum_k A(m,k) * B(k,n);  A(m,k) = A[m*sam + k*sak], B(k,n) = B[k*sbk + n*sbn].
struct GemmArgs {
    const float* A; const float* B; float* C;
    int M, N, K;
    long long sam, sak, sbk, sbn, ldc;
    const float* bias;     // per-n bias or null
    const float* mask;     // C *= (mask[m*ldc+n] > 0) or null
    int relu_a, relu_b;    // max(x,0) on operand load
    int relu_out;          // max(c,0) in the epilogue (after bias)
    int accumulate;        // C += result
    int a_bf16, b_bf16, mask_bf16, c_bf16;   // the operand / mask / output is bf16 in HBM (arithmetic stays fp32)
    float* ws; size_t ws_floats;             // split-K slab workspace of the calling context (null: never split)
};
void launch_gemm(const GemmArgs& g, hipStream_t st);

// ---------------------------------------------------------------- small kernels (misc.hip)
void launch_maxpool_fwd(const float* in, float* out, uint8_t* arg, int n, int hw, int c, hipStream_t st);
void launch_maxpool_bwd(const float* dout, const uint8_t* arg, float* din, int n, int hw, int c, hipStream_t st);
void launch_maxpool_fwd_bf16(const void* in, void* out, uint8_t* arg, int n, int hw, int c, hipStream_t st);
void launch_maxpool_bwd_bf16(const void* dout, const uint8_t* arg, void* din, int n, int hw, int c, hipStream_t st);
void launch_heads_bwd(const float* dY, const float* feat, const float* Wh, int relu_mask, float* dfeat, float* gW, float* gb, float* ws /* >= 256 * (O*H + O) floats */,
                      int n, int H, int O, hipStream_t st, bool with_reduce = true, hipEvent_t done_ev = nullptr);
void launch_heads_bwd_reduce(const float* ws, float* gW, float* gb, int n, int H, int O, hipStream_t st);      // heads' data + weight + bias gradient in one launch (+ the slab sum); H <= 256, O <= 16
void launch_reduce_slabs(const float* partial, int nslab, int slab_len, float* dst_w, int n_w, float* dst_b, int n_b,
                         hipStream_t st);
void launch_colsum_acc(const float* dY, int M, int N, int ld, float* db, float* col_ws /* >= 64 x 4096 floats */, hipStream_t st);
void launch_gather_rows(const float* src, const int32_t* idx, long long base, float* dst, int n, int d, hipStream_t st);

struct LossHP { float eps_clip, value_coef, entropy_coef, x_entropy_coef, entropy_mult, fs_coef; };
// per-sample scalars are gathered through idx (flat index t*E+e) from the (T,E) rollout arrays
struct LossArgs {
    const float* hout;        // [n][A+1] logits then value
    const int32_t* idx;       // [n]
    const int32_t* act; const float* old_logp; const float* old_value; const float* ret; const float* adv;
    float* dY;                // [n][A+1] gradient wrt logits/value (loss_bwd)
    float* partial;           // [nblk][8+A]
    float* stats;             // [16+A] finalised: 0 pi_loss 1 value_loss 2 entropy 3 x_ent 4 total 5 fs 6 marg 8.. q[A]
    int n, A;
    float inv_n_global;       // 1 / (global minibatch size)
    LossHP hp;
};
int  loss_blocks(int n);
void launch_loss_fwd(const LossArgs& a, hipStream_t st);
// phase bit 0: block partials -> this rank's share of the global means; bit 1: derived terms + log record
void launch_loss_finalize(const LossArgs& a, int nblk, int phase, const float* fs_ptr, float* log_slot, hipStream_t st);
void launch_logp_all(const float* hout, int n, int A, float* lp_out, float* value_out, hipStream_t st);
void launch_loss_bwd(const LossArgs& a, hipStream_t st);
// several global minibatches in one gathered batch (mi_minibatch_multi): samples of segment k are [start[k], start[k+1])
constexpr int MI_MAX_SEG = 16;
constexpr int FS_PARTS = 8;           // column blocks of the feature-sparsity metric's second stage
struct SegTab { int n_seg; int start[MI_MAX_SEG + 1]; };
int  loss_blocks_seg(const SegTab& st);
// with_bwd: also writes dY (the loss_bwd pass) -- only for a loss without batch-level terms (x_entropy_coef == 0)
void launch_loss_fwd_seg(const LossArgs& a, const SegTab& st, bool with_bwd, hipStream_t stream);
// one workgroup per segment: its block partials -> stats_base + 32 k (phase bit 0), derived terms + record log_base + 8 k (bit 1).
// fs_parts (from launch_fs_metric_seg, or null): the segment's feature-sparsity metric is finished here and left in fs_out[k]
void launch_loss_finalize_seg(const LossArgs& a, const SegTab& st, int phase, float* stats_base, const double* fs_parts, int fs_d, float* fs_out,
                              float* log_base, hipStream_t stream);
void launch_loss_finalize_records(const LossArgs& a, int n_rec, float* stats_base, const float* fs_base, float* log_base, hipStream_t stream);
// colmax_scratch: 512 x d floats; fs_parts: n_seg x 8 fp64 column-block sums of tanh(100 max_b relu(h))
void launch_fs_metric_seg(const void* flat_pre, int bf16, const SegTab& st, int d, float* colmax_scratch, double* fs_parts, hipStream_t stream);
int  fs_groups_per_segment(int n_seg);
void launch_fs_grad(const void* x, int bf16, int n, int d, const float* part, int G, void* Gd, float fs_coef, float* colmax, int* arg, hipStream_t st);
// multi-rank feature-sparsity term: per-column 64-bit candidates for ONE max-all-reduce, then the winner applies (misc.hip)
void launch_fs_keys(const void* x, int bf16, int n, int d, const float* part, int G, const int32_t* gpos, float* colmax, int* arg,
                    long long* keys, long long* keys_local, hipStream_t st);
void launch_fs_apply_keys(void* Gd, int bf16, int d, const long long* keys, const long long* keys_local, const int* arg, float fs_coef, hipStream_t st);
void launch_fs_from_keys(const long long* keys, int d, float* fs_out, hipStream_t st);
int  fs_metric_groups();
void launch_fs_metric(const void* flat_pre, int bf16, int n, int d, float* colmax_scratch, float* fs_out, hipStream_t st);

void launch_gae(const float* rew, const float* done, const float* value, float* adv, float* ret, int T, int E,
                float gamma, float lmbda, int use_gae, hipStream_t st);
void launch_advnorm_stats(const float* adv, int n, double* stats3, hipStream_t st);   // stats3 = {count, mean, M2}
void launch_advnorm_apply(float* adv, int n, const double* stats3, hipStream_t st);
void launch_advnorm_merge(const double* all, int R, double* stats3, hipStream_t st);   // all: R x {count, mean, M2}

void launch_sample(const float* hout, int n, int A, const float* u, unsigned long long seed, unsigned long long ctr,
                   int32_t* act, float* logp, float* value, hipStream_t st);

void launch_philox_debug(const uint32_t* in6, int n, uint32_t* out4, float* u_out, hipStream_t st);   // test hook (mi_debug_philox)

void launch_heads_sample(const float* feat, const float* Wh, const float* bh, int n, int H, int A, const float* u,
                         unsigned long long seed, unsigned long long ctr, int32_t* act, float* logp, float* value, float* pack,
                         float* hout, const float* rd, float* rew_dst, float* done_dst, hipStream_t st,
                         unsigned* done_ctr = nullptr, unsigned* host_flag = nullptr, unsigned ticket = 0);
void launch_sumsq(const float* g, long long n, double* out, double* part, hipStream_t st);          // out[0] = sum g^2 (deterministic)
void launch_sumsq_partials(const float* g, long long n, double* part /* 128 doubles */, hipStream_t st);
void launch_adam(float* p, float* g, float* m, float* v, long long n, const double* sumsq /* the sum, or npart partial sums */, int npart, float max_norm, float lr,
                 float beta1, float beta2, float eps, float step_size_scale, float bc2_sqrt, float* gnorm_out,
                 hipStream_t st);
void launch_fill(float* p, long long n, float v, hipStream_t st);
void launch_value_seed(float* dY, int n, int A, hipStream_t st);
void launch_conv1_input_grad(const void* dC, int bf16, const float* W, float* dX, int n, hipStream_t st);
void launch_mask_rows(const float* h, const float* done, float* out, int n, int H, hipStream_t st);   // out = h * (1 - done[row])
void launch_gru_value_bwd(const float* gi, const float* gh, const float* hm, const float* wv, float* dgates, int n, int H, hipStream_t st);
void launch_gru_gates(const float* gi, const float* gh, const float* hm, float* h_out, float* feat_out, int n, int H, hipStream_t st);

// ---------------------------------------------------------------- embedder.fc on the bf16 matrix cores (fc_bf16.hip)
struct FcNtArgs;
void launch_fc_fwd_small_bf16(const void* X, const unsigned short* Wp, const float* bias, float* feat, int n, hipStream_t st);
void launch_fc_pack(const float* w, unsigned short* wp, unsigned short* wt, int N, int K, hipStream_t st);
void launch_fc_fwd_bf16(const void* x_bf16, const unsigned short* wp, const float* bias, float* y, int n, hipStream_t st);
void launch_fc_dgrad_bf16(const float* dy, const unsigned short* wt, const void* mask_bf16, void* dx_bf16, int n, hipStream_t st);
void launch_fc_tn(const float* A, const unsigned short* B, float* gW, float* ws, size_t ws_floats, int M, int N, int K, hipStream_t st);

// blocks 2 + 3 of a rollout-sized batch (n <= 256) in one launch (rollout_bf16.hip): bank / bias = their ten convs in network order
void launch_rollout_tail_bf16(const void* x, void* y, int n, const unsigned short* const* bank, const float* const* bias, hipStream_t st);
// fused residual block forward, bf16 mode (resblock_bf16.hip); s = ConvShape of the block's convs
bool launch_conv_pool_fwd_bf16(ConvShape s, const ConvArgs& a, void* p_out, uint8_t* p_arg, hipStream_t st);
void launch_resblock_bf16(ConvShape s, const void* x, const float* b1, const float* b2, void* a_out, void* y_out, int n,
                          const unsigned short* bank1, const unsigned short* bank2, hipStream_t st);
int  resblock_bwd_full_grid(int n);
void launch_resblock_bwd_full_bf16(const void* dy, const void* a_fwd, const void* x_fwd, void* dx_out, void* da_out, int n,
                                   const unsigned short* bank2_t, const unsigned short* bank1_t, float* slab2, float* slab1, hipStream_t st, hipEvent_t done_ev = nullptr /* set: recorded at this launch's completion; only the default (16d) kernel honours it, the caller checks launch_resblock_bwd_full_event_ok() */);
bool launch_resblock_bwd_full_event_ok();
int  resblock_bwd_full32_grid(ConvShape s, int n);
void launch_resblock_bwd_full32_bf16(ConvShape s, const void* dy, const void* a_fwd, const void* x_fwd, void* dx_out, void* da_out, int n,
                                     const unsigned short* bank2_t, const unsigned short* bank1_t, float* slab2, float* slab1, hipStream_t st);
int  conv_bwd_fused_grid(ConvShape s, int n);     // grid (= slab count) of the fused data + weight gradient launch, or -1
void launch_resblock_pair_bf16(ConvShape s, const void* x, const float* const* b, void* a1_out, void* y1_out, void* a2_out, void* y2_out, int n,
                               const unsigned short* const* bank, hipStream_t st);
void launch_resblock_bwd_bf16(ConvShape s, const void* dy, const void* a_fwd, const void* x_fwd, void* da_out, void* dx_out, int n,
                              const unsigned short* bank2_t, const unsigned short* bank1_t, hipStream_t st);
// pre-packed bf16 filter banks: [rows][WS] with WS = NK*32+16, K laid out tap-major (conv_bf16.hip); rows = output
// channels of the pass (dgrad: transposed + tap-mirrored view).  One descriptor per bank, device-resident.
// one conv layer's slabs for reduce_all_slabs: [nslab][slab_len] floats at ws + src_off -> grads[w_off..] (first n_w) and grads[b_off..]
struct SlabDesc { long long src_off, w_off, b_off; int nslab, slab_len, n_w; };
void launch_reduce_all_slabs(const float* ws, float* grads, const SlabDesc* d_desc, int n_desc, int max_slab_len, hipStream_t st);
struct BankDesc { long long w_off, out_off; int rows, cin_pass, co_f, ci_f, transw, ws, nk; };
int  bank_ws(int cin_pass);                 // elements per bank row
void launch_pack_banks(const float* params, unsigned short* banks, const BankDesc* d_desc, int n_desc, hipStream_t st);
// a launcher asked for a shape it has no kernel for: recorded here (thread-local), reported by the engine's next NETCHK as error -4
void mi_launch_fail(const char* msg);
const char* mi_launch_failed_take();      // the message (and clears it), or null
void launch_heads_fwd(const float* feat, const float* Wh, const float* bh, float* hout, int n, int O /* <= 16 outputs, H == 256 */, hipStream_t st);
void launch_pull_bytes(const void* host_src /* pinned, device-visible */, void* dst, size_t bytes /* multiple of 16 */, hipStream_t st);
void launch_pull_i32(const int32_t* host_src /* pinned, device-visible */, int32_t* dst, int n, hipStream_t st);
void launch_repack_all(const float* params, unsigned short* banks, const BankDesc* d_desc, int n_desc, const float* c1_w, unsigned short* c1_bank /* or null */,
                       const float* fc_w, unsigned short* fc_wp, unsigned short* fc_wt, hipStream_t st);      // the three re-packing launches of an IMPALA bf16 context in one
