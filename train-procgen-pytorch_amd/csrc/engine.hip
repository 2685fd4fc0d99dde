// Context, network program and the C ABI (include/mi355ppo.h) of the MI355X PPO hot path.
//
// Device-resident state per context (one per GPU):
//   rollout ring   frames uint8 NHWC [(T+1)][E][64][64][3]  (or fp32 [(T+1)][E][obs_dim] for the MLP)
//                  rew/done/logp/adv/ret fp32 [T][E], act int32 [T][E], value fp32 [T+1][E]
//   parameters     ONE flat fp32 buffer (+ grad, exp_avg, exp_avg_sq of the same shape): filter banks
//                  as [co][tap][ci], fc columns in NHWC-flatten order, heads as one (A+1) x H matrix
//   activations    NHWC fp32, every tensor the backward pass needs, sized for max_batch samples
// The reference keeps all of this on the host in fp32 and re-uploads a gathered minibatch for every
// update (common/storage.py:112-128); here the minibatch gather is an index read inside the first conv.
#include "common.h"
#include "../../include/mi355ppo.h"
#include <rccl/rccl.h>
#include <math.h>
#include <stdio.h>
#include <string.h>
#include <string>
#include <unordered_map>
#include <vector>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <thread>

static std::vector<uint16_t> host_to_bf16(const float* x, size_t n);
static thread_local std::string g_err;
// A worker thread of the pipelined rollout issues one env group's pass on that group's stream with that group's slice of the split-K
// workspace: the network program reads both through these thread-local overrides (null on every other thread: the context's own).
static thread_local hipStream_t tl_stream = nullptr;
static thread_local float* tl_ws = nullptr;
static thread_local size_t tl_ws_floats = 0;
#define CUR(c) (tl_stream ? tl_stream : (c)->stream)
const char* mi_last_error(void) { return g_err.c_str(); }
static int fail(int code, const std::string& msg) { g_err = msg; return code; }

#define HIPC(x)                                                                                          \
    do {                                                                                                 \
        hipError_t e_ = (x);                                                                             \
        if (e_ != hipSuccess)                                                                            \
            return fail(-2, std::string(#x) + ": " + hipGetErrorString(e_) + " @" + std::to_string(__LINE__)); \
    } while (0)
#define NCCLC(x)                                                                                         \
    do {                                                                                                 \
        ncclResult_t r_ = (x);                                                                           \
        if (r_ != ncclSuccess) return fail(-5, std::string(#x) + ": " + ncclGetErrorString(r_) + " @" + std::to_string(__LINE__)); \
    } while (0)
#define ARG(c, msg) do { if (!(c)) return fail(-1, std::string("invalid argument: ") + msg); } while (0)
#define NETCHK(c) do { if (const char* lf_ = mi_launch_failed_take()) return fail(-4, lf_);                                   \
                       std::string m_ = net_err_take(c); if (!m_.empty()) return fail(-4, m_); } while (0)

struct mi_ctx;
static std::string net_err_take(mi_ctx* c);                 // (the network program also runs on the env-group worker threads: guarded)
static void net_err_set(mi_ctx* c, const std::string& m);
static int join_groups(mi_ctx* c);
extern "C" int mi_comm_destroy(mi_ctx* c);
// every entry point that issues work on the context's main stream first orders it behind the env-group streams of a pipelined rollout
#define JOIN(c) do { if ((c)->groups_live) { int r_ = join_groups(c); if (r_) return r_; } } while (0)

enum TKind { K_PLAIN = 0, K_CONVW, K_FCW };
struct TensorDesc {
    std::string name;
    int64_t ref_off, dev_off, n;
    int kind, co, ci;
};

struct ConvLayer { ConvShape shape; int64_t w_off, b_off; int cin, cout, hw; long long bank_f, bank_d; };   // bank offsets (bf16 mode) or -1
struct Block { float *C, *P0, *A1, *P1, *A2, *P2; uint8_t* PI; int cin, cout, hin; };   // activation buffers hold fp32 or bf16 (ctx.bf)
struct Linear { int64_t w_off, b_off; int in, out; };

// ---- live kernel timing (bench.py roofline leg)
enum ProfClass { PC_CONV_FWD = 0, PC_CONV_DGRAD = 5, PC_CONV_WGRAD = 10, PC_POOL_FWD = 15, PC_POOL_BWD, PC_GEMM, PC_SLAB_REDUCE, PC_RESBLOCK, PC_RESBLOCK_BWD = PC_RESBLOCK + 5, PC_COUNT = PC_RESBLOCK_BWD + 5 };
static const char* kProfNames[PC_COUNT] = {
    "conv_fwd_3_16_64", "conv_fwd_16_16_32", "conv_fwd_16_32_32", "conv_fwd_32_32_16", "conv_fwd_32_32_8",
    "conv_dgrad_3_16_64(unused)", "conv_dgrad_16_16_32", "conv_dgrad_16_32_32", "conv_dgrad_32_32_16", "conv_dgrad_32_32_8",
    "conv_wgrad_3_16_64", "conv_wgrad_16_16_32", "conv_wgrad_16_32_32", "conv_wgrad_32_32_16", "conv_wgrad_32_32_8",
    "maxpool_fwd", "maxpool_bwd", "gemm", "slab_reduce",
    "resblock_fwd_(unused)", "resblock_fwd_16_16_32", "resblock_fwd_(unused)", "resblock_fwd_32_32_16", "resblock_fwd_32_32_8",
    "resblock_dgrad_(unused)", "resblock_dgrad_16_16_32", "resblock_dgrad_(unused)", "resblock_dgrad_32_32_16", "resblock_dgrad_32_32_8"};
struct ProfPending { hipEvent_t a, b; int cls, phase; long long units; double bytes, flops; };
struct Profiler {
    bool on = false;
    bool all_phases = false;       // false: update phase only (the rollout's ~5.6k tiny launches per iteration are not bracketed)
    int phase = 0;
    int period = 1, mb_count = 0;  // update phase: bracket every period-th minibatch (two event records per launch cost ~7 us of
    bool sample_now = true;        // stream time: 11 ms per hard-500 iteration when every launch is bracketed)
    std::vector<ProfPending> pend;
    std::vector<hipEvent_t> pool;
    double ms[2][PC_COUNT] = {};
    long long launches[2][PC_COUNT] = {}, units[2][PC_COUNT] = {};
    double bytes[2][PC_COUNT] = {}, flops[2][PC_COUNT] = {};
};

struct mi_ctx;
static void prof_harvest(mi_ctx* c);
// one step of one env group, as the submitting thread hands it to the group's worker
struct GroupJob { int t; const void* frames; size_t bytes; bool pull; bool have_rd, last; const float* u; unsigned long long seed; unsigned ticket; };
struct GroupWorker {
    std::thread th; std::mutex mu; std::condition_variable cv;
    std::atomic<unsigned> posted{0}, done{0}; std::atomic<bool> sleeping{false}, quit{false};
    GroupJob job{}; int rc = 0; std::string err;
};

struct mi_ctx {
    Profiler prof;
    mi_config cfg;
    hipStream_t stream;
    bool own_stream;
    int T, E, A, H, NB;
    bool bf;              // IMPALA activations / activation gradients stored as bf16 (mi_config.precision == 1)
    double es;            // bytes per activation element
    int64_t n_params;
    std::vector<TensorDesc> tensors;
    float *params, *grads, *adam_m, *adam_v;
    // rollout
    uint8_t* frames;      // impala
    float* obsf;          // mlp
    size_t obs_bytes_per_env;
    float *rew, *done, *logp, *adv, *ret, *value;
    int32_t* act;
    double* adv_stats;
    // network
    std::vector<ConvLayer> convs;
    Block blk[3];
    Linear fc;            // impala fc 2048->256
    std::vector<Linear> mlp;
    std::vector<float*> mlp_act;   // X0 (input), h1..hL
    int64_t wh_off, bh_off;        // heads: (A+1) x H weights, (A+1) bias (device order)
    float *feat, *hout, *dY, *dfeat, *GC, *GP[3];
    float* slabs; size_t slab_floats;
    void* sal_dc; float* sal_dx; const float* sal_src;       // value saliency: conv-out gradient temp (bf16 mode), input gradient, where net_backward left block 1's gradient
    long long slab_off[15]; SlabDesc h_slab_desc[15]; SlabDesc* d_slab_desc; int slab_desc_n, slab_desc_cached_n;   // per-layer slab regions; ONE reduce launch per backward pass
    float *gemm_ws, *col_ws, *fs_scratch, *fs_val; size_t gemm_ws_floats;      // split-K / column-sum workspaces: per context
    float* lut;
    unsigned short* lut16;     // uint8 -> bf16(k/255) table (bf16 mode, block1.conv)
    uint8_t* stage_frames; float* stage_obs;
    int32_t* d_idx;
    float *loss_partial, *loss_stats, *loss_log; int log_count, log_cap;
    double* fs_parts;                                     // [MI_MAX_SEG][8] column-block sums of the feature-sparsity metric
    float *stats_ring, *fs_ring; LossArgs ring_args;      // multirank mode 2: per-minibatch raw stats [log_cap][32] (+ rank-local fs), finalised after ONE all-reduce
    double* sumsq; float* gnorm;
    float* d_u; float* d_lp;
    unsigned short* banks; BankDesc* d_bank_desc; int n_banks;   // bf16 mode: pre-packed conv filter banks
    unsigned short* c1_bank;                                   // bf16 mode: block1.conv forward bank (conv1 kernels' LDS layout)
    unsigned short *fc_wp, *fc_wt;            // bf16 mode: packed fc.weight images ([256][2048] and [2048][256])
    bool fc_packed_valid;
    float *d_pack, *h_pack, *h_rd, *d_rd;     // packed rollout read-back {act,logp,value} x E ; packed {rew,done} upload
    unsigned *d_done_ctr, *h_flag, roll_ticket;   // rollout step: workgroup counter, host-visible completion ticket (heads_sample_kernel)
    int32_t* s_act; float *s_logp, *s_val; bool staged_valid;
    // recurrent rollout (GRU cell, never trained)
    bool gru_on; float *gru_wih, *gru_whh, *gru_bih, *gru_bhh, *h_state, *h_masked, *gru_gi, *gru_gh, *d_done;
    float *gru_x, *gru_dg; bool sal_keep_x, bwd_from_dfeat;      // value saliency through the GRU: the cell's input (embedder output), d gates; net_backward starts at dfeat
    // pinned host staging
    // index staging ring: a slot is rewritten only after the H2D copy that read it has completed
    static constexpr int IDX_RING = 32;
    int32_t* h_idx_ring[IDX_RING]; hipEvent_t idx_ev[IDX_RING]; bool idx_used[IDX_RING]; int idx_next, idx_ev_deferred;
    float* h_f; int32_t* h_i; size_t h_f_floats;
    int multirank;
    LossArgs pending; int pending_n;
    // pipelined rollout (mi_rollout_submit / mi_rollout_wait): contiguous env groups, each on its own stream with its own rows of the
    // activation buffers, so that one group's frame upload + forward runs beside the host's wait for another group's actions
    static constexpr int MAX_GROUPS = 4;
    int n_groups; hipStream_t main_stream, gs[MAX_GROUPS]; hipEvent_t ev_fork[MAX_GROUPS], ev_join[MAX_GROUPS];
    bool g_forked[MAX_GROUPS], g_busy[MAX_GROUPS], g_last[MAX_GROUPS], g_dirty[MAX_GROUPS]; unsigned g_ticket[MAX_GROUPS]; bool groups_live;
    struct GroupWorker* gw[MAX_GROUPS];      // one host thread per group issues that group's copies + launches (a step is ~9 API calls = ~30 us of host time)
    std::atomic<int64_t> copy_slot_ns{0}; double copy_rate_bytes_per_us;      // uploads of the env groups take turns on the PCIe link (group_issue)
    std::unordered_map<const void*, bool> pull_ok; bool no_pull;      // frame buffers a kernel may read (mi_debug_flags bit 2: always DMA)
    // Side stream of a minibatch pass: the logged statistics (feature-sparsity metric, loss records) and embedder.fc's weight / bias
    // gradients are needed by nobody before the optimizer step, so they run beside the backward pass instead of in front of it
    // (seven small launches + fc_tn: ~65 us of kernels per 2.7 ms minibatch, of which the update gets ~15 us back -- 64.6 -> 64.2 ms per
    // iteration, same-box A/B by the debug flag: fc_tn and the column maxima are real work that now shares the machine with fc_dgrad).
    // Fork after heads_bwd, join in front of the slab sums.
    hipStream_t side_stream; hipEvent_t ev_side_fork, ev_side_join;
    bool side_on;               // mi_debug_flags bit 4 clears it (A/B tests)
    struct SideJob { bool armed; LossArgs a; SegTab st; int mode; float* ring; float* fsr; float* log; } side;
    bool rollout_tail;          // bf16 inference passes of <= 256 samples run blocks 2 + 3 as one launch (mi_debug_flags bit 0 clears it: A/B tests)
    float *fs_colmax, fs_grad_coef; int *fs_arg, fs_G;      // feature-sparsity gradient (fs_coef != 0): column maxima / first arg-max rows of the minibatch
    // ... on more than one rank (multirank mode 1): per-column candidates for the max-all-reduce (MI_PTR_FS_KEYS), this rank's own copy, and
    // the global minibatch positions of the pending pass's rows (mi_minibatch_positions)
    long long *fs_keys, *fs_keys_local; int32_t *d_gpos, *h_gpos; int gpos_n; bool fs_global_pending, fs_global_apply;
    // data-parallel collectives (RCCL over xGMI), SURVEY 8(e): one communicator per context, a side stream for the gradient all-reduce
    ncclComm_t comm, comm_grad; int comm_world, comm_rank; hipStream_t comm_stream; hipEvent_t ev_ar_ready, ev_ar_done;   // comm: main-stream collectives; comm_grad: the side stream's
    bool ar_armed, ar_issued, ar_inflight; double* adv_all;
    std::string net_err; std::mutex net_err_mu;   // set by the (void) network program on an unsupported launch (any thread); every entry point reports it as -4
};

// ------------------------------------------------------------------------------------------ layout tables
static void add_tensor(mi_ctx* c, const std::string& name, int64_t n, int kind, int co, int ci, int64_t& ref, int64_t dev) {
    TensorDesc t{name, ref, dev, n, kind, co, ci};
    c->tensors.push_back(t);
    ref += n;
}

static std::string net_err_take(mi_ctx* c) { std::lock_guard<std::mutex> lk(c->net_err_mu); std::string m; m.swap(c->net_err); return m; }
static void net_err_set(mi_ctx* c, const std::string& m) { std::lock_guard<std::mutex> lk(c->net_err_mu); if (c->net_err.empty()) c->net_err = m; }

static void build_impala_layout(mi_ctx* c) {
    // reference order = policy.parameters(): embedder.block{1,2,3}.{conv,res1.conv1,res1.conv2,res2.conv1,res2.conv2}.{weight,bias},
    // embedder.fc.{weight,bias}, fc_policy.{weight,bias}, fc_value.{weight,bias}   
    const int chan[4] = {3, 16, 32, 32};
    const ConvShape first[3] = {CS_3_16_64, CS_16_32_32, CS_32_32_16};
    const ConvShape resid[3] = {CS_16_16_32, CS_32_32_16, CS_32_32_8};
    const char* sub[5] = {"conv", "res1.conv1", "res1.conv2", "res2.conv1", "res2.conv2"};
    int64_t ref = 0;
    for (int b = 0; b < 3; ++b)
        for (int k = 0; k < 5; ++k) {
            const int ci = (k == 0) ? chan[b] : chan[b + 1], co = chan[b + 1];
            const std::string base = "embedder.block" + std::to_string(b + 1) + "." + sub[k];
            ConvLayer L;
            L.shape = (k == 0) ? first[b] : resid[b];
            conv_shape_dims(L.shape, &L.cin, &L.cout, &L.hw);
            L.w_off = ref;
            add_tensor(c, base + ".weight", (int64_t)co * ci * 9, K_CONVW, co, ci, ref, ref);
            L.b_off = ref;
            add_tensor(c, base + ".bias", co, K_PLAIN, 0, 0, ref, ref);
            c->convs.push_back(L);
        }
    c->fc.in = 2048; c->fc.out = c->H;
    c->fc.w_off = ref; add_tensor(c, "embedder.fc.weight", (int64_t)c->H * 2048, K_FCW, c->H, 2048, ref, ref);
    c->fc.b_off = ref; add_tensor(c, "embedder.fc.bias", c->H, K_PLAIN, 0, 0, ref, ref);
    const int64_t h0 = ref;
    c->wh_off = h0; c->bh_off = h0 + (int64_t)(c->A + 1) * c->H;
    add_tensor(c, "fc_policy.weight", (int64_t)c->A * c->H, K_PLAIN, 0, 0, ref, c->wh_off);
    add_tensor(c, "fc_policy.bias", c->A, K_PLAIN, 0, 0, ref, c->bh_off);
    add_tensor(c, "fc_value.weight", c->H, K_PLAIN, 0, 0, ref, c->wh_off + (int64_t)c->A * c->H);
    add_tensor(c, "fc_value.bias", 1, K_PLAIN, 0, 0, ref, c->bh_off + c->A);
    c->n_params = ref;
}
// The armed gradient exchange splits the flat gradient at embedder.fc.weight: region B = [0, fc.w_off) must hold exactly the conv
// layers, region A = [fc.w_off, n_params) the fc layer and the heads (net_backward).  Checked once per context.
static bool impala_regions_ok(const mi_ctx* c) {
    for (const ConvLayer& L : c->convs)
        if (L.w_off < 0 || L.b_off <= L.w_off || L.b_off + L.cout > c->fc.w_off) return false;
    return c->fc.w_off < c->fc.b_off && c->fc.b_off + c->H <= c->wh_off && c->wh_off < c->bh_off && c->bh_off + c->A + 1 == c->n_params;
}

static void build_mlp_layout(mi_ctx* c) {
    // MLPModel (common/model.py:954-980): Linear(in,w) ReLU [Linear(w,w) ReLU]*(depth-2) Linear(w,latent)
    const int d = c->cfg.mlp_depth, w = c->cfg.mlp_width;
    int64_t ref = 0;
    auto lin = [&](const std::string& nm, int in, int out) {
        Linear L; L.in = in; L.out = out;
        L.w_off = ref; add_tensor(c, nm + ".weight", (int64_t)in * out, K_PLAIN, 0, 0, ref, ref);
        L.b_off = ref; add_tensor(c, nm + ".bias", out, K_PLAIN, 0, 0, ref, ref);
        c->mlp.push_back(L);
    };
    lin("embedder.model.0", c->cfg.obs_dim, w);
    for (int k = 0; k < d - 2; ++k) lin("embedder.model.2." + std::to_string(2 * k), w, w);
    lin("embedder.model.3", w, c->H);
    const int64_t h0 = ref;
    c->wh_off = h0; c->bh_off = h0 + (int64_t)(c->A + 1) * c->H;
    add_tensor(c, "fc_policy.weight", (int64_t)c->A * c->H, K_PLAIN, 0, 0, ref, c->wh_off);
    add_tensor(c, "fc_policy.bias", c->A, K_PLAIN, 0, 0, ref, c->bh_off);
    add_tensor(c, "fc_value.weight", c->H, K_PLAIN, 0, 0, ref, c->wh_off + (int64_t)c->A * c->H);
    add_tensor(c, "fc_value.bias", 1, K_PLAIN, 0, 0, ref, c->bh_off + c->A);
    c->n_params = ref;
}

// reference layout <-> device layout for one tensor (host side)
static void to_device_layout(const TensorDesc& t, const float* ref, float* dev) {
    if (t.kind == K_CONVW) {            // [co][ci][3][3] -> [co][tap][ci]
        for (int co = 0; co < t.co; ++co)
            for (int ci = 0; ci < t.ci; ++ci)
                for (int tap = 0; tap < 9; ++tap) dev[((int64_t)co * 9 + tap) * t.ci + ci] = ref[((int64_t)co * t.ci + ci) * 9 + tap];
    } else if (t.kind == K_FCW) {       // columns c*64 + h*8 + w (NCHW flatten, model.py:54-56) -> (h*8+w)*32 + c
        for (int o = 0; o < t.co; ++o)
            for (int ch = 0; ch < 32; ++ch)
                for (int p = 0; p < 64; ++p) dev[(int64_t)o * 2048 + p * 32 + ch] = ref[(int64_t)o * 2048 + ch * 64 + p];
    } else memcpy(dev, ref, t.n * sizeof(float));
}
static void to_ref_layout(const TensorDesc& t, const float* dev, float* ref) {
    if (t.kind == K_CONVW) {
        for (int co = 0; co < t.co; ++co)
            for (int ci = 0; ci < t.ci; ++ci)
                for (int tap = 0; tap < 9; ++tap) ref[((int64_t)co * t.ci + ci) * 9 + tap] = dev[((int64_t)co * 9 + tap) * t.ci + ci];
    } else if (t.kind == K_FCW) {
        for (int o = 0; o < t.co; ++o)
            for (int ch = 0; ch < 32; ++ch)
                for (int p = 0; p < 64; ++p) ref[(int64_t)o * 2048 + ch * 64 + p] = dev[(int64_t)o * 2048 + p * 32 + ch];
    } else memcpy(ref, dev, t.n * sizeof(float));
}

static int upload_flat(mi_ctx* c, float* dbuf, const float* flat, int64_t n) {
    ARG(n == c->n_params, "flat vector length != mi_param_count"); JOIN(c);
    std::vector<float> tmp(n);
    for (auto& t : c->tensors) to_device_layout(t, flat + t.ref_off, tmp.data() + t.dev_off);
    HIPC(hipMemcpyAsync(dbuf, tmp.data(), n * sizeof(float), hipMemcpyHostToDevice, c->stream));
    HIPC(hipStreamSynchronize(c->stream));
    return 0;
}
static int download_flat(mi_ctx* c, const float* dbuf, float* flat, int64_t n) {
    ARG(n == c->n_params, "flat vector length != mi_param_count"); JOIN(c);
    std::vector<float> tmp(n);
    HIPC(hipMemcpyAsync(tmp.data(), dbuf, n * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIPC(hipStreamSynchronize(c->stream));
    for (auto& t : c->tensors) to_ref_layout(t, tmp.data() + t.dev_off, flat + t.ref_off);
    return 0;
}

// ------------------------------------------------------------------------------------------ create / destroy
template <typename T>
static hipError_t dalloc(T** p, size_t count) {
    hipError_t e = hipMalloc((void**)p, count * sizeof(T) + 256);
    if (e == hipSuccess) e = hipMemset(*p, 0, count * sizeof(T) + 256);
    // hipMemset is asynchronous on the NULL stream and the context's stream is non-blocking: without this wait a
    // kernel launched right after (the op-level test hooks do that) can be overtaken by the zero fill
    if (e == hipSuccess) e = hipDeviceSynchronize();
    return e;
}

int mi_create(const mi_config* cfg, mi_ctx** out) {
    ARG(cfg && out, "null cfg/out");
    ARG(cfg->arch == MI_ARCH_IMPALA || cfg->arch == MI_ARCH_MLP, "arch");
    ARG(cfg->n_actions >= 1 && cfg->n_actions <= 16, "n_actions must be in [1,16]");
    ARG(cfg->n_steps >= 1 && cfg->n_envs >= 1 && cfg->max_batch >= 1, "n_steps/n_envs/max_batch");
    ARG(cfg->precision == 0 || cfg->precision == 1, "precision must be 0 (fp32) or 1 (bf16 activations)");
    if (cfg->arch == MI_ARCH_MLP) ARG(cfg->obs_dim >= 1 && cfg->mlp_depth >= 2 && cfg->mlp_width >= 1 && cfg->out_dim >= 1, "mlp dims");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(-3, "no HIP device: the MI355X library has no CPU fallback");
    HIPC(hipSetDevice(cfg->device));
    mi_ctx* c = new mi_ctx();
    c->cfg = *cfg;
    c->T = cfg->n_steps; c->E = cfg->n_envs; c->A = cfg->n_actions;
    c->H = (cfg->arch == MI_ARCH_IMPALA) ? 256 : cfg->out_dim;
    c->NB = cfg->max_batch < cfg->n_envs ? cfg->n_envs : cfg->max_batch;
    c->bf = (cfg->arch == MI_ARCH_IMPALA) && cfg->precision == 1;
    c->es = c->bf ? 2.0 : 4.0;
    if (cfg->stream) { c->stream = (hipStream_t)cfg->stream; c->own_stream = false; }
    else { HIPC(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)); c->own_stream = true; }
    HIPC(hipEventCreateWithFlags(&c->ev_side_fork, hipEventDisableTiming)); HIPC(hipEventCreateWithFlags(&c->ev_side_join, hipEventDisableTiming));
    c->side_stream = nullptr; c->side_on = true; c->side.armed = false;
    if (cfg->arch == MI_ARCH_IMPALA) build_impala_layout(c); else build_mlp_layout(c);
    if (cfg->arch == MI_ARCH_IMPALA && !impala_regions_ok(c)) { delete c; return fail(-1, "internal: parameter layout does not split at embedder.fc.weight"); }

    const int64_t P = c->n_params, T = c->T, E = c->E, NB = c->NB;
    HIPC(dalloc(&c->params, P)); HIPC(dalloc(&c->grads, P)); HIPC(dalloc(&c->adam_m, P)); HIPC(dalloc(&c->adam_v, P));
    HIPC(dalloc(&c->rew, T * E)); HIPC(dalloc(&c->done, T * E)); HIPC(dalloc(&c->logp, T * E));
    HIPC(dalloc(&c->adv, T * E)); HIPC(dalloc(&c->ret, T * E)); HIPC(dalloc(&c->value, (T + 1) * E));
    HIPC(dalloc(&c->act, T * E)); HIPC(dalloc(&c->adv_stats, 4));
    c->frames = nullptr; c->obsf = nullptr; c->stage_frames = nullptr; c->stage_obs = nullptr;
    if (cfg->arch == MI_ARCH_IMPALA) {
        c->obs_bytes_per_env = 64 * 64 * 3;
        HIPC(dalloc(&c->frames, (size_t)(T + 1) * E * c->obs_bytes_per_env));
        HIPC(dalloc(&c->stage_frames, (size_t)NB * c->obs_bytes_per_env));
        const int chan[4] = {3, 16, 32, 32};
        int hin = 64;
        for (int b = 0; b < 3; ++b) {
            Block& k = c->blk[b];
            k.cin = chan[b]; k.cout = chan[b + 1]; k.hin = hin;
            const size_t X = (size_t)NB * hin * hin * k.cout, p = X / 4;
            k.C = nullptr;                       // conv output before the pool: bf16 mode keeps it in LDS (fused conv + pool kernels)
            if (!c->bf) HIPC(dalloc(&k.C, X));
            HIPC(dalloc(&k.PI, p));
            HIPC(dalloc(&k.P0, p)); HIPC(dalloc(&k.A1, p)); HIPC(dalloc(&k.P1, p)); HIPC(dalloc(&k.A2, p)); HIPC(dalloc(&k.P2, p));
            hin /= 2;
        }
        c->GC = nullptr;                         // gradient of the pre-pool conv output: bf16 mode rebuilds it in LDS (PoolStage)
        if (!c->bf) HIPC(dalloc(&c->GC, (size_t)NB * 64 * 64 * 16));
        for (int k = 0; k < 3; ++k) HIPC(dalloc(&c->GP[k], (size_t)NB * 32 * 32 * 16));
        // every conv layer owns a slab region (persistent grids never exceed 4 workgroups x 256 CUs)
        c->slab_floats = 0;
        for (size_t l = 0; l < c->convs.size() && l < 15; ++l) {
            c->slab_off[l] = (long long)c->slab_floats;
            c->slab_floats += (size_t)1024 * (size_t)(c->convs[l].cout * 9 * c->convs[l].cin + c->convs[l].cout);
        }
        HIPC(dalloc(&c->slabs, c->slab_floats));
        HIPC(hipMalloc((void**)&c->d_slab_desc, sizeof(SlabDesc) * 15)); c->slab_desc_n = 0; c->slab_desc_cached_n = -1;
        HIPC(dalloc(&c->fs_scratch, (size_t)MI_MAX_SEG * 128 * 2048));      // (segments x) FS_GROUPS x 2048 partial column maxima (misc.hip)
        HIPC(dalloc(&c->fs_colmax, (size_t)2048)); HIPC(dalloc(&c->fs_arg, (size_t)2048));
        HIPC(dalloc(&c->fs_keys, (size_t)2048)); HIPC(dalloc(&c->fs_keys_local, (size_t)2048)); HIPC(dalloc(&c->d_gpos, (size_t)NB));
        HIPC(hipHostMalloc((void**)&c->h_gpos, (size_t)NB * 4, hipHostMallocDefault));
    } else {
        c->obs_bytes_per_env = (size_t)cfg->obs_dim * sizeof(float);
        HIPC(dalloc(&c->obsf, (size_t)(T + 1) * E * cfg->obs_dim));
        HIPC(dalloc(&c->stage_obs, (size_t)NB * cfg->obs_dim));
        c->mlp_act.resize(c->mlp.size() + 1);
        HIPC(dalloc(&c->mlp_act[0], (size_t)NB * cfg->obs_dim));
        for (size_t l = 0; l < c->mlp.size(); ++l) HIPC(dalloc(&c->mlp_act[l + 1], (size_t)NB * c->mlp[l].out));
        c->GC = nullptr; c->slabs = nullptr; c->fs_scratch = nullptr; c->d_slab_desc = nullptr; c->slab_desc_n = 0; c->slab_desc_cached_n = -1;
        const int wmax = cfg->mlp_width > c->H ? cfg->mlp_width : c->H;
        for (int k = 0; k < 2; ++k) HIPC(dalloc(&c->GP[k], (size_t)NB * wmax));
        c->GP[2] = nullptr;
    }
    HIPC(dalloc(&c->feat, (size_t)NB * c->H)); HIPC(dalloc(&c->dfeat, (size_t)NB * c->H));
    HIPC(dalloc(&c->hout, (size_t)NB * (c->A + 1))); HIPC(dalloc(&c->dY, (size_t)NB * (c->A + 1)));
    HIPC(dalloc(&c->d_lp, (size_t)NB * c->A));
    const size_t gws = (size_t)8 << 20;
    HIPC(dalloc(&c->gemm_ws, gws)); c->gemm_ws_floats = gws;
    HIPC(dalloc(&c->col_ws, (size_t)64 * 4096));
    HIPC(dalloc(&c->fs_val, 4));
    HIPC(dalloc(&c->lut, 256));
    {
        float h[256];
        for (int k = 0; k < 256; ++k) h[k] = (float)((double)k / 255.0);   // ScaledFloatFrame: obs / 255.0 in fp64, then fp32
        HIPC(hipMemcpy(c->lut, h, sizeof(h), hipMemcpyHostToDevice));
    }
    HIPC(dalloc(&c->lut16, 256));
    {
        float h[256]; for (int k = 0; k < 256; ++k) h[k] = (float)((double)k / 255.0);
        std::vector<uint16_t> b = host_to_bf16(h, 256);
        HIPC(hipMemcpy(c->lut16, b.data(), 512, hipMemcpyHostToDevice));
    }
    HIPC(dalloc(&c->d_idx, (size_t)NB));
    HIPC(dalloc(&c->loss_partial, (size_t)(loss_blocks(NB) + 1 + MI_MAX_SEG) * 32));
    HIPC(dalloc(&c->loss_stats, 64));
    c->log_cap = 4096; c->log_count = 0;
    HIPC(dalloc(&c->loss_log, (size_t)c->log_cap * 8));
    HIPC(dalloc(&c->stats_ring, (size_t)c->log_cap * 32)); HIPC(dalloc(&c->fs_ring, (size_t)c->log_cap)); HIPC(dalloc(&c->fs_parts, (size_t)MI_MAX_SEG * 8));
    HIPC(dalloc(&c->sumsq, 2 + 128)); HIPC(dalloc(&c->gnorm, 2));
    HIPC(dalloc(&c->d_u, (size_t)E));
    HIPC(dalloc(&c->d_pack, (size_t)3 * E)); HIPC(dalloc(&c->d_rd, (size_t)2 * E));
    // written / read by a RUNNING kernel while the host polls the ticket: fine-grained coherent mapping, whatever HIP_HOST_COHERENT says
    const unsigned hflags = hipHostMallocCoherent | hipHostMallocMapped;
    HIPC(hipHostMalloc((void**)&c->h_pack, (size_t)3 * E * 4, hflags)); HIPC(hipHostMalloc((void**)&c->h_rd, (size_t)2 * E * 4, hflags));
    HIPC(dalloc(&c->d_done_ctr, (size_t)16)); HIPC(hipHostMalloc((void**)&c->h_flag, 64, hflags)); memset(c->h_flag, 0, 64); c->roll_ticket = 0;
    HIPC(dalloc(&c->s_act, (size_t)E)); HIPC(dalloc(&c->s_logp, (size_t)E)); HIPC(dalloc(&c->s_val, (size_t)E)); c->staged_valid = false;
    for (int k = 0; k < mi_ctx::IDX_RING; ++k) {
        HIPC(hipHostMalloc((void**)&c->h_idx_ring[k], (size_t)NB * sizeof(int32_t), hflags));      // coherent + mapped: a kernel reads it
        HIPC(hipEventCreateWithFlags(&c->idx_ev[k], hipEventDisableTiming));
        c->idx_used[k] = false;
    }
    c->idx_next = 0; c->idx_ev_deferred = -1;
    c->h_f_floats = (size_t)4 * (E > 64 ? E : 64);
    HIPC(hipHostMalloc((void**)&c->h_f, c->h_f_floats * sizeof(float)));
    HIPC(hipHostMalloc((void**)&c->h_i, (size_t)E * sizeof(int32_t)));
    c->comm = nullptr; c->comm_grad = nullptr; c->comm_world = 1; c->comm_rank = 0; c->comm_stream = nullptr; c->ev_ar_ready = c->ev_ar_done = nullptr;
    c->ar_armed = c->ar_issued = c->ar_inflight = false; c->adv_all = nullptr;
    c->fs_grad_coef = 0.f; c->fs_G = 0; c->rollout_tail = true; c->no_pull = false; c->copy_rate_bytes_per_us = getenv("MI355_COPY_GBPS") ? atof(getenv("MI355_COPY_GBPS")) * 1000.0 : 40000.0;
    if (cfg->arch != MI_ARCH_IMPALA) { c->fs_colmax = nullptr; c->fs_arg = nullptr; c->fs_keys = c->fs_keys_local = nullptr; c->d_gpos = nullptr; c->h_gpos = nullptr; }
    c->gpos_n = -1; c->fs_global_pending = c->fs_global_apply = false;
    c->n_groups = 1; c->groups_live = false; c->main_stream = c->stream;
    for (int g = 0; g < mi_ctx::MAX_GROUPS; ++g) { c->gw[g] = nullptr; c->gs[g] = nullptr; c->ev_fork[g] = c->ev_join[g] = nullptr; c->g_forked[g] = c->g_busy[g] = c->g_last[g] = c->g_dirty[g] = false; c->g_ticket[g] = 0; }
    c->multirank = 0; c->pending_n = -1; c->sal_dc = nullptr; c->sal_dx = nullptr; c->sal_src = nullptr;
    c->fc_wp = c->fc_wt = nullptr; c->fc_packed_valid = false;
    if (c->bf) { HIPC(dalloc(&c->fc_wp, (size_t)256 * 2048)); HIPC(dalloc(&c->fc_wt, (size_t)256 * 2048)); }
    c->banks = nullptr; c->d_bank_desc = nullptr; c->n_banks = 0; c->c1_bank = nullptr;
    if (c->bf && c->cfg.arch == MI_ARCH_IMPALA) HIPC(dalloc(&c->c1_bank, (size_t)conv1_bank_elems()));
    for (auto& L : c->convs) { L.bank_f = -1; L.bank_d = -1; }
    if (c->bf) {
        std::vector<BankDesc> desc;
        long long off = 0;
        for (auto& L : c->convs) {
            if (L.cin == 3) continue;                                   // block1.conv stays on the fp32-MFMA kernel
            BankDesc f{L.w_off, off, L.cout, L.cin, L.cout, L.cin, 0, bank_ws(L.cin), L.cin == 32 ? 9 : 5};
            L.bank_f = off; off += (long long)f.rows * f.ws; desc.push_back(f);
            BankDesc d{L.w_off, off, L.cin, L.cout, L.cout, L.cin, 1, bank_ws(L.cout), L.cout == 32 ? 9 : 5};   // dgrad pass: cin_pass = forward cout
            L.bank_d = off; off += (long long)d.rows * d.ws; desc.push_back(d);
        }
        c->n_banks = (int)desc.size();
        HIPC(dalloc(&c->banks, (size_t)off));
        HIPC(hipMalloc((void**)&c->d_bank_desc, desc.size() * sizeof(BankDesc)));
        HIPC(hipMemcpy(c->d_bank_desc, desc.data(), desc.size() * sizeof(BankDesc), hipMemcpyHostToDevice));
    }
    c->gru_x = c->gru_dg = nullptr; c->sal_keep_x = c->bwd_from_dfeat = false;
    c->gru_on = false; c->gru_wih = c->gru_whh = c->gru_bih = c->gru_bhh = c->h_state = c->h_masked = c->gru_gi = c->gru_gh = c->d_done = nullptr;
    HIPC(hipDeviceSynchronize());
    *out = c;
    return 0;
}

int mi_destroy(mi_ctx* c) {
    if (!c) return 0;
    mi_comm_destroy(c);
    for (int g = 0; g < mi_ctx::MAX_GROUPS; ++g)
        if (c->gw[g]) {
            GroupWorker* w = c->gw[g];
            w->quit.store(true); { std::lock_guard<std::mutex> lk(w->mu); } w->cv.notify_one();
            w->th.join(); delete w; c->gw[g] = nullptr;
        }
    for (int g = 0; g < mi_ctx::MAX_GROUPS; ++g) if (c->gs[g]) hipStreamSynchronize(c->gs[g]);
    prof_harvest(c);
    hipStreamSynchronize(c->stream);
    for (int g = 0; g < mi_ctx::MAX_GROUPS; ++g) if (c->gs[g]) { hipStreamDestroy(c->gs[g]); hipEventDestroy(c->ev_fork[g]); hipEventDestroy(c->ev_join[g]); }
    for (hipEvent_t e : c->prof.pool) hipEventDestroy(e);
    float* fl[] = {c->params, c->grads, c->adam_m, c->adam_v, c->rew, c->done, c->logp, c->adv, c->ret, c->value, c->obsf,
                   c->feat, c->dfeat, c->hout, c->dY, c->GC, c->GP[0], c->GP[1], c->GP[2], c->slabs, c->gemm_ws, c->col_ws,
                   c->fs_scratch, c->fs_val, c->lut, c->stage_obs, c->loss_partial, c->loss_stats, c->loss_log, c->gnorm,
                   c->d_u, c->d_lp};
    for (float* p : fl) if (p) hipFree(p);
    if (c->cfg.arch == MI_ARCH_IMPALA)
        for (int b = 0; b < 3; ++b) { Block& k = c->blk[b]; hipFree(k.C); hipFree(k.PI); hipFree(k.P0); hipFree(k.A1); hipFree(k.P1); hipFree(k.A2); hipFree(k.P2); }
    for (float* p : c->mlp_act) if (p) hipFree(p);
    if (c->frames) hipFree(c->frames);
    if (c->stage_frames) hipFree(c->stage_frames);
    hipFree(c->s_act); hipFree(c->s_logp); hipFree(c->s_val);
    if (c->fc_wp) hipFree(c->fc_wp); if (c->fc_wt) hipFree(c->fc_wt);
    if (c->banks) hipFree(c->banks); if (c->d_bank_desc) hipFree(c->d_bank_desc); if (c->c1_bank) hipFree(c->c1_bank);
    hipFree(c->stats_ring); hipFree(c->fs_ring); hipFree(c->fs_parts); if (c->d_slab_desc) hipFree(c->d_slab_desc); if (c->sal_dc) hipFree(c->sal_dc); if (c->sal_dx) hipFree(c->sal_dx); hipFree(c->d_pack); hipFree(c->d_rd); hipHostFree(c->h_pack); hipHostFree(c->h_rd); hipFree(c->d_done_ctr); if (c->side_stream) hipStreamDestroy(c->side_stream); hipEventDestroy(c->ev_side_fork); hipEventDestroy(c->ev_side_join); hipHostFree(c->h_flag);
    { float* gr[] = {c->gru_wih, c->gru_whh, c->gru_bih, c->gru_bhh, c->h_state, c->h_masked, c->gru_gi, c->gru_gh, c->d_done, c->gru_x, c->gru_dg}; for (float* q : gr) if (q) hipFree(q); }
    hipFree(c->act); hipFree(c->adv_stats); hipFree(c->d_idx); hipFree(c->sumsq);
    if (c->fs_colmax) hipFree(c->fs_colmax); if (c->fs_arg) hipFree(c->fs_arg);
    if (c->fs_keys) hipFree(c->fs_keys); if (c->fs_keys_local) hipFree(c->fs_keys_local); if (c->d_gpos) hipFree(c->d_gpos); if (c->h_gpos) hipHostFree(c->h_gpos);
    for (int k = 0; k < mi_ctx::IDX_RING; ++k) { hipHostFree(c->h_idx_ring[k]); hipEventDestroy(c->idx_ev[k]); }
    hipHostFree(c->h_f); hipHostFree(c->h_i);
    if (c->own_stream) hipStreamDestroy(c->stream);
    delete c;
    return 0;
}

void* mi_host_alloc(size_t bytes) {
    void* p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 1) != hipSuccess) { g_err = "hipHostMalloc failed"; return nullptr; }
    return p;
}
void mi_host_free(void* p) { if (p) hipHostFree(p); }
// Page-lock caller-owned memory in place (an env's own frame buffer): mi_rollout_submit / mi_put_obs then DMA straight out of it, no
// staging copy on the host.  Fails (-2) where the runtime refuses the range (e.g. pages already registered through another pointer).
int mi_host_register(void* p, size_t bytes) {
    ARG(p && bytes, "null");
    if (hipHostRegister(p, bytes, hipHostRegisterDefault) != hipSuccess) { (void)hipGetLastError(); return fail(-2, "hipHostRegister refused the range"); }
    return 0;
}
int mi_host_unregister(void* p) {
    ARG(p, "null");
    if (hipHostUnregister(p) != hipSuccess) { (void)hipGetLastError(); return fail(-2, "hipHostUnregister failed"); }
    return 0;
}

int mi_sync(mi_ctx* c) { ARG(c, "ctx"); JOIN(c); HIPC(hipStreamSynchronize(c->stream)); return 0; }

int64_t mi_param_count(mi_ctx* c) { return c ? c->n_params : -1; }
int mi_set_params(mi_ctx* c, const float* flat, int64_t n) { ARG(c && flat, "null"); c->fc_packed_valid = false; return upload_flat(c, c->params, flat, n); }
int mi_get_params(mi_ctx* c, float* flat, int64_t n) { ARG(c && flat, "null"); return download_flat(c, c->params, flat, n); }
// PPO.train hands the freshly updated policy to the validation rollouts (agents/ppo.py:241-252 run them with the SAME policy object):
// here the inference-only twin context takes the parameters device to device, in stream order on both sides (no host round trip).
int mi_copy_params(mi_ctx* dst, mi_ctx* src) {
    ARG(dst && src && dst != src, "null / same context"); JOIN(src); JOIN(dst);
    ARG(dst->cfg.arch == src->cfg.arch && dst->n_params == src->n_params && dst->A == src->A && dst->H == src->H && dst->cfg.device == src->cfg.device,
        "contexts of different architecture / size / device");
    hipEvent_t ready = nullptr, done = nullptr;
    HIPC(hipEventCreateWithFlags(&ready, hipEventDisableTiming)); HIPC(hipEventCreateWithFlags(&done, hipEventDisableTiming));
    HIPC(hipEventRecord(ready, src->stream));                      // the optimizer step that wrote src's parameters
    HIPC(hipStreamWaitEvent(dst->stream, ready, 0));
    HIPC(hipMemcpyAsync(dst->params, src->params, (size_t)src->n_params * 4, hipMemcpyDeviceToDevice, dst->stream));
    HIPC(hipEventRecord(done, dst->stream));
    HIPC(hipStreamWaitEvent(src->stream, done, 0));                // src's next optimizer step must not overtake the copy
    dst->fc_packed_valid = false;                                  // packed bf16 filter images are rebuilt before dst's next pass
    HIPC(hipEventDestroy(ready)); HIPC(hipEventDestroy(done));     // (released by the runtime once they have completed)
    return 0;
}
int mi_get_grads(mi_ctx* c, float* flat, int64_t n) { ARG(c && flat, "null"); return download_flat(c, c->grads, flat, n); }
int mi_set_adam_state(mi_ctx* c, const float* m, const float* v, int64_t n) {
    ARG(c && m && v, "null");
    int r = upload_flat(c, c->adam_m, m, n);
    return r ? r : upload_flat(c, c->adam_v, v, n);
}
int mi_get_adam_state(mi_ctx* c, float* m, float* v, int64_t n) {
    ARG(c && m && v, "null");
    int r = download_flat(c, c->adam_m, m, n);
    return r ? r : download_flat(c, c->adam_v, v, n);
}

// ------------------------------------------------------------------------------------------ rollout storage
int mi_put_obs(mi_ctx* c, int32_t t, const void* obs, size_t bytes) {
    ARG(c && obs, "null"); JOIN(c); ARG(t >= 0 && t <= c->T, "t out of range");
    const size_t want = (size_t)c->E * c->obs_bytes_per_env;
    ARG(bytes == want, "obs byte count != E * bytes_per_env");
    char* dst = c->frames ? (char*)c->frames : (char*)c->obsf;
    HIPC(hipMemcpyAsync(dst + (size_t)t * want, obs, bytes, hipMemcpyHostToDevice, c->stream));
    return 0;
}
int mi_get_obs(mi_ctx* c, int32_t t, void* obs, size_t bytes) {
    ARG(c && obs, "null"); JOIN(c); ARG(t >= 0 && t <= c->T, "t out of range");
    const size_t want = (size_t)c->E * c->obs_bytes_per_env;
    ARG(bytes == want, "obs byte count != E * bytes_per_env");
    const char* src = c->frames ? (const char*)c->frames : (const char*)c->obsf;
    HIPC(hipMemcpyAsync(obs, src + (size_t)t * want, bytes, hipMemcpyDeviceToHost, c->stream));
    HIPC(hipStreamSynchronize(c->stream));
    return 0;
}
int mi_put_step(mi_ctx* c, int32_t t, const float* rew, const float* done) {
    ARG(c && rew && done, "null"); JOIN(c); ARG(t >= 0 && t < c->T, "t out of range");
    const size_t b = (size_t)c->E * sizeof(float);
    HIPC(hipMemcpyAsync(c->rew + (size_t)t * c->E, rew, b, hipMemcpyHostToDevice, c->stream));
    HIPC(hipMemcpyAsync(c->done + (size_t)t * c->E, done, b, hipMemcpyHostToDevice, c->stream));
    return 0;
}
int mi_put_policy_outputs(mi_ctx* c, int32_t t, const int32_t* act, const float* logp, const float* value) {
    ARG(c, "null"); JOIN(c); ARG(t >= 0 && t <= c->T, "t out of range");
    const size_t E = c->E;
    if (act) { ARG(t < c->T, "act at t==T"); HIPC(hipMemcpyAsync(c->act + t * E, act, E * 4, hipMemcpyHostToDevice, c->stream)); }
    if (logp) { ARG(t < c->T, "logp at t==T"); HIPC(hipMemcpyAsync(c->logp + t * E, logp, E * 4, hipMemcpyHostToDevice, c->stream)); }
    if (value) HIPC(hipMemcpyAsync(c->value + t * E, value, E * 4, hipMemcpyHostToDevice, c->stream));
    HIPC(hipStreamSynchronize(c->stream));     // caller buffers may be pageable temporaries
    return 0;
}
static float* field_ptr(mi_ctx* c, int f, int64_t* n) {
    const int64_t TE = (int64_t)c->T * c->E;
    *n = TE;
    switch (f) {
        case MI_F_REW: return c->rew; case MI_F_DONE: return c->done; case MI_F_LOGP: return c->logp;
        case MI_F_ADV: return c->adv; case MI_F_RET: return c->ret;
        case MI_F_VALUE: *n = TE + c->E; return c->value;
        default: return nullptr;
    }
}
int mi_read_field(mi_ctx* c, int32_t f, float* out, int64_t n) {
    ARG(c && out, "null"); JOIN(c);
    if (f == MI_F_ACT) {
        ARG(n == (int64_t)c->T * c->E, "length");
        std::vector<int32_t> tmp(n);
        HIPC(hipMemcpyAsync(tmp.data(), c->act, n * 4, hipMemcpyDeviceToHost, c->stream));
        HIPC(hipStreamSynchronize(c->stream));
        for (int64_t k = 0; k < n; ++k) out[k] = (float)tmp[k];
        return 0;
    }
    int64_t want; float* p = field_ptr(c, f, &want);
    ARG(p, "field"); ARG(n == want, "length");
    HIPC(hipMemcpyAsync(out, p, n * 4, hipMemcpyDeviceToHost, c->stream));
    HIPC(hipStreamSynchronize(c->stream));
    return 0;
}
int mi_write_field(mi_ctx* c, int32_t f, const float* in, int64_t n) {
    ARG(c && in, "null"); JOIN(c);
    if (f == MI_F_ACT) {
        ARG(n == (int64_t)c->T * c->E, "length");
        std::vector<int32_t> tmp(n);
        for (int64_t k = 0; k < n; ++k) tmp[k] = (int32_t)in[k];
        HIPC(hipMemcpyAsync(c->act, tmp.data(), n * 4, hipMemcpyHostToDevice, c->stream));
        HIPC(hipStreamSynchronize(c->stream));
        return 0;
    }
    int64_t want; float* p = field_ptr(c, f, &want);
    ARG(p, "field"); ARG(n == want, "length");
    HIPC(hipMemcpyAsync(p, in, n * 4, hipMemcpyHostToDevice, c->stream));
    HIPC(hipStreamSynchronize(c->stream));
    return 0;
}

// ------------------------------------------------------------------------------------------ profiler
static hipEvent_t prof_event(mi_ctx* c) {
    hipEvent_t e;
    if (!c->prof.pool.empty()) { e = c->prof.pool.back(); c->prof.pool.pop_back(); return e; }
    hipEventCreate(&e);
    return e;
}
static void prof_harvest(mi_ctx* c) {
    if (c->prof.pend.empty()) return;
    hipStreamSynchronize(c->stream);
    for (int g = 0; g < mi_ctx::MAX_GROUPS; ++g) if (c->gs[g]) hipStreamSynchronize(c->gs[g]);
    for (auto& p : c->prof.pend) {
        float ms = 0.f;
        hipEventElapsedTime(&ms, p.a, p.b);
        c->prof.ms[p.phase][p.cls] += ms; c->prof.launches[p.phase][p.cls]++; c->prof.units[p.phase][p.cls] += p.units;
        c->prof.bytes[p.phase][p.cls] += p.bytes; c->prof.flops[p.phase][p.cls] += p.flops;
        c->prof.pool.push_back(p.a); c->prof.pool.push_back(p.b);
    }
    c->prof.pend.clear();
}
struct ProfScope {
    mi_ctx* c; ProfPending p; bool live;
    // bytes / flops: ALGORITHMIC figures of this launch (layer-boundary model)
    ProfScope(mi_ctx* c_, int cls, long long units, double bytes, double flops)
        : c(c_), live(!tl_stream && c_->prof.on && (c_->prof.phase == 1 ? c_->prof.sample_now : c_->prof.all_phases)) {
        if (!live) return;
        p.a = prof_event(c); p.b = prof_event(c); p.cls = cls; p.phase = c->prof.phase; p.units = units; p.bytes = bytes; p.flops = flops;
        hipEventRecord(p.a, c->stream);
    }
    ~ProfScope() {
        if (!live) return;
        hipEventRecord(p.b, c->stream);
        c->prof.pend.push_back(p);
        if (c->prof.pend.size() >= 4096) prof_harvest(c);
    }
};
int mi_profile_enable(mi_ctx* c, int32_t enabled) {
    ARG(c, "null");
    if (!enabled) prof_harvest(c);
    c->prof.on = (enabled & 0xff) != 0; c->prof.all_phases = (enabled & 0xff) == 2;
    c->prof.period = (enabled >> 8) > 0 ? (enabled >> 8) : 1; c->prof.mb_count = 0; c->prof.sample_now = true;
    return 0;
}
const char* mi_profile_class_name(int32_t id) { return (id >= 0 && id < PC_COUNT) ? kProfNames[id] : ""; }
int mi_profile_read(mi_ctx* c, double* rows, int32_t max_rows, int32_t* n_rows, int32_t reset) {
    ARG(c && rows && n_rows, "null"); JOIN(c);
    prof_harvest(c);
    int n = 0;
    for (int ph = 0; ph < 2; ++ph)
        for (int k = 0; k < PC_COUNT; ++k)
            if (c->prof.launches[ph][k] > 0 && n < max_rows) {
                double* r = rows + (size_t)n * 7;
                r[0] = k; r[1] = ph; r[2] = (double)c->prof.launches[ph][k]; r[3] = c->prof.ms[ph][k]; r[4] = (double)c->prof.units[ph][k];
                r[5] = c->prof.bytes[ph][k]; r[6] = c->prof.flops[ph][k];
                ++n;
            }
    *n_rows = n;
    if (reset) { memset(c->prof.ms, 0, sizeof(c->prof.ms)); memset(c->prof.launches, 0, sizeof(c->prof.launches)); memset(c->prof.units, 0, sizeof(c->prof.units));
                 memset(c->prof.bytes, 0, sizeof(c->prof.bytes)); memset(c->prof.flops, 0, sizeof(c->prof.flops)); }
    return 0;
}

// ------------------------------------------------------------------------------------------ network program
struct InputSrc { const void* base; const int32_t* idx; long long first; };   // frames or obs rows

static void conv_fwd(mi_ctx* c, const ConvLayer& L, const void* in, const InputSrc* src, int relu_in, const float* res, float* out, int n) {
    ConvArgs a{};
    a.in = src ? src->base : in; a.idx = src ? src->idx : nullptr; a.in_base = src ? src->first : 0;
    a.w = c->params + L.w_off; a.bias = c->params + L.b_off; a.res = res; a.mask = nullptr; a.out = out;
    a.lut = c->lut; a.n = n; a.relu_in = relu_in; a.bf16 = c->bf;
    a.wbank = (c->bf && L.bank_f >= 0) ? c->banks + L.bank_f : nullptr;
    a.lut16 = c->bf ? c->lut16 : nullptr;
    const double px = (double)n * L.hw * L.hw, es = c->es;
    ProfScope ps(c, PC_CONV_FWD + (int)L.shape, n, px * ((L.cin == 3 ? 3.0 : es * L.cin) + es * L.cout * (res ? 2 : 1)), px * 18.0 * L.cin * L.cout);
    launch_conv_fwd(L.shape, a, CUR(c));
}
static void conv_dgrad(mi_ctx* c, const ConvLayer& L, const float* dout, const float* mask, const float* res, float* din, int n, const uint8_t* pool_arg = nullptr) {
    ConvArgs a{};
    a.pool_arg = pool_arg;
    a.in = dout; a.w = c->params + L.w_off; a.bias = nullptr; a.res = res; a.mask = mask; a.out = din;
    a.lut = c->lut; a.n = n; a.relu_in = 0; a.bf16 = c->bf;
    a.wbank = (c->bf && L.bank_d >= 0) ? c->banks + L.bank_d : nullptr;
    const double px = (double)n * L.hw * L.hw;
    const double pool_b = pool_arg ? c->es * (px / 4 * L.cout + 2.0 * px * L.cout) : 0.0;      // POOLIN: the max-pool backward (p + 2X of SURVEY 8(d)) rides along
    ProfScope ps(c, PC_CONV_DGRAD + (int)L.shape, n, px * c->es * (L.cout + L.cin * (1 + (mask ? 1 : 0) + (res ? 1 : 0))) + pool_b, px * 18.0 * L.cin * L.cout);
    launch_conv_dgrad(L.shape, a, CUR(c));
}
static void conv_wgrad(mi_ctx* c, const ConvLayer& L, const void* in, const InputSrc* src, int relu_in, const float* dout, int n, const uint8_t* pool_arg = nullptr) {
    WgradArgs a{};
    a.pool_arg = pool_arg;
    a.in = src ? src->base : in; a.idx = src ? src->idx : nullptr; a.in_base = src ? src->first : 0;
    const int layer = (int)(&L - c->convs.data());
    a.dout = dout; a.partial = c->slabs + c->slab_off[layer]; a.lut = c->lut; a.n = n; a.relu_in = relu_in; a.bf16 = c->bf;
    a.lut16 = c->bf ? c->lut16 : nullptr;
    const int grid = wgrad_grid_for(L.shape, n, c->bf);
    if (grid < 1) return;
    if (grid > 1024) { net_err_set(c, "weight-gradient launch needs more than the 1024 slabs a layer owns"); return; }
    const double px = (double)n * L.hw * L.hw;
    { // SURVEY 8(d) layer-boundary bytes; block1.conv from the pooled gradient also carries the max-pool backward (p + 2X)
      const double pool_b = (pool_arg && L.cin == 3) ? c->es * (px / 4 * L.cout + 2.0 * px * L.cout) : 0.0;
      ProfScope ps(c, PC_CONV_WGRAD + (int)L.shape, n, px * ((L.cin == 3 ? 3.0 : c->es * L.cin) + c->es * L.cout) + pool_b, px * 18.0 * L.cin * L.cout);
      launch_conv_wgrad(L.shape, a, CUR(c)); }
    // the slabs of all layers are summed by ONE launch at the end of net_backward (conv_wgrad_reduce_all)
    const int wlen = L.cout * 9 * L.cin;
    c->h_slab_desc[c->slab_desc_n++] = SlabDesc{c->slab_off[layer], (long long)L.w_off, (long long)L.b_off, grid, wlen + L.cout, wlen};
}
// Gradient all-reduce of an ARMED backward pass (mi_allreduce_arm): a region of the flat gradient goes to the side stream as soon as
// the main stream has written it for the last time.  Region A = [fc.weight .. end) (embedder.fc + heads: 84 % of the parameters, final
// right after the first three launches of the backward pass, so their exchange hides behind the whole conv stack's backward);
// region B = [0, fc.weight) (the 15 conv layers: final only once the per-workgroup slabs are summed, at the very end).
static void issue_grad_allreduce(mi_ctx* c, int64_t off, int64_t n, bool last) {
    if (!c->ar_armed || !c->comm || n <= 0) return;
    hipEventRecord(c->ev_ar_ready, CUR(c));
    hipStreamWaitEvent(c->comm_stream, c->ev_ar_ready, 0);
    ncclResult_t r = ncclAllReduce(c->grads + off, c->grads + off, (size_t)n, ncclFloat, ncclSum, c->comm_grad, c->comm_stream);
    if (r != ncclSuccess) { net_err_set(c, std::string("ncclAllReduce (gradients): ") + ncclGetErrorString(r)); return; }
    if (last) { hipEventRecord(c->ev_ar_done, c->comm_stream); c->ar_armed = false; c->ar_issued = true; c->ar_inflight = true; }
}

static void conv_wgrad_reduce_all(mi_ctx* c, int n, int first = 0) {          // first: entries [0, first) are summed already (net_backward's second fork)
    if (c->slab_desc_n <= 0) return;
    int max_len = 0; double bytes = 0;
    for (int k = first; k < c->slab_desc_n; ++k) { max_len = std::max(max_len, c->h_slab_desc[k].slab_len); bytes += 4.0 * c->h_slab_desc[k].nslab * c->h_slab_desc[k].slab_len; }
    // the descriptor table only depends on the batch size: re-uploaded when it changes (pageable source copied at call time)
    if (c->slab_desc_cached_n != n) {
        hipMemcpyAsync(c->d_slab_desc, c->h_slab_desc, sizeof(SlabDesc) * c->slab_desc_n, hipMemcpyHostToDevice, CUR(c));
        c->slab_desc_cached_n = n;
    }
    { ProfScope ps(c, PC_SLAB_REDUCE, n, bytes, 0.0);
      launch_reduce_all_slabs(c->slabs, c->grads, c->d_slab_desc + first, c->slab_desc_n - first, max_len, CUR(c)); }
    c->slab_desc_n = 0;
}

static void linear_fwd(mi_ctx* c, const float* X, int relu_x, const float* W, const float* b, float* Y, int n, int in, int out, int relu_out, int x_bf16 = 0) {
    GemmArgs g{};
    g.ws = tl_ws ? tl_ws : c->gemm_ws; g.ws_floats = tl_ws ? tl_ws_floats : c->gemm_ws_floats;
    g.a_bf16 = x_bf16;
    g.A = X; g.B = W; g.C = Y; g.M = n; g.N = out; g.K = in;
    g.sam = in; g.sak = 1; g.sbk = 1; g.sbn = in; g.ldc = out;
    g.bias = b; g.relu_a = relu_x; g.relu_out = relu_out;
    ProfScope ps(c, PC_GEMM, n, 4.0 * ((double)n * in + (double)in * out + (double)n * out), 2.0 * n * in * out);
    launch_gemm(g, CUR(c));
}
// dX = dY W  (* mask > 0)
static void linear_dgrad(mi_ctx* c, const float* dY, const float* W, const float* mask, float* dX, int n, int in, int out, int x_bf16 = 0) {
    GemmArgs g{};
    g.ws = tl_ws ? tl_ws : c->gemm_ws; g.ws_floats = tl_ws ? tl_ws_floats : c->gemm_ws_floats;
    g.mask_bf16 = x_bf16; g.c_bf16 = x_bf16;          // mask source and dX are activation-typed
    g.A = dY; g.B = W; g.C = dX; g.M = n; g.N = in; g.K = out;
    g.sam = out; g.sak = 1; g.sbk = in; g.sbn = 1; g.ldc = in; g.mask = mask;
    ProfScope ps(c, PC_GEMM, n, 4.0 * ((double)n * out + (double)in * out + (double)n * in * (mask ? 2 : 1)), 2.0 * n * in * out);
    launch_gemm(g, CUR(c));
}
// gW += dY^T relu?(X) ; gb += colsum(dY)
static void linear_wgrad(mi_ctx* c, const float* dY, const float* X, int relu_x, float* gW, float* gb, int n, int in, int out, int x_bf16 = 0) {
    GemmArgs g{};
    g.ws = tl_ws ? tl_ws : c->gemm_ws; g.ws_floats = tl_ws ? tl_ws_floats : c->gemm_ws_floats;
    g.b_bf16 = x_bf16;
    g.A = dY; g.B = X; g.C = gW; g.M = out; g.N = in; g.K = n;
    g.sam = 1; g.sak = out; g.sbk = in; g.sbn = 1; g.ldc = in; g.relu_b = relu_x; g.accumulate = 1;
    ProfScope ps(c, PC_GEMM, n, 4.0 * ((double)n * out + (double)n * in + (double)in * out), 2.0 * n * in * out);
    launch_gemm(g, CUR(c));
    launch_colsum_acc(dY, n, out, out, gb, c->col_ws, CUR(c));
}

static void net_heads(mi_ctx* c, int n, int soff = 0) {
    if (c->H == 256 && c->A + 1 <= 16 && n >= 1024) {      // update-sized batches: dedicated kernel (misc.hip heads_fwd_kernel)
        ProfScope ps(c, PC_GEMM, n, 4.0 * ((double)n * c->H + (double)n * (c->A + 1) + (double)c->H * (c->A + 1)), 2.0 * n * c->H * (c->A + 1));
        launch_heads_fwd(c->feat + (size_t)soff * c->H, c->params + c->wh_off, c->params + c->bh_off, c->hout + (size_t)soff * (c->A + 1), n, c->A + 1, CUR(c));
        return;
    }
    linear_fwd(c, c->feat + (size_t)soff * c->H, 0, c->params + c->wh_off, c->params + c->bh_off, c->hout + (size_t)soff * (c->A + 1), n, c->H, c->A + 1, 0);
}
// h' = GRU(feat, h_state * (1 - done)); feat <- h' ; h_state <- h'   (n == E rows)
static void net_gru(mi_ctx* c, int n, int soff = 0) {
    const int H = c->H;
    const size_t o = (size_t)soff * H;
    launch_mask_rows(c->h_state + o, c->d_done + soff, c->h_masked + o, n, H, CUR(c));
    linear_fwd(c, c->feat + o, 0, c->gru_wih, c->gru_bih, c->gru_gi + 3 * o, n, H, 3 * H, 0);
    linear_fwd(c, c->h_masked + o, 0, c->gru_whh, c->gru_bhh, c->gru_gh + 3 * o, n, H, 3 * H, 0);
    if (c->sal_keep_x) hipMemcpyAsync(c->gru_x + o, c->feat + o, (size_t)n * H * 4, hipMemcpyDeviceToDevice, CUR(c));     // (the gates kernel overwrites feat with h')
    launch_gru_gates(c->gru_gi + 3 * o, c->gru_gh + 3 * o, c->h_masked + o, c->h_state + o, c->feat + o, n, H, CUR(c));
}

static void fc_refresh(mi_ctx* c) {
    if (c->bf && !c->fc_packed_valid) {
        launch_repack_all(c->params, c->banks, c->d_bank_desc, c->n_banks, c->convs.empty() ? nullptr : c->params + c->convs[0].w_off,
                          c->convs.empty() ? nullptr : c->c1_bank, c->params + c->fc.w_off, c->fc_wp, c->fc_wt, CUR(c));
        c->fc_packed_valid = true;
    }
}

// soff: first row of the activation buffers this pass may use (env groups of the pipelined rollout run side by side on
// their own streams, each in its own rows); 0 everywhere else
static void net_forward(mi_ctx* c, const InputSrc& src, int n, bool recurrent = false, bool with_heads = true, bool train = false, int soff = 0) {
    fc_refresh(c);          // bf16 mode: packed fc / conv filter images follow the parameters
    float* const feat = c->feat + (size_t)soff * c->H;
    if (c->cfg.arch == MI_ARCH_IMPALA) {
        const float* prev = nullptr;
        auto view = [&](int b) {       // block b's buffers from row soff on (element size: c->es bytes; arg-max: 1 byte)
            Block k = c->blk[b];
            if (soff) {
                const size_t pe = (size_t)(k.hin / 2) * (k.hin / 2) * k.cout, po = (size_t)soff * pe * (size_t)c->es;
                auto sh = [&](float* q, size_t bytes) { return q ? (float*)((char*)q + bytes) : q; };
                k.C = sh(k.C, po * 4); k.P0 = sh(k.P0, po); k.A1 = sh(k.A1, po); k.P1 = sh(k.P1, po); k.A2 = sh(k.A2, po); k.P2 = sh(k.P2, po);
                k.PI += (size_t)soff * pe;
            }
            return k;
        };
        // rollout-sized inference batches (bf16): blocks 2 and 3 in ONE launch, one workgroup per image (rollout_bf16.hip)
        const bool fused_tail = c->bf && !train && n <= 256 && c->rollout_tail;
        for (int b = 0; b < 3; ++b) {
            Block k = view(b);
            const ConvLayer* L = &c->convs[b * 5];
            if (fused_tail && b == 1) {
                const unsigned short* bk[10]; const float* bb[10];
                for (int q = 0; q < 10; ++q) { bk[q] = c->banks + c->convs[5 + q].bank_f; bb[q] = c->params + c->convs[5 + q].b_off; }
                const Block k3 = view(2);
                { const double px2 = (double)n * 32 * 32, px3 = (double)n * 16 * 16;
                  ProfScope ps(c, PC_RESBLOCK + (int)CS_32_32_16, n, 2.0 * (px2 * 16 + px3 / 4 * 32), px2 * 18.0 * 16 * 32 + 5.0 * px3 * 18.0 * 32 * 32 + 4.0 * (px3 / 4) * 18.0 * 32 * 32);
                  launch_rollout_tail_bf16(prev, k3.P2, n, bk, bb, CUR(c)); }
                prev = k3.P2;
                break;
            }
            if (b == 0 && c->bf) {           // block1.conv + max pool fused: the 64x64x16 conv output never reaches HBM
                ConvArgs a{};
                a.in = src.base; a.idx = src.idx; a.in_base = src.first; a.w = c->params + L[0].w_off; a.bias = c->params + L[0].b_off;
                a.n = n; a.bf16 = 1; a.lut16 = c->lut16; a.wbank = c->c1_bank;
                const double px = (double)n * 64 * 64;
                ProfScope ps(c, PC_CONV_FWD + (int)L[0].shape, n, px * 3.0 + 2.0 * (2.0 * px * 16 + px / 4 * 16), px * 18.0 * 3 * 16);      // conv I + X, pool X + p
                launch_conv1_pool_fwd_bf16(a, c->lut16, k.P0, k.PI, CUR(c));
            } else if (c->bf && L[0].bank_f >= 0) {       // block2.conv / block3.conv + max pool fused as well (convpool_bf16.hip)
                ConvArgs a{};
                a.in = prev; a.bias = c->params + L[0].b_off; a.n = n; a.bf16 = 1; a.wbank = c->banks + L[0].bank_f;
                const double px = (double)n * L[0].hw * L[0].hw;
                ProfScope ps(c, PC_CONV_FWD + (int)L[0].shape, n, 2.0 * (px * L[0].cin + 2.0 * px * L[0].cout + px / 4 * L[0].cout), px * 18.0 * L[0].cin * L[0].cout);      // 8(d): I + 2X + p
                if (!launch_conv_pool_fwd_bf16(L[0].shape, a, k.P0, k.PI, CUR(c))) { net_err_set(c, "no fused conv+pool kernel for this conv shape"); return; }
            } else {
            if (b == 0) conv_fwd(c, L[0], nullptr, &src, 0, nullptr, k.C, n);
            else conv_fwd(c, L[0], prev, nullptr, 0, nullptr, k.C, n);
            { ProfScope ps(c, PC_POOL_FWD, n, (double)n * k.hin * k.hin * k.cout * (c->es * 1.25 + 0.25), 0.0);
              if (c->bf) launch_maxpool_fwd_bf16(k.C, k.P0, k.PI, n, k.hin, k.cout, CUR(c)); else launch_maxpool_fwd(k.C, k.P0, k.PI, n, k.hin, k.cout, CUR(c)); }
            }
            if (c->bf) {                     // res1 + res2 in ONE launch; intermediates reach HBM only when a backward pass follows
                const double px = (double)n * L[1].hw * L[1].hw, ch = L[1].cout;
                const float* bb[4] = {c->params + L[1].b_off, c->params + L[2].b_off, c->params + L[3].b_off, c->params + L[4].b_off};
                const unsigned short* bk[4] = {c->banks + L[1].bank_f, c->banks + L[2].bank_f, c->banks + L[3].bank_f, c->banks + L[4].bank_f};
                ProfScope ps(c, PC_RESBLOCK + (int)L[1].shape, n, px * ch * 2.0 * 10, 4.0 * px * 18.0 * ch * ch);      // 8(d): 2 blocks x (2 convs x 2p + skip p) = 10p (the kernel itself moves 5p)
                launch_resblock_pair_bf16(L[1].shape, k.P0, bb, train ? k.A1 : nullptr, train ? k.P1 : nullptr, train ? k.A2 : nullptr, k.P2, n, bk, CUR(c));
            } else {
                conv_fwd(c, L[1], k.P0, nullptr, 1, nullptr, k.A1, n);
                conv_fwd(c, L[2], k.A1, nullptr, 1, k.P0, k.P1, n);
                conv_fwd(c, L[3], k.P1, nullptr, 1, nullptr, k.A2, n);
                conv_fwd(c, L[4], k.A2, nullptr, 1, k.P1, k.P2, n);
            }
            prev = k.P2;
        }
        const float* last_p2 = prev;
        if (c->bf) {                            // bf16 matrix cores on the packed [256][2048] weight image (fc_bf16.hip)
            ProfScope ps(c, PC_GEMM, n, 2.0 * n * 2048 + 2.0 * 2048 * 256 + 4.0 * n * 256, 2.0 * n * 2048 * 256);
            if (n >= 1024) launch_fc_fwd_bf16(last_p2, c->fc_wp, c->params + c->fc.b_off, feat, n, CUR(c));
            else launch_fc_fwd_small_bf16(last_p2, c->fc_wp, c->params + c->fc.b_off, feat, n, CUR(c));   // rollout-sized: latency-bound
        } else
            linear_fwd(c, last_p2, 1, c->params + c->fc.w_off, c->params + c->fc.b_off, feat, n, 2048, c->H, 1, c->bf);
    } else {
        float* x0 = c->mlp_act[0] + (size_t)soff * c->cfg.obs_dim;
        launch_gather_rows((const float*)src.base, src.idx, src.first, x0, n, c->cfg.obs_dim, CUR(c));
        const size_t L = c->mlp.size();
        const float* x = x0;
        for (size_t l = 0; l < L; ++l) {
            float* y = (l + 1 == L) ? feat : c->mlp_act[l + 1] + (size_t)soff * c->mlp[l].out;
            linear_fwd(c, x, 0, c->params + c->mlp[l].w_off, c->params + c->mlp[l].b_off, y, n, c->mlp[l].in, c->mlp[l].out, l + 1 < L);
            x = y;
        }
    }
    if (recurrent && c->gru_on) net_gru(c, n, soff);
    if (with_heads) net_heads(c, n, soff);
}

#ifndef SIDE_EXT_EVENT
#define SIDE_EXT_EVENT 1
#endif
// backward from dY (n x (A+1)); gradients accumulate into c->grads
static void net_backward(mi_ctx* c, const InputSrc& src, int n) {
    const bool impala = c->cfg.arch == MI_ARCH_IMPALA;
    if (c->bwd_from_dfeat) {
        // value saliency of a recurrent policy: c->dfeat already holds d value / d (embedder output) (through the GRU cell, mi_value_saliency)
    } else if (c->H <= 256 && c->A + 1 <= 16 && !tl_ws) {        // one launch (+ its slab sum) for the heads' three gradients (misc.hip: heads_bwd_kernel)
        ProfScope ps(c, PC_GEMM, n, 4.0 * ((double)n * (c->A + 1) + 2.0 * n * c->H + (double)c->H * (c->A + 1)), 4.0 * n * c->H * (c->A + 1));
        launch_heads_bwd(c->dY, c->feat, c->params + c->wh_off, impala ? 1 : 0, c->dfeat, c->grads + c->wh_off, c->grads + c->bh_off, c->gemm_ws,
                         n, c->H, c->A + 1, CUR(c), !(c->side.armed && !tl_stream),      // (side stream armed: it also sums the slabs ...
                         (c->side.armed && !tl_stream && SIDE_EXT_EVENT) ? c->ev_side_fork : nullptr);      //  ... and forks on this launch's completion)
    } else {
        linear_wgrad(c, c->dY, c->feat, 0, c->grads + c->wh_off, c->grads + c->bh_off, n, c->H, c->A + 1);
        linear_dgrad(c, c->dY, c->params + c->wh_off, impala ? c->feat : nullptr, c->dfeat, n, c->H, c->A + 1);
    }
    if (!impala) {
        const size_t L = c->mlp.size();
        const float* dy = c->dfeat;
        for (size_t l = L; l-- > 0;) {
            linear_wgrad(c, dy, c->mlp_act[l], 0, c->grads + c->mlp[l].w_off, c->grads + c->mlp[l].b_off, n, c->mlp[l].in, c->mlp[l].out);
            if (l == 0) { c->sal_src = dy; break; }
            float* dx = c->GP[l & 1];
            linear_dgrad(c, dy, c->params + c->mlp[l].w_off, c->mlp_act[l], dx, n, c->mlp[l].in, c->mlp[l].out);
            dy = dx;
        }
        issue_grad_allreduce(c, 0, c->n_params, true);
        return;
    }
    const bool fc16 = c->bf && n >= 1024;
    bool side_forked = false;
    int slabs_done = 0;          // leading entries of the slab table already summed on the side stream
    bool fork2_on_launch = false;
    if (fc16 && c->side.armed && !tl_stream) {
        // fork: the side stream takes the logged statistics and embedder.fc's weight / bias gradients (mi_ctx::side_stream); this stream
        // goes straight on to the data gradient.  Every buffer the side work touches (block-3 output, feat, dfeat, loss partial sums,
        // the split-K / column-sum / metric workspaces, the fc + head slices of grads) is next written after the join below.
        fc_refresh(c);
        const mi_ctx::SideJob& j = c->side;
        // the first env group's stream when there is one (idle during an update, joined by JOIN(); one hardware queue less in use), else an own stream
        hipStream_t ss = (c->n_groups > 0 && c->gs[0]) ? c->gs[0] : c->side_stream;
        if (!ss) { hipStreamCreateWithFlags(&c->side_stream, hipStreamNonBlocking); ss = c->side_stream; }
        if (!SIDE_EXT_EVENT) hipEventRecord(c->ev_side_fork, c->stream);          // (else: heads_bwd_kernel's own completion, launch_heads_bwd above)
        hipStreamWaitEvent(ss, c->ev_side_fork, 0);
        if (c->idx_ev_deferred >= 0) { hipEventRecord(c->idx_ev[c->idx_ev_deferred], ss); c->idx_ev_deferred = -1; }      // (minibatch_impl: the index slot's "read" marker)
        tl_stream = ss;
        launch_heads_bwd_reduce(c->gemm_ws, c->grads + c->wh_off, c->grads + c->bh_off, n, c->H, c->A + 1, ss);      // (before fc_tn reuses the slabs)
        launch_fs_metric_seg(c->blk[2].P2, c->bf, j.st, 2048, c->fs_scratch, c->fs_parts, ss);
        launch_loss_finalize_seg(j.a, j.st, j.mode, j.ring, c->fs_parts, 2048, j.fsr, j.log, ss);
        launch_fc_tn(c->dfeat, (const unsigned short*)c->blk[2].P2, c->grads + c->fc.w_off, c->gemm_ws, (size_t)8 << 20, 256, 2048, n, ss);
        launch_colsum_acc(c->dfeat, n, 256, 256, c->grads + c->fc.b_off, c->col_ws, ss);
        tl_stream = nullptr;
        hipEventRecord(c->ev_side_join, ss);
        c->side.armed = false;
        side_forked = true;
    } else if (fc16) {
        fc_refresh(c);
        { ProfScope ps(c, PC_GEMM, n, 2.0 * n * 2048 + 4.0 * n * 256 + 4.0 * 2048 * 256, 2.0 * n * 2048 * 256);
          launch_fc_tn(c->dfeat, (const unsigned short*)c->blk[2].P2, c->grads + c->fc.w_off, c->gemm_ws, (size_t)8 << 20, 256, 2048, n, CUR(c)); }
        launch_colsum_acc(c->dfeat, n, 256, 256, c->grads + c->fc.b_off, c->col_ws, CUR(c));
    } else
        linear_wgrad(c, c->dfeat, c->blk[2].P2, 1, c->grads + c->fc.w_off, c->grads + c->fc.b_off, n, 2048, c->H, c->bf);
    issue_grad_allreduce(c, c->fc.w_off, c->n_params - c->fc.w_off, false);       // region A: fc + heads gradients are final
    float* Gout = c->GP[0];
    float* Ga = c->GP[1];
    float* Gb = c->GP[2];
    if (fc16) {
        ProfScope ps(c, PC_GEMM, n, 4.0 * n * 256 + 2.0 * 2048 * 256 + 2.0 * 2.0 * n * 2048, 2.0 * n * 2048 * 256);
        launch_fc_dgrad_bf16(c->dfeat, c->fc_wt, c->blk[2].P2, Gout, n, CUR(c));
    } else
        linear_dgrad(c, c->dfeat, c->params + c->fc.w_off, c->blk[2].P2, Gout, n, 2048, c->H, c->bf);
    if (c->fs_grad_coef != 0.f) {    // + fs_coef * d(feature sparsity) / d(block3 output): one element per column (launch_fs_grad, misc.hip)
        if (c->fs_global_apply) { if (n > 0) launch_fs_apply_keys(Gout, c->bf, 2048, c->fs_keys, c->fs_keys_local, c->fs_arg, c->fs_grad_coef, CUR(c)); }      // the column's winner among the ranks
        else launch_fs_grad(c->blk[2].P2, c->bf, n, 2048, c->fs_scratch, c->fs_G, Gout, c->fs_grad_coef, c->fs_colmax, c->fs_arg, CUR(c));
    }
    for (int b = 2; b >= 0; --b) {
        Block& k = c->blk[b];
        const ConvLayer* L = &c->convs[b * 5];
        // fused data gradients where they measured faster than two dgrad launches (16 channels @32x32: 14.5 vs 15.9 ms per
        // iteration; 32 @8x8: equal); at 32 channels @16x16 the two separate launches win (8.3 vs 10.2 ms)
        if (c->bf && (L[1].shape == CS_32_32_16 || L[1].shape == CS_32_32_8)) {
            // 32 channels @16x16: whole backward of each residual block in one launch (resblock_bwd_full32_bf16_kernel)
            const double px = (double)n * L[1].hw * L[1].hw, ch = L[1].cout;
            auto rb_full32 = [&](const ConvLayer& l1, const ConvLayer& l2, const float* dy, const float* a_fwd, const float* x_fwd, float* dx) {
                const int grid = resblock_bwd_full32_grid(l1.shape, n);
                const int i1 = (int)(&l1 - c->convs.data()), i2 = (int)(&l2 - c->convs.data());
                { ProfScope ps(c, PC_RESBLOCK_BWD + (int)l1.shape, n, px * ch * 2.0 * 7, 4.0 * px * 18.0 * ch * ch);      // 8(d): 2 convs x 3p + skip-gradient p = 7p (the kernel itself moves 4p)
                  launch_resblock_bwd_full32_bf16(l1.shape, dy, a_fwd, x_fwd, dx, nullptr, n, c->banks + l2.bank_d, c->banks + l1.bank_d,
                                                  c->slabs + c->slab_off[i2], c->slabs + c->slab_off[i1], CUR(c)); }
                const int wlen = l1.cout * 9 * l1.cin;
                c->h_slab_desc[c->slab_desc_n++] = SlabDesc{c->slab_off[i2], (long long)l2.w_off, (long long)l2.b_off, grid, wlen + l2.cout, wlen};
                c->h_slab_desc[c->slab_desc_n++] = SlabDesc{c->slab_off[i1], (long long)l1.w_off, (long long)l1.b_off, grid, wlen + l1.cout, wlen};
            };
            rb_full32(L[3], L[4], Gout, k.A2, k.P1, Gb);
            rb_full32(L[1], L[2], Gb, k.A1, k.P0, Gout);
        } else if (c->bf) {
            // both data gradients of a residual block in one launch (resblock_bf16.hip): the gradient of conv1's output goes
            // to HBM once (the weight-gradient kernels read it) and to LDS for the second transposed conv
            const double px = (double)n * L[1].hw * L[1].hw, ch = L[1].cout;
            auto rb_bwd = [&](const ConvLayer& l1, const ConvLayer& l2, const float* dy, const float* a_fwd, const float* x_fwd, float* da, float* dx) {
                ProfScope ps(c, PC_RESBLOCK_BWD + (int)l1.shape, n, px * ch * 2.0 * 5, 2.0 * px * 18.0 * ch * ch);
                launch_resblock_bwd_bf16(l1.shape, dy, a_fwd, x_fwd, da, dx, n, c->banks + l2.bank_d, c->banks + l1.bank_d, CUR(c));
            };
            if (L[1].shape == CS_16_16_32) {
                // 16 channels @32x32: data gradients AND both weight gradients in one launch (resblock_bwd_full_bf16_kernel);
                // the gradient of conv1's output never reaches HBM
                auto rb_full = [&](const ConvLayer& l1, const ConvLayer& l2, const float* dy, const float* a_fwd, const float* x_fwd, float* dx, hipEvent_t done_ev = nullptr) {
                    const int grid = resblock_bwd_full_grid(n);
                    const int i1 = (int)(&l1 - c->convs.data()), i2 = (int)(&l2 - c->convs.data());
                    { ProfScope ps(c, PC_RESBLOCK_BWD + (int)l1.shape, n, px * ch * 2.0 * 7, 4.0 * px * 18.0 * ch * ch);      // 8(d): 2 convs x 3p + skip-gradient p = 7p (the kernel itself moves 4p)
                      launch_resblock_bwd_full_bf16(dy, a_fwd, x_fwd, dx, nullptr, n, c->banks + l2.bank_d, c->banks + l1.bank_d,
                                                    c->slabs + c->slab_off[i2], c->slabs + c->slab_off[i1], CUR(c), done_ev); }
                    const int wlen = l1.cout * 9 * l1.cin;
                    c->h_slab_desc[c->slab_desc_n++] = SlabDesc{c->slab_off[i2], (long long)l2.w_off, (long long)l2.b_off, grid, wlen + l2.cout, wlen};
                    c->h_slab_desc[c->slab_desc_n++] = SlabDesc{c->slab_off[i1], (long long)l1.w_off, (long long)l1.b_off, grid, wlen + l1.cout, wlen};
                };
                rb_full(L[3], L[4], Gout, k.A2, k.P1, Gb);      // res2: P2 = conv2(relu(A2)) + P1 ; A2 = conv1(relu(P1))
                // (block 1's res1 is the last launch in front of the second fork: the fork event is this launch's own completion)
                fork2_on_launch = b == 0 && SIDE_EXT_EVENT && c->side_on && !tl_stream && n >= 1024 && c->slab_desc_cached_n == n && launch_resblock_bwd_full_event_ok();
                rb_full(L[1], L[2], Gb, k.A1, k.P0, Gout, fork2_on_launch ? c->ev_side_fork : nullptr);      // res1: P1 = conv2(relu(A1)) + P0 ; A1 = conv1(relu(P0))
            } else {
            // res2: P2 = conv2(relu(A2)) + P1 ; A2 = conv1(relu(P1))
            rb_bwd(L[3], L[4], Gout, k.A2, k.P1, Ga, Gb);
            conv_wgrad(c, L[4], k.A2, nullptr, 1, Gout, n);
            conv_wgrad(c, L[3], k.P1, nullptr, 1, Ga, n);
            // res1: P1 = conv2(relu(A1)) + P0 ; A1 = conv1(relu(P0))
            rb_bwd(L[1], L[2], Gb, k.A1, k.P0, Ga, Gout);
            conv_wgrad(c, L[2], k.A1, nullptr, 1, Gb, n);
            conv_wgrad(c, L[1], k.P0, nullptr, 1, Ga, n);
            }
        } else {
        // res2: P2 = conv2(relu(A2)) + P1 ; A2 = conv1(relu(P1))
        conv_wgrad(c, L[4], k.A2, nullptr, 1, Gout, n);
        conv_dgrad(c, L[4], Gout, k.A2, nullptr, Ga, n);
        conv_wgrad(c, L[3], k.P1, nullptr, 1, Ga, n);
        conv_dgrad(c, L[3], Ga, k.P1, Gout, Gb, n);
        // res1: P1 = conv2(relu(A1)) + P0 ; A1 = conv1(relu(P0))
        conv_wgrad(c, L[2], k.A1, nullptr, 1, Gb, n);
        conv_dgrad(c, L[2], Gb, k.A1, nullptr, Ga, n);
        conv_wgrad(c, L[1], k.P0, nullptr, 1, Ga, n);
        conv_dgrad(c, L[1], Ga, k.P0, Gb, Gout, n);
        }
        // max pool, then the block's first conv
        if (b == 0 && c->bf) {                 // pool backward fused into the staging
            c->sal_src = Gout;
            // second fork: block1.conv's weight gradient is the last kernel of the pass and nothing but its own slabs depends on it, so the
            // slab sums of the 14 layers before it run beside it (the table on the device is the cached one of this batch size: its last
            // entry is block1.conv's) and only that last entry is summed behind it
            if (c->side_on && !tl_stream && n >= 1024 && c->slab_desc_cached_n == n && c->slab_desc_n >= 1) {
                hipStream_t ss = (c->n_groups > 0 && c->gs[0]) ? c->gs[0] : c->side_stream;
                if (!ss) { hipStreamCreateWithFlags(&c->side_stream, hipStreamNonBlocking); ss = c->side_stream; }
                int max_len = 0;
                for (int q = 0; q < c->slab_desc_n; ++q) max_len = std::max(max_len, c->h_slab_desc[q].slab_len);
                if (!fork2_on_launch) hipEventRecord(c->ev_side_fork, c->stream);
                hipStreamWaitEvent(ss, c->ev_side_fork, 0);
                launch_reduce_all_slabs(c->slabs, c->grads, c->d_slab_desc, c->slab_desc_n, max_len, ss);
                hipEventRecord(c->ev_side_join, ss);
                side_forked = true;
                slabs_done = c->slab_desc_n;
            }
            conv_wgrad(c, L[0], nullptr, &src, 0, Gout, n, k.PI);
            break;
        }
        if (c->bf) {            // blocks 2, 3: both consumers of the conv-output gradient rebuild it from (pooled gradient, arg-max)
            const int fgrid = conv_bwd_fused_grid(L[0].shape, n);
            if (fgrid > 0) {        // block2.conv: data AND weight gradient in one launch (the max-pool backward gather runs once)
                const int layer = (int)(&L[0] - c->convs.data());
                ConvArgs a{};
                a.in = Gout; a.pool_arg = k.PI; a.out = Ga; a.n = n; a.bf16 = 1; a.wbank = c->banks + L[0].bank_d;
                a.wg_in = c->blk[b - 1].P2; a.wg_partial = c->slabs + c->slab_off[layer];
                const double px = (double)n * L[0].hw * L[0].hw;
                { ProfScope ps(c, PC_CONV_DGRAD + (int)L[0].shape, n, 2.0 * (px / 4 * L[0].cout + 3.0 * px * L[0].cout + 2.0 * px * L[0].cin), 2.0 * px * 18.0 * L[0].cin * L[0].cout);      // 8(d): pool bwd p + 2X, conv bwd X + 2I
                  launch_conv_dgrad(L[0].shape, a, CUR(c)); }
                const int wlen = L[0].cout * 9 * L[0].cin;
                c->h_slab_desc[c->slab_desc_n++] = SlabDesc{c->slab_off[layer], (long long)L[0].w_off, (long long)L[0].b_off, fgrid, wlen + L[0].cout, wlen};
            } else {
                conv_wgrad(c, L[0], c->blk[b - 1].P2, nullptr, 0, Gout, n, k.PI);
                conv_dgrad(c, L[0], Gout, nullptr, nullptr, Ga, n, k.PI);
            }
            std::swap(Gout, Ga);
            continue;
        }
        { ProfScope ps(c, PC_POOL_BWD, n, (double)n * k.hin * k.hin * k.cout * (c->es * 1.25 + 0.25), 0.0);
          if (c->bf) launch_maxpool_bwd_bf16(Gout, k.PI, c->GC, n, k.hin, k.cout, CUR(c)); else launch_maxpool_bwd(Gout, k.PI, c->GC, n, k.hin, k.cout, CUR(c)); }
        if (b == 0) { c->sal_src = c->GC; conv_wgrad(c, L[0], nullptr, &src, 0, c->GC, n); }
        else {
            conv_wgrad(c, L[0], c->blk[b - 1].P2, nullptr, 0, c->GC, n);
            conv_dgrad(c, L[0], c->GC, nullptr, nullptr, Gout, n);
        }
    }
    if (side_forked) hipStreamWaitEvent(c->stream, c->ev_side_join, 0);          // join: statistics, fc gradients (and the first slab sums) are in place behind this point
    conv_wgrad_reduce_all(c, n, slabs_done);
    issue_grad_allreduce(c, 0, c->fc.w_off, true);                                 // region B: the conv layers' gradients
}

// ------------------------------------------------------------------------------------------ predict / forward
int mi_policy_step(mi_ctx* c, int32_t t, uint64_t seed, const float* u, int64_t* act_out, float* logp_out, float* value_out) {
    ARG(c, "null"); JOIN(c); ARG(t >= 0 && t <= c->T, "t out of range");
    const int E = c->E;
    InputSrc src{c->frames ? (const void*)c->frames : (const void*)c->obsf, nullptr, (long long)t * E};
    const float* du = nullptr;
    if (u) { HIPC(hipMemcpyAsync(c->d_u, u, (size_t)E * 4, hipMemcpyHostToDevice, c->stream)); du = c->d_u; }
    c->prof.phase = 0;
    net_forward(c, src, E, true);
    const bool last = (t == c->T);
    launch_sample(c->hout, E, c->A, du, seed, (unsigned long long)t * E, last ? nullptr : c->act + (size_t)t * E,
                  last ? nullptr : c->logp + (size_t)t * E, c->value + (size_t)t * E, c->stream);
    HIPC(hipGetLastError()); NETCHK(c);
    if (!act_out && !logp_out && !value_out) { if (u) HIPC(hipStreamSynchronize(c->stream)); return 0; }
    if (act_out && !last) HIPC(hipMemcpyAsync(c->h_i, c->act + (size_t)t * E, (size_t)E * 4, hipMemcpyDeviceToHost, c->stream));
    if (logp_out && !last) HIPC(hipMemcpyAsync(c->h_f, c->logp + (size_t)t * E, (size_t)E * 4, hipMemcpyDeviceToHost, c->stream));
    if (value_out) HIPC(hipMemcpyAsync(c->h_f + E, c->value + (size_t)t * E, (size_t)E * 4, hipMemcpyDeviceToHost, c->stream));
    HIPC(hipStreamSynchronize(c->stream));
    if (act_out && !last) for (int e = 0; e < E; ++e) act_out[e] = c->h_i[e];
    if (logp_out && !last) memcpy(logp_out, c->h_f, (size_t)E * 4);
    if (value_out) memcpy(value_out, c->h_f + E, (size_t)E * 4);
    return 0;
}

int mi_rollout_step(mi_ctx* c, int32_t t, const float* rew_prev, const float* done_prev, uint64_t seed, const float* u,
                    int64_t* act_out, float* logp_out, float* value_out) {
    ARG(c, "null"); JOIN(c); ARG(t >= 0 && t <= c->T, "t out of range");
    const int E = c->E;
    if (rew_prev || done_prev) {
        ARG(rew_prev && done_prev && t >= 1, "rew_prev/done_prev come together and belong to step t-1");
        // pinned, device-visible staging: the head kernel reads {rew, done} straight from host memory and writes its
        // packed result straight back -- no copy kernels on the step's critical path (the stream sync below fences both)
        memcpy(c->h_rd, rew_prev, (size_t)E * 4); memcpy(c->h_rd + E, done_prev, (size_t)E * 4);
        if (c->gru_on) HIPC(hipMemcpyAsync(c->d_done, c->h_rd + E, (size_t)E * 4, hipMemcpyHostToDevice, c->stream));
    }
    const bool have_rd = rew_prev != nullptr;
    InputSrc src{c->frames ? (const void*)c->frames : (const void*)c->obsf, nullptr, (long long)t * E};
    const float* du = nullptr;
    if (u) { HIPC(hipMemcpyAsync(c->d_u, u, (size_t)E * 4, hipMemcpyHostToDevice, c->stream)); du = c->d_u; }
    c->prof.phase = 0;
    net_forward(c, src, E, true, false);
    const bool last = (t == c->T);
    launch_heads_sample(c->feat, c->params + c->wh_off, c->params + c->bh_off, E, c->H, c->A, du, seed, (unsigned long long)t * E,
                        last ? nullptr : c->act + (size_t)t * E, last ? nullptr : c->logp + (size_t)t * E, c->value + (size_t)t * E,
                        c->h_pack, nullptr, have_rd ? c->h_rd : nullptr, have_rd ? c->rew + (size_t)(t - 1) * E : nullptr,
                        have_rd ? c->done + (size_t)(t - 1) * E : nullptr, c->stream, c->d_done_ctr, c->h_flag, ++c->roll_ticket);
    HIPC(hipGetLastError()); NETCHK(c);
    // The last workgroup of the head kernel publishes the ticket after all results (h_pack) are visible to the host and all reads of
    // h_rd / u are done: spinning on it returns ~5 us earlier than hipStreamSynchronize (12.9 -> 8.1 us for launch + wait of a small
    // kernel, scratch/synclat.hip), 257 times per iteration.  A kernel that never finishes (fault) falls back to the stream wait,
    // which reports the error.
    {
        const unsigned want = c->roll_ticket;
        bool seen = false;
        for (unsigned long long spin = 0; spin < (1ull << 34); ++spin) {
            if (__atomic_load_n(c->h_flag, __ATOMIC_ACQUIRE) == want) { seen = true; break; }
            __builtin_ia32_pause();
            if ((spin & 0xfffff) == 0xfffff && hipStreamQuery(c->stream) != hipErrorNotReady) break;     // finished without a ticket, or failed
        }
        if (!seen) HIPC(hipStreamSynchronize(c->stream));
    }
    for (int e = 0; e < E; ++e) {
        if (act_out && !last) act_out[e] = (int64_t)c->h_pack[3 * e];
        if (logp_out && !last) logp_out[e] = c->h_pack[3 * e + 1];
        if (value_out) value_out[e] = c->h_pack[3 * e + 2];
    }
    return 0;
}

// ------------------------------------------------------------------------------------------ pipelined rollout (env groups)
// The reference's loop (agents/ppo.py:225-236) is strictly serial per step: obs -> H2D -> forward -> D2H act -> env.step.  An env's
// next frame depends only on its OWN action, so the E envs split into G contiguous groups whose chains
//     upload frames(t, g) -> forward + sample (t, g) -> actions on the host -> [env.step of group g on the host] -> upload frames(t+1, g)
// are independent: group g's upload (PCIe, ~37 us for 128 frames) and forward run on stream gs[g] while the host waits for / steps
// another group.  Per group: its own stream, rows [e0, e0 + E/G) of the activation buffers, its slice of the pinned hand-off
// buffers, its own completion ticket.  Numbers are those of mi_rollout_step (same kernels, same Philox counters t*E + e).
static void worker_drain(GroupWorker* w) {       // until the worker has issued everything that was posted to it
    if (!w) return;
    while (w->done.load(std::memory_order_acquire) != w->posted.load(std::memory_order_acquire)) __builtin_ia32_pause();
}
static int join_groups(mi_ctx* c) {
    for (int g = 0; g < c->n_groups; ++g) {
        if (!c->gs[g]) continue;
        worker_drain(c->gw[g]);
        if (c->g_dirty[g]) {
            HIPC(hipEventRecord(c->ev_join[g], c->gs[g]));
            HIPC(hipStreamWaitEvent(c->main_stream, c->ev_join[g], 0));
            c->g_dirty[g] = false;
        }
        c->g_forked[g] = false;          // the next submit of this group orders itself behind the main stream again
    }
    c->groups_live = false;
    return 0;
}

// the device half of a group step, on the group's worker thread (tl_stream = the group's stream)
static int group_issue(mi_ctx* c, int g, const GroupJob& j) {
    const int E = c->E, ng = E / c->n_groups, e0 = g * ng;
    hipStream_t st = tl_stream;
    char* ring = c->frames ? (char*)c->frames : (char*)c->obsf;
    if (j.frames) {
        char* dst = ring + ((size_t)j.t * E + e0) * c->obs_bytes_per_env;
        // One upload at a time.  Uploads of several groups issued together share the PCIe link and all finish late and TOGETHER: the
        // groups' chains then stay in lock-step and every step pays the shared-link copy time (two stable regimes were measured at
        // E = 256, G = 4: 107 and 135-138 us per policy step).  Each upload reserves the link for bytes / rate from the moment the previous
        // reservation ends (a few us of spinning on the worker thread, only when groups bunch), which puts the chains out of step again.
        if (c->copy_rate_bytes_per_us > 0) {
            const int64_t gap = (int64_t)((double)j.bytes * 1000.0 / c->copy_rate_bytes_per_us);
            const auto clk = [] { return (int64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
            int64_t slot = c->copy_slot_ns.load(), start;
            do { const int64_t now = clk(); start = now > slot ? now : slot; } while (!c->copy_slot_ns.compare_exchange_weak(slot, start + gap));
            while (clk() < start) __builtin_ia32_pause();
        }
        // page-locked, device-visible frames are PULLED by a kernel on the group's stream: a DMA copy in front of the first conv costs the
        // hand-over from the compute queue to the copy engine and back on top of the transfer (misc.hip pull_i32_kernel)
        if (j.pull) launch_pull_bytes(j.frames, dst, j.bytes, st);
        else HIPC(hipMemcpyAsync(dst, j.frames, j.bytes, hipMemcpyHostToDevice, st));
    }
    float* h_rd = c->h_rd + 2 * e0;      // this group's {rew[ng], done[ng]} (pinned, device-visible), filled by the submitting thread
    if (j.have_rd && c->gru_on) HIPC(hipMemcpyAsync(c->d_done + e0, h_rd + ng, (size_t)ng * 4, hipMemcpyHostToDevice, st));
    const float* du = nullptr;
    if (j.u) { HIPC(hipMemcpyAsync(c->d_u + e0, j.u, (size_t)ng * 4, hipMemcpyHostToDevice, st)); du = c->d_u + e0; }
    InputSrc src{c->frames ? (const void*)c->frames : (const void*)c->obsf, nullptr, (long long)j.t * E + e0};
    net_forward(c, src, ng, true, false, false, e0);
    const size_t o = (size_t)j.t * E + e0;
    launch_heads_sample(c->feat + (size_t)e0 * c->H, c->params + c->wh_off, c->params + c->bh_off, ng, c->H, c->A, du, j.seed, (unsigned long long)j.t * E + e0,
                        j.last ? nullptr : c->act + o, j.last ? nullptr : c->logp + o, c->value + o, c->h_pack + 3 * e0, nullptr,
                        j.have_rd ? h_rd : nullptr, j.have_rd ? c->rew + o - E : nullptr, j.have_rd ? c->done + o - E : nullptr, st,
                        c->d_done_ctr + 1 + g, c->h_flag + 1 + g, j.ticket);
    HIPC(hipGetLastError());
    if (const char* lf = mi_launch_failed_take()) return fail(-4, lf);      // (this worker thread's launchers)
    return 0;
}
static void group_worker_main(mi_ctx* c, int g) {
    GroupWorker* w = c->gw[g];
    hipSetDevice(c->cfg.device);
    tl_stream = c->gs[g];
    tl_ws_floats = c->gemm_ws_floats / mi_ctx::MAX_GROUPS; tl_ws = c->gemm_ws + (size_t)g * tl_ws_floats;     // concurrent groups: disjoint split-K slabs
    unsigned seen = 0;
    for (;;) {
        int spins = 0;
        while (w->posted.load(std::memory_order_acquire) == seen && !w->quit.load()) {
            if (++spins < 40000) __builtin_ia32_pause();
            else {                                   // idle for a few hundred us (update phase): sleep until the next post
                std::unique_lock<std::mutex> lk(w->mu);
                w->sleeping.store(true);
                w->cv.wait_for(lk, std::chrono::milliseconds(50), [&] { return w->posted.load() != seen || w->quit.load(); });
                w->sleeping.store(false);
                spins = 0;
            }
        }
        if (w->quit.load()) return;
        seen = w->posted.load(std::memory_order_acquire);
        w->rc = group_issue(c, g, w->job);
        if (w->rc) w->err = g_err;
        w->done.store(seen, std::memory_order_release);
    }
}

int mi_rollout_groups(mi_ctx* c, int32_t n_groups) {
    ARG(c, "null"); ARG(n_groups >= 1 && n_groups <= mi_ctx::MAX_GROUPS, "1 .. 4 groups");
    ARG(c->E % n_groups == 0, "n_envs must be divisible by the number of groups");
    for (int g = 0; g < c->n_groups; ++g) ARG(!c->g_busy[g], "a group step is still in flight: mi_rollout_wait it first");
    JOIN(c);
    for (int g = 0; g < n_groups; ++g)
        if (!c->gs[g]) {
            HIPC(hipStreamCreateWithFlags(&c->gs[g], hipStreamNonBlocking));
            HIPC(hipEventCreateWithFlags(&c->ev_fork[g], hipEventDisableTiming));
            HIPC(hipEventCreateWithFlags(&c->ev_join[g], hipEventDisableTiming));
            c->gw[g] = new GroupWorker();
            c->gw[g]->th = std::thread(group_worker_main, c, g);
        }
    c->n_groups = n_groups;
    return 0;
}

int mi_rollout_submit(mi_ctx* c, int32_t t, int32_t g, const void* frames, size_t bytes, const float* rew_prev, const float* done_prev,
                      uint64_t seed, const float* u) {
    ARG(c, "null"); ARG(t >= 0 && t <= c->T, "t out of range"); ARG(g >= 0 && g < c->n_groups, "group out of range");
    ARG(c->gs[g], "call mi_rollout_groups first"); ARG(!c->g_busy[g], "this group's previous step has not been waited for");
    ARG(c->pending_n < 0, "a multirank minibatch is pending");
    const int E = c->E, ng = E / c->n_groups, e0 = g * ng;
    ARG(!frames || bytes == (size_t)ng * c->obs_bytes_per_env, "frames byte count != (E / groups) * bytes_per_env");
    if (rew_prev || done_prev) ARG(rew_prev && done_prev && t >= 1, "rew_prev/done_prev come together and belong to step t-1");
    GroupWorker* w = c->gw[g];
    worker_drain(w);
    if (!c->g_forked[g]) {               // first step since the main stream last worked: parameters / packed banks must be in place
        fc_refresh(c);
        HIPC(hipEventRecord(c->ev_fork[g], c->main_stream));
        HIPC(hipStreamWaitEvent(c->gs[g], c->ev_fork[g], 0));
        c->g_forked[g] = true;
    }
    c->groups_live = true; c->g_dirty[g] = true;
    const bool have_rd = rew_prev != nullptr;
    if (have_rd) { float* h_rd = c->h_rd + 2 * e0; memcpy(h_rd, rew_prev, (size_t)ng * 4); memcpy(h_rd + ng, done_prev, (size_t)ng * 4); }
    const bool last = (t == c->T);
    bool pull = false;
    // Pull only what is latency-bound: measured per policy step, E = 64 in 2 groups (393 KB each) 73-76 us pulled vs 84-86 us copied,
    // E = 256 in 4 groups (786 KB each, four pulls competing) 121-135 vs 108-112 us -- shader reads of host memory move fewer bytes per
    // second than the copy engine, so above 512 KB the DMA's hand-over is the smaller price.
    if (frames && !c->no_pull && bytes <= (512u << 10) && (bytes & 15) == 0 && ((uintptr_t)frames & 15) == 0) {      // device-visible pinned memory? (asked once per buffer)
        auto it = c->pull_ok.find(frames);
        if (it == c->pull_ok.end()) {
            hipPointerAttribute_t at{};
            const bool ok = hipPointerGetAttributes(&at, frames) == hipSuccess && at.type == hipMemoryTypeHost && at.devicePointer == frames;
            if (!ok) (void)hipGetLastError();
            if (c->pull_ok.size() > 64) c->pull_ok.clear();
            it = c->pull_ok.emplace(frames, ok).first;
        }
        pull = it->second;
    }
    w->job = GroupJob{t, frames, bytes, pull, have_rd, last, u, seed, ++c->g_ticket[g]};
    w->posted.fetch_add(1, std::memory_order_seq_cst);
    if (w->sleeping.load()) { { std::lock_guard<std::mutex> lk(w->mu); } w->cv.notify_one(); }
    c->g_busy[g] = true; c->g_last[g] = last;
    return 0;
}

int mi_rollout_wait(mi_ctx* c, int32_t g, int64_t* act_out, float* logp_out, float* value_out) {
    ARG(c, "null"); ARG(g >= 0 && g < c->n_groups, "group out of range"); ARG(c->g_busy[g], "nothing submitted for this group");
    const int ng = c->E / c->n_groups, e0 = g * ng;
    const unsigned want = c->g_ticket[g];
    volatile unsigned* flag = c->h_flag + 1 + g;
    GroupWorker* w = c->gw[g];
    bool seen = false;
    for (unsigned long long spin = 0; spin < (1ull << 34); ++spin) {
        if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) == want) { seen = true; break; }
        __builtin_ia32_pause();
        if ((spin & 0xfffff) == 0xfffff && w->done.load() == w->posted.load() && hipStreamQuery(c->gs[g]) != hipErrorNotReady) break;   // issued, finished, no ticket: failed
    }
    worker_drain(w);
    c->g_busy[g] = false;
    if (w->rc) { const int rc = w->rc; w->rc = 0; return fail(rc, "group worker: " + w->err); }
    if (!seen) HIPC(hipStreamSynchronize(c->gs[g]));
    NETCHK(c);
    const float* pk = c->h_pack + 3 * e0;
    const bool last = c->g_last[g];
    for (int e = 0; e < ng; ++e) {
        if (act_out && !last) act_out[e] = (int64_t)pk[3 * e];
        if (logp_out && !last) logp_out[e] = pk[3 * e + 1];
        if (value_out) value_out[e] = pk[3 * e + 2];
    }
    return 0;
}

int mi_predict_staged(mi_ctx* c, const void* obs, size_t bytes, uint64_t seed, uint64_t counter, const float* u,
                      int64_t* act_out, float* logp_out, float* value_out) {
    ARG(c && obs, "null"); JOIN(c);
    const int E = c->E;
    ARG(bytes == (size_t)E * c->obs_bytes_per_env, "obs byte count != E * bytes_per_env");
    void* stage = c->stage_frames ? (void*)c->stage_frames : (void*)c->stage_obs;
    HIPC(hipMemcpyAsync(stage, obs, bytes, hipMemcpyHostToDevice, c->stream));
    const float* du = nullptr;
    if (u) { HIPC(hipMemcpyAsync(c->d_u, u, (size_t)E * 4, hipMemcpyHostToDevice, c->stream)); du = c->d_u; }
    InputSrc src{stage, nullptr, 0};
    net_forward(c, src, E, true);
    launch_sample(c->hout, E, c->A, du, seed, counter, c->s_act, c->s_logp, c->s_val, c->stream);
    HIPC(hipGetLastError()); NETCHK(c);
    HIPC(hipMemcpyAsync(c->h_i, c->s_act, (size_t)E * 4, hipMemcpyDeviceToHost, c->stream));
    HIPC(hipMemcpyAsync(c->h_f, c->s_logp, (size_t)E * 4, hipMemcpyDeviceToHost, c->stream));
    HIPC(hipMemcpyAsync(c->h_f + E, c->s_val, (size_t)E * 4, hipMemcpyDeviceToHost, c->stream));
    HIPC(hipStreamSynchronize(c->stream));
    c->staged_valid = true;
    if (act_out) for (int e = 0; e < E; ++e) act_out[e] = c->h_i[e];
    if (logp_out) memcpy(logp_out, c->h_f, (size_t)E * 4);
    if (value_out) memcpy(value_out, c->h_f + E, (size_t)E * 4);
    return 0;
}

// PPO.predict_w_value_saliency (agents/ppo.py:83-94): predict + d value / d observation.  grad_out: IMPALA [E][64][64][3] (NHWC,
// wrt the k/255 float frames), MLP [E][obs_dim].  Runs the training-mode forward and the whole backward pass with dY = e_value on
// the E staged observations; the parameter gradients it produces on the way are discarded (the gradient buffer is zeroed again),
// so it must not be called between mi_minibatch and mi_optimizer_step of an accumulating update.
int mi_value_saliency(mi_ctx* c, const void* obs, size_t bytes, uint64_t seed, uint64_t counter, const float* u,
                      int64_t* act_out, float* logp_out, float* value_out, float* grad_out) {
    ARG(c && obs && grad_out, "null"); JOIN(c);
    ARG(c->pending_n < 0, "a multirank minibatch is pending");
    const int E = c->E;
    const bool impala = c->cfg.arch == MI_ARCH_IMPALA;
    ARG(bytes == (size_t)E * c->obs_bytes_per_env, "obs byte count != E * bytes_per_env");
    void* stage = c->stage_frames ? (void*)c->stage_frames : (void*)c->stage_obs;
    HIPC(hipMemcpyAsync(stage, obs, bytes, hipMemcpyHostToDevice, c->stream));
    const float* du = nullptr;
    if (u) { HIPC(hipMemcpyAsync(c->d_u, u, (size_t)E * 4, hipMemcpyHostToDevice, c->stream)); du = c->d_u; }
    InputSrc src{stage, nullptr, 0};
    c->prof.phase = 0;
    const bool rec = c->gru_on;
    if (rec && !c->gru_x) { HIPC(dalloc(&c->gru_x, (size_t)E * c->H)); HIPC(dalloc(&c->gru_dg, (size_t)E * 3 * c->H)); }
    c->sal_keep_x = rec;
    net_forward(c, src, E, rec, true, true);          // recurrent: h' = GRU(embedder output, h (1 - done)) as a policy step does, heads on h'
    c->sal_keep_x = false;
    launch_sample(c->hout, E, c->A, du, seed, counter, c->s_act, c->s_logp, c->s_val, c->stream);
    c->sal_src = nullptr;
    if (rec) {
        // value = w_v . h' + b_v: back through the GRU cell to its input x (common/model.py:219-225 under autograd, agents/ppo.py:88-89), then
        // through the embedder's final ReLU (IMPALA) -- d value / d x = dgates W_ih -- and on down the usual backward pass from dfeat
        launch_gru_value_bwd(c->gru_gi, c->gru_gh, c->h_masked, c->params + c->wh_off + (size_t)c->A * c->H, c->gru_dg, E, c->H, c->stream);
        linear_dgrad(c, c->gru_dg, c->gru_wih, impala ? c->gru_x : nullptr, c->dfeat, E, c->H, 3 * c->H);
        c->bwd_from_dfeat = true;
        net_backward(c, src, E);
        c->bwd_from_dfeat = false;
    } else {
        launch_value_seed(c->dY, E, c->A, c->stream);
        net_backward(c, src, E);
    }
    ARG(c->sal_src, "backward did not reach the first layer");
    const size_t gfloats = impala ? (size_t)E * 64 * 64 * 3 : (size_t)E * c->cfg.obs_dim;
    if (!c->sal_dx) HIPC(dalloc(&c->sal_dx, impala ? (size_t)c->NB * 64 * 64 * 3 : (size_t)c->NB * c->cfg.obs_dim));
    if (impala) {
        const void* dC = c->sal_src;
        if (c->bf) {                                       // the conv-output gradient is not materialised in bf16 mode: rebuild it from the pooled one
            if (!c->sal_dc) HIPC(hipMalloc(&c->sal_dc, (size_t)c->NB * 64 * 64 * 16 * 2 + 256));
            launch_maxpool_bwd_bf16(c->sal_src, c->blk[0].PI, c->sal_dc, E, 64, 16, c->stream);
            dC = c->sal_dc;
        }
        launch_conv1_input_grad(dC, c->bf, c->params + c->convs[0].w_off, c->sal_dx, E, c->stream);
    } else {
        linear_dgrad(c, c->sal_src, c->params + c->mlp[0].w_off, nullptr, c->sal_dx, E, c->mlp[0].in, c->mlp[0].out);
    }
    launch_fill(c->grads, c->n_params, 0.f, c->stream);    // discard the parameter gradients of this pass
    HIPC(hipGetLastError()); NETCHK(c);
    HIPC(hipMemcpyAsync(c->h_i, c->s_act, (size_t)E * 4, hipMemcpyDeviceToHost, c->stream));
    HIPC(hipMemcpyAsync(c->h_f, c->s_logp, (size_t)E * 4, hipMemcpyDeviceToHost, c->stream));
    HIPC(hipMemcpyAsync(c->h_f + E, c->s_val, (size_t)E * 4, hipMemcpyDeviceToHost, c->stream));
    HIPC(hipMemcpyAsync(grad_out, c->sal_dx, gfloats * 4, hipMemcpyDeviceToHost, c->stream));
    HIPC(hipStreamSynchronize(c->stream));
    c->staged_valid = true;
    if (act_out) for (int e = 0; e < E; ++e) act_out[e] = c->h_i[e];
    if (logp_out) memcpy(logp_out, c->h_f, (size_t)E * 4);
    if (value_out) memcpy(value_out, c->h_f + E, (size_t)E * 4);
    return 0;
}

int mi_commit_staged(mi_ctx* c, int32_t t) {
    ARG(c, "null"); JOIN(c); ARG(t >= 0 && t <= c->T, "t out of range"); ARG(c->staged_valid, "nothing staged: call mi_predict_staged first");
    const size_t E = c->E, ob = E * c->obs_bytes_per_env;
    char* ring = c->frames ? (char*)c->frames : (char*)c->obsf;
    const void* stage = c->stage_frames ? (const void*)c->stage_frames : (const void*)c->stage_obs;
    HIPC(hipMemcpyAsync(ring + (size_t)t * ob, stage, ob, hipMemcpyDeviceToDevice, c->stream));
    if (t < c->T) {
        HIPC(hipMemcpyAsync(c->act + t * E, c->s_act, E * 4, hipMemcpyDeviceToDevice, c->stream));
        HIPC(hipMemcpyAsync(c->logp + t * E, c->s_logp, E * 4, hipMemcpyDeviceToDevice, c->stream));
    }
    HIPC(hipMemcpyAsync(c->value + t * E, c->s_val, E * 4, hipMemcpyDeviceToDevice, c->stream));
    return 0;
}

int mi_set_gru(mi_ctx* c, const float* w_ih, const float* w_hh, const float* b_ih, const float* b_hh) {
    ARG(c && w_ih && w_hh && b_ih && b_hh, "null"); JOIN(c);
    const size_t H = c->H, E = c->E;
    if (!c->gru_wih) {
        HIPC(dalloc(&c->gru_wih, 3 * H * H)); HIPC(dalloc(&c->gru_whh, 3 * H * H)); HIPC(dalloc(&c->gru_bih, 3 * H)); HIPC(dalloc(&c->gru_bhh, 3 * H));
        HIPC(dalloc(&c->h_state, E * H)); HIPC(dalloc(&c->h_masked, E * H)); HIPC(dalloc(&c->gru_gi, E * 3 * H)); HIPC(dalloc(&c->gru_gh, E * 3 * H));
        HIPC(dalloc(&c->d_done, E));
    }
    HIPC(hipMemcpy(c->gru_wih, w_ih, 3 * H * H * 4, hipMemcpyHostToDevice)); HIPC(hipMemcpy(c->gru_whh, w_hh, 3 * H * H * 4, hipMemcpyHostToDevice));
    HIPC(hipMemcpy(c->gru_bih, b_ih, 3 * H * 4, hipMemcpyHostToDevice)); HIPC(hipMemcpy(c->gru_bhh, b_hh, 3 * H * 4, hipMemcpyHostToDevice));
    c->gru_on = true;
    return 0;
}
int mi_rec_state(mi_ctx* c, const float* hidden, const float* done) {
    ARG(c, "null"); JOIN(c); ARG(c->gru_on, "no GRU set: call mi_set_gru first");
    const size_t H = c->H, E = c->E;
    if (hidden) HIPC(hipMemcpyAsync(c->h_state, hidden, E * H * 4, hipMemcpyHostToDevice, c->stream));
    if (done) HIPC(hipMemcpyAsync(c->d_done, done, E * 4, hipMemcpyHostToDevice, c->stream));
    else HIPC(hipMemsetAsync(c->d_done, 0, E * 4, c->stream));
    HIPC(hipStreamSynchronize(c->stream));
    return 0;
}
int mi_get_hidden(mi_ctx* c, float* hidden) {
    ARG(c && hidden, "null"); JOIN(c); ARG(c->gru_on, "no GRU set");
    HIPC(hipMemcpyAsync(hidden, c->h_state, (size_t)c->E * c->H * 4, hipMemcpyDeviceToHost, c->stream));
    HIPC(hipStreamSynchronize(c->stream));
    return 0;
}

static int forward_common(mi_ctx* c, const void* obs, int32_t n, bool recurrent, float* logp_all, float* value, float* feat);
int mi_forward_rec(mi_ctx* c, const void* obs, float* logp_all, float* value, float* hidden_out) {
    ARG(c && obs, "null"); ARG(c->gru_on, "no GRU set");
    int r = forward_common(c, obs, c->E, true, logp_all, value, hidden_out);   // feat == h' after the GRU
    return r;
}
int mi_forward(mi_ctx* c, const void* obs, int32_t n, float* logp_all, float* value, float* feat) {
    return forward_common(c, obs, n, false, logp_all, value, feat);
}
static int forward_common(mi_ctx* c, const void* obs, int32_t n, bool recurrent, float* logp_all, float* value, float* feat) {
    ARG(c && obs, "null"); JOIN(c); ARG(n >= 1 && n <= c->NB, "n must be in [1, max_batch]");
    c->staged_valid = false;
    void* stage = c->stage_frames ? (void*)c->stage_frames : (void*)c->stage_obs;
    HIPC(hipMemcpyAsync(stage, obs, (size_t)n * c->obs_bytes_per_env, hipMemcpyHostToDevice, c->stream));
    InputSrc src{stage, nullptr, 0};
    net_forward(c, src, n, recurrent);
    launch_logp_all(c->hout, n, c->A, c->d_lp, nullptr, c->stream);
    HIPC(hipGetLastError()); NETCHK(c);
    if (logp_all) HIPC(hipMemcpyAsync(logp_all, c->d_lp, (size_t)n * c->A * 4, hipMemcpyDeviceToHost, c->stream));
    if (feat) HIPC(hipMemcpyAsync(feat, c->feat, (size_t)n * c->H * 4, hipMemcpyDeviceToHost, c->stream));
    std::vector<float> h;
    if (value) { h.resize((size_t)n * (c->A + 1)); HIPC(hipMemcpyAsync(h.data(), c->hout, h.size() * 4, hipMemcpyDeviceToHost, c->stream)); }
    HIPC(hipStreamSynchronize(c->stream));
    if (value) for (int k = 0; k < n; ++k) value[k] = h[(size_t)k * (c->A + 1) + c->A];
    return 0;
}

// ------------------------------------------------------------------------------------------ estimates
int mi_compute_estimates(mi_ctx* c, float gamma, float lmbda, int32_t use_gae, int32_t normalize_adv) {
    ARG(c, "null"); JOIN(c);
    launch_gae(c->rew, c->done, c->value, c->adv, c->ret, c->T, c->E, gamma, lmbda, use_gae, c->stream);
    if (normalize_adv) {
        launch_advnorm_stats(c->adv, c->T * c->E, c->adv_stats, c->stream);
        launch_advnorm_apply(c->adv, c->T * c->E, c->adv_stats, c->stream);
    }
    HIPC(hipGetLastError()); NETCHK(c);
    return 0;
}
int mi_adv_stats(mi_ctx* c, double s[3]) {
    ARG(c && s, "null"); JOIN(c);
    launch_advnorm_stats(c->adv, c->T * c->E, c->adv_stats, c->stream);
    HIPC(hipMemcpyAsync(s, c->adv_stats, 3 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPC(hipStreamSynchronize(c->stream));
    return 0;
}
int mi_adv_apply(mi_ctx* c, const double s[3]) {
    ARG(c && s, "null"); JOIN(c);
    HIPC(hipMemcpyAsync(c->adv_stats, s, 3 * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIPC(hipStreamSynchronize(c->stream));
    launch_advnorm_apply(c->adv, c->T * c->E, c->adv_stats, c->stream);
    HIPC(hipGetLastError()); NETCHK(c);
    return 0;
}

// ------------------------------------------------------------------------------------------ minibatch / optimiser
static InputSrc minibatch_src(mi_ctx* c) {
    return InputSrc{c->frames ? (const void*)c->frames : (const void*)c->obsf, c->d_idx, 0};
}

// One forward + backward pass over n gathered samples that belong to n_seg GLOBAL minibatches (segment k = seg_n[k] consecutive
// entries of idx, every segment a global minibatch of n_global samples): per-sample gradients scale with 1 / n_global, the loss
// statistics are taken and logged per segment.  n_seg > 1 is gradient accumulation done in one launch set (agents/ppo.py:170-177
// sums the gradients of the accumulated minibatches before the optimizer step, so only the fp32 summation order changes); it
// needs a loss without batch-level terms (x_entropy_coef == 0, fs_coef == 0).
static bool side_eligible(const mi_ctx* c, int n, bool batch_terms) {
    return c->side_on && c->cfg.arch == MI_ARCH_IMPALA && c->bf && n >= 1024 && !batch_terms && !c->ar_armed && !c->comm && !c->bwd_from_dfeat &&
           c->H <= 256 && c->A + 1 <= 16;
}
static int minibatch_impl(mi_ctx* c, const int64_t* idx, int32_t n, const int32_t* seg_n, int32_t n_seg, int32_t n_global, const mi_hparams* hp) {
    ARG(c && hp, "null"); JOIN(c); ARG(n >= 0 && n <= c->NB, "n_idx must be in [0, max_batch]"); ARG(n_global >= 1, "n_global");
    ARG(n == 0 || idx, "idx"); ARG(n_seg >= 1 && n_seg <= MI_MAX_SEG && seg_n, "1 .. 16 segments");
    ARG(c->log_count + n_seg <= c->log_cap, "loss log full: call mi_loss_log_read(reset=1)");
    ARG(c->pending_n < 0, "previous multirank minibatch not finished");
    { long long tot = 0; for (int k = 0; k < n_seg; ++k) { ARG(seg_n[k] >= 0, "negative segment"); tot += seg_n[k]; } ARG(tot == n, "segments do not add up to n_idx"); }
    const bool batch_terms = hp->x_entropy_coef != 0.f || hp->fs_coef != 0.f;
    ARG(hp->fs_coef == 0.f || c->multirank != 2, "fs_coef != 0 needs the column maxima over the GLOBAL minibatch before the backward pass: multirank mode 1");
    ARG(hp->fs_coef == 0.f || c->multirank == 0 || c->cfg.arch != MI_ARCH_IMPALA || c->gpos_n == n,
        "fs_coef != 0 on several ranks: call mi_minibatch_positions with this pass's global minibatch positions first");
    ARG(n_seg == 1 || (!batch_terms && c->multirank != 1), "several minibatches per call need x_entropy_coef == 0, fs_coef == 0 and multirank mode 0 or 2");
    const int64_t TE = (int64_t)c->T * c->E;
    for (int k = 0; k < n; ++k) ARG(idx[k] >= 0 && idx[k] < TE, "minibatch index out of range");
    if (n) {
        const int slot = c->idx_next;
        c->idx_next = (slot + 1) % mi_ctx::IDX_RING;
        if (c->idx_used[slot]) HIPC(hipEventSynchronize(c->idx_ev[slot]));     // the DMA that read this slot is done
        int32_t* h = c->h_idx_ring[slot];
        for (int k = 0; k < n; ++k) h[k] = (int32_t)idx[k];
        launch_pull_i32(h, c->d_idx, n, c->stream);              // (not hipMemcpyAsync: see pull_i32_kernel)
        // "slot read" marker: an event record between the pull and the first conv kernel is a ~5 us bubble on the main stream (kernel
        // traces: 5-7 us between pull_i32 and repack_all); when this pass forks the side stream, the marker is recorded THERE, behind
        // the fork (which is behind the pull in main-stream order)
        if (side_eligible(c, n, batch_terms) && c->multirank != 1) c->idx_ev_deferred = slot;
        else HIPC(hipEventRecord(c->idx_ev[slot], c->stream));
        c->idx_used[slot] = true;
    }
    InputSrc src = minibatch_src(c);
    c->prof.phase = 1;
    c->prof.sample_now = (c->prof.mb_count++ % c->prof.period) == 0;
    net_forward(c, src, n, false, true, true);
    const bool impala = c->cfg.arch == MI_ARCH_IMPALA;
    LossArgs a{};
    a.hout = c->hout; a.idx = c->d_idx; a.act = c->act; a.old_logp = c->logp; a.old_value = c->value; a.ret = c->ret; a.adv = c->adv;
    a.dY = c->dY; a.partial = c->loss_partial; a.stats = c->loss_stats; a.n = n; a.A = c->A;
    a.inv_n_global = 1.0f / (float)n_global;
    a.hp = LossHP{hp->eps_clip, hp->value_coef, hp->entropy_coef, hp->x_entropy_coef, hp->entropy_multiplier, hp->fs_coef};
    if (c->multirank == 2)
        // deferred statistics: nothing in the backward pass needs the cross-rank sums when x_entropy_coef == 0 and fs_coef == 0, so this
        // rank's partial sums go to ring slot log_count and are summed over the ranks ONCE per optimize() (mi_loss_log_finalize)
        ARG(!batch_terms, "multirank mode 2 needs x_entropy_coef == 0 and fs_coef == 0 (use mode 1)");
    if (c->multirank != 1) {
        // modes 0 and 2: loss terms + logged statistics of all segments with one launch each.  The rank-local sums land in the
        // statistics ring (mode 0 derives the records right away, mode 2 after the cross-rank sum in mi_loss_log_finalize); without
        // batch-level terms the sample gradients dY come out of the same pass.
        SegTab st{};
        st.n_seg = n_seg;
        for (int k = 0; k < n_seg; ++k) st.start[k + 1] = st.start[k] + seg_n[k];
        float* ring = c->stats_ring + (size_t)c->log_count * 32;
        float* fsr = c->fs_ring + c->log_count;
        a.stats = ring;                                  // (x-entropy gradient, mode 0, n_seg == 1: the batch-mean action distribution)
        // no batch-level loss terms, bf16 IMPALA at update size, gradients exchanged (if at all) behind the pass: metric + records leave the
        // critical path (net_backward forks); the in-library armed exchange hands region A over in the middle of the pass and keeps the old order
        const bool side = side_eligible(c, n, batch_terms);
        if (side) {
            launch_loss_fwd_seg(a, st, true, c->stream);
            c->side = mi_ctx::SideJob{true, a, st, c->multirank == 2 ? 1 : 3, ring, fsr, c->multirank == 2 ? nullptr : c->loss_log + (size_t)c->log_count * 8};
        } else {
        if (impala) launch_fs_metric_seg(c->blk[2].P2, c->bf, st, 2048, c->fs_scratch, c->fs_parts, c->stream);
        launch_loss_fwd_seg(a, st, !batch_terms, c->stream);
        launch_loss_finalize_seg(a, st, c->multirank == 2 ? 1 : 3, ring, impala ? c->fs_parts : nullptr, 2048, fsr,
                                 c->multirank == 2 ? nullptr : c->loss_log + (size_t)c->log_count * 8, c->stream);
        }
        c->ring_args = a;
        c->log_count += n_seg;
        if (batch_terms) launch_loss_bwd(a, c->stream);
        if (impala && hp->fs_coef != 0.f && n > 0) { c->fs_grad_coef = hp->fs_coef; c->fs_G = fs_groups_per_segment(n_seg); }
        net_backward(c, src, n);
        c->fs_grad_coef = 0.f;
        HIPC(hipGetLastError()); NETCHK(c);
        return 0;
    }
    // mode 1: the cross-rank sum of the statistics comes between the loss forward and backward (mi_minibatch_finish)
    if (impala) launch_fs_metric(c->blk[2].P2, c->bf, n, 2048, c->fs_scratch, c->fs_val, c->stream);
    c->fs_global_pending = impala && hp->fs_coef != 0.f;
    if (c->fs_global_pending) {      // this rank's per-column candidates; the caller max-all-reduces MI_PTR_FS_KEYS before mi_minibatch_finish
        if (n > 0) launch_fs_keys(c->blk[2].P2, c->bf, n, 2048, c->fs_scratch, fs_metric_groups(), c->d_gpos, c->fs_colmax, c->fs_arg, c->fs_keys, c->fs_keys_local, c->stream);
        else { HIPC(hipMemsetAsync(c->fs_keys, 0, 2048 * 8, c->stream)); HIPC(hipMemsetAsync(c->fs_keys_local, 0, 2048 * 8, c->stream)); }
    }
    c->gpos_n = -1;
    launch_loss_fwd(a, c->stream);
    launch_loss_finalize(a, loss_blocks(n), 1, nullptr, nullptr, c->stream);
    c->pending = a; c->pending_n = n;
    HIPC(hipGetLastError()); NETCHK(c);
    return 0;
}

int mi_minibatch(mi_ctx* c, const int64_t* idx, int32_t n, int32_t n_global, const mi_hparams* hp) {
    return minibatch_impl(c, idx, n, &n, 1, n_global, hp);
}
int mi_minibatch_multi(mi_ctx* c, const int64_t* idx, int32_t n, const int32_t* seg_n, int32_t n_seg, int32_t n_global, const mi_hparams* hp) {
    return minibatch_impl(c, idx, n, seg_n, n_seg, n_global, hp);
}

// Global minibatch positions of the NEXT mi_minibatch's rows (ascending; the rank's share of a global minibatch keeps the global order):
// needed when fs_coef != 0 on more than one rank -- ties between equal column maxima go to the row that comes first globally.
int mi_minibatch_positions(mi_ctx* c, const int32_t* gpos, int32_t n) {
    ARG(c && (gpos || n == 0), "null"); JOIN(c); ARG(c->cfg.arch == MI_ARCH_IMPALA && c->d_gpos, "IMPALA contexts only"); ARG(n >= 0 && n <= c->NB, "n");
    if (n > 0) {
        HIPC(hipStreamSynchronize(c->stream));                 // (the pinned staging buffer of the previous call is free; this path is not the fast one)
        memcpy(c->h_gpos, gpos, (size_t)n * 4);
        HIPC(hipMemcpyAsync(c->d_gpos, c->h_gpos, (size_t)n * 4, hipMemcpyHostToDevice, c->stream));
    }
    c->gpos_n = n;
    return 0;
}
int mi_set_multirank(mi_ctx* c, int32_t enabled) { ARG(c, "null"); ARG(enabled >= 0 && enabled <= 2, "mode"); c->multirank = enabled; return 0; }

// multirank mode 2: after the caller summed stats_ring[0 .. log_count*32) over the ranks, derive every minibatch's log record
int mi_loss_log_finalize(mi_ctx* c) {
    ARG(c, "null"); JOIN(c); ARG(c->multirank == 2, "only in multirank mode 2");
    const bool impala = c->cfg.arch == MI_ARCH_IMPALA;
    launch_loss_finalize_records(c->ring_args, c->log_count, c->stats_ring, impala ? c->fs_ring : nullptr, c->loss_log, c->stream);
    HIPC(hipGetLastError()); NETCHK(c);
    return 0;
}

int mi_minibatch_finish(mi_ctx* c) {
    ARG(c, "null"); JOIN(c); ARG(c->pending_n >= 0, "no pending minibatch");
    const bool impala = c->cfg.arch == MI_ARCH_IMPALA;
    float* slot = c->loss_log + (size_t)c->log_count * 8;
    if (c->fs_global_pending) launch_fs_from_keys(c->fs_keys, 2048, c->fs_val, c->stream);      // the metric of the GLOBAL minibatch (keys are all-reduced by now)
    launch_loss_finalize(c->pending, 0, 2, impala ? c->fs_val : nullptr, slot, c->stream);
    c->log_count++;
    launch_loss_bwd(c->pending, c->stream);
    InputSrc src = minibatch_src(c);
    if (c->fs_global_pending) { c->fs_grad_coef = c->pending.hp.fs_coef; c->fs_global_apply = true; }
    net_backward(c, src, c->pending_n);
    c->fs_grad_coef = 0.f; c->fs_global_apply = false; c->fs_global_pending = false;
    c->pending_n = -1;
    HIPC(hipGetLastError()); NETCHK(c);
    return 0;
}

int mi_optimizer_step(mi_ctx* c, float lr, float max_norm, int32_t step, float* gnorm_out) {
    ARG(c, "null"); JOIN(c); ARG(step >= 1, "adam_step is 1-based");
    const double b1 = 0.9, b2 = 0.999;
    const double bc1 = 1.0 - pow(b1, (double)step), bc2 = 1.0 - pow(b2, (double)step);
    const float step_size = (float)((double)lr / bc1), bc2_sqrt = (float)sqrt(bc2);
    if (c->ar_inflight) { HIPC(hipStreamWaitEvent(c->stream, c->ev_ar_done, 0)); c->ar_inflight = false; }
    c->ar_issued = false; c->ar_armed = false;
    launch_sumsq_partials(c->grads, c->n_params, c->sumsq + 2, c->stream);          // 128 partial sums; the Adam kernel's waves add them up themselves
    launch_adam(c->params, c->grads, c->adam_m, c->adam_v, c->n_params, c->sumsq + 2, 128, max_norm, lr, (float)b1, (float)b2, 1e-5f,
                step_size, bc2_sqrt, c->gnorm, c->stream);
    c->fc_packed_valid = false;
    HIPC(hipGetLastError()); NETCHK(c);
    if (gnorm_out) {
        HIPC(hipMemcpyAsync(c->h_f, c->gnorm, 4, hipMemcpyDeviceToHost, c->stream));
        HIPC(hipStreamSynchronize(c->stream));
        *gnorm_out = c->h_f[0];
    }
    return 0;
}

int mi_loss_log_read(mi_ctx* c, float* out, int32_t max_records, int32_t* n_records, int32_t reset) {
    ARG(c && n_records, "null"); JOIN(c);
    const int n = c->log_count < max_records ? c->log_count : max_records;
    if (out && n > 0) {
        HIPC(hipMemcpyAsync(out, c->loss_log, (size_t)n * 8 * 4, hipMemcpyDeviceToHost, c->stream));
        HIPC(hipStreamSynchronize(c->stream));
    }
    *n_records = n;
    if (reset) c->log_count = 0;
    return 0;
}

// ------------------------------------------------------------------------------------------ collectives (RCCL over xGMI)
int mi_comm_unique_id(void* out, size_t bytes) {
    ARG(out && bytes >= sizeof(ncclUniqueId), "need a 128-byte buffer");
    ncclUniqueId id;
    NCCLC(ncclGetUniqueId(&id));
    memcpy(out, &id, sizeof id);
    return 0;
}
int mi_comm_init(mi_ctx* c, const void* id_bytes, size_t bytes, int32_t rank, int32_t world) {
    ARG(c && id_bytes && bytes >= sizeof(ncclUniqueId), "null / short id"); ARG(world >= 1 && rank >= 0 && rank < world, "rank / world");
    ARG(!c->comm, "communicator already initialised");
    JOIN(c);
    HIPC(hipSetDevice(c->cfg.device));
    ncclUniqueId id;
    memcpy(&id, id_bytes, sizeof id);
    NCCLC(ncclCommInitRank(&c->comm, world, id, rank));
    // The gradient regions travel on a side stream while the main stream may issue its own collectives (loss statistics, feature-
    // sparsity keys, advantage statistics).  One communicator driven from two streams has no defined order between the two streams'
    // operations; each stream therefore owns a communicator.  Both are used in the same host order on every rank (the update schedule
    // is rank-uniform: mi355/dist.py update_plan), which is what RCCL needs of two communicators on one device.
    NCCLC(ncclCommSplit(c->comm, 0, rank, &c->comm_grad, nullptr));
    c->comm_world = world; c->comm_rank = rank;
    HIPC(hipStreamCreateWithFlags(&c->comm_stream, hipStreamNonBlocking));
    HIPC(hipEventCreateWithFlags(&c->ev_ar_ready, hipEventDisableTiming));
    HIPC(hipEventCreateWithFlags(&c->ev_ar_done, hipEventDisableTiming));
    HIPC(dalloc(&c->adv_all, (size_t)3 * world + 4));
    return 0;
}
int mi_comm_destroy(mi_ctx* c) {
    if (!c || !c->comm) return 0;
    hipStreamSynchronize(c->comm_stream); hipStreamSynchronize(c->stream);
    if (c->comm_grad) { ncclCommDestroy(c->comm_grad); c->comm_grad = nullptr; }
    ncclCommDestroy(c->comm); c->comm = nullptr;
    hipStreamDestroy(c->comm_stream); hipEventDestroy(c->ev_ar_ready); hipEventDestroy(c->ev_ar_done);
    hipFree(c->adv_all); c->adv_all = nullptr; c->comm_world = 1; c->comm_rank = 0;
    c->ar_armed = c->ar_issued = c->ar_inflight = false;
    return 0;
}
int mi_allreduce_arm(mi_ctx* c) {
    ARG(c, "null"); ARG(c->comm, "no communicator: call mi_comm_init first"); ARG(!c->ar_inflight, "a gradient all-reduce is already in flight");
    c->ar_armed = true;
    return 0;
}
int mi_allreduce_grads(mi_ctx* c) {
    ARG(c, "null"); ARG(c->comm, "no communicator: call mi_comm_init first"); JOIN(c);
    if (c->ar_issued) return 0;              // the armed backward pass already sent both regions
    c->ar_armed = true;
    issue_grad_allreduce(c, 0, c->n_params, true);
    NETCHK(c);
    return 0;
}
int mi_allreduce_buffer(mi_ctx* c, int32_t which, int64_t n) {
    ARG(c, "null"); ARG(c->comm, "no communicator: call mi_comm_init first"); JOIN(c);
    if (which == MI_PTR_FS_KEYS) {                             // max over the ranks of the per-column candidates
        ARG(c->fs_keys && n == 2048, "MI_PTR_FS_KEYS: the 2048 keys of an IMPALA context");
        NCCLC(ncclAllReduce(c->fs_keys, c->fs_keys, 2048, ncclInt64, ncclMax, c->comm, c->stream));
        return 0;
    }
    float* p = nullptr; int64_t cap = 0;
    switch (which) {
        case MI_PTR_LOSS_STATS: p = c->loss_stats; cap = 32; break;
        case MI_PTR_STATS_RING: p = c->stats_ring; cap = (int64_t)c->log_cap * 32; break;
        case MI_PTR_GRADS: p = c->grads; cap = c->n_params; break;
        default: return fail(-1, "mi_allreduce_buffer: unknown buffer id");
    }
    ARG(n >= 1 && n <= cap, "count");
    NCCLC(ncclAllReduce(p, p, (size_t)n, ncclFloat, ncclSum, c->comm, c->stream));
    return 0;
}
// Storage.compute_estimates' advantage normalisation over ALL ranks' envs (common/storage.py:78-79): local {count, mean, M2} in fp64,
// all-gather, Chan merge and the apply pass, all on the device (no host round trip)
int mi_adv_normalize_global(mi_ctx* c) {
    ARG(c, "null"); ARG(c->comm, "no communicator: call mi_comm_init first"); JOIN(c);
    launch_advnorm_stats(c->adv, c->T * c->E, c->adv_stats, c->stream);
    NCCLC(ncclAllGather(c->adv_stats, c->adv_all, 3, ncclDouble, c->comm, c->stream));
    launch_advnorm_merge(c->adv_all, c->comm_world, c->adv_stats, c->stream);
    launch_advnorm_apply(c->adv, c->T * c->E, c->adv_stats, c->stream);
    HIPC(hipGetLastError());
    return 0;
}

int mi_device_ptr(mi_ctx* c, int32_t which, void** ptr, int64_t* n) {
    ARG(c && ptr && n, "null"); JOIN(c);
    switch (which) {
        case MI_PTR_GRADS: *ptr = c->grads; *n = c->n_params; return 0;
        case MI_PTR_LOSS_STATS: *ptr = c->loss_stats; *n = 32; return 0;
        case MI_PTR_PARAMS: *ptr = c->params; *n = c->n_params; return 0;
        case MI_PTR_STATS_RING: *ptr = c->stats_ring; *n = (int64_t)c->log_cap * 32; return 0;
        case MI_PTR_FS_KEYS: ARG(c->fs_keys, "IMPALA contexts only"); *ptr = c->fs_keys; *n = 2048; return 0;      // 2048 int64 keys
        default: return fail(-1, "unknown pointer id");
    }
}

// ------------------------------------------------------------------------------------------ op-level test entry points
static int shape_of(int cin, int cout, int hw, ConvShape* s) {
    for (int k = 0; k < CS_COUNT; ++k) {
        int a, b, h; conv_shape_dims((ConvShape)k, &a, &b, &h);
        if (a == cin && b == cout && h == hw) { *s = (ConvShape)k; return 0; }
    }
    return fail(-1, "unsupported conv shape");
}

// host-side activation conversion for the op-level entry points of a bf16 context (round to nearest even)
static std::vector<uint16_t> host_to_bf16(const float* x, size_t n) {
    std::vector<uint16_t> o(n);
    for (size_t k = 0; k < n; ++k) { uint32_t u; memcpy(&u, x + k, 4); o[k] = (uint16_t)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16); }
    return o;
}
static void host_from_bf16(const uint16_t* h, float* x, size_t n) {
    for (size_t k = 0; k < n; ++k) { const uint32_t u = ((uint32_t)h[k]) << 16; memcpy(x + k, &u, 4); }
}
// upload an activation tensor in the context's storage type
static int upload_act(mi_ctx* c, const float* host, size_t n, void** dev) {
    HIPC(hipMalloc(dev, n * 4 + 256));
    if (c->bf) { auto h = host_to_bf16(host, n); HIPC(hipMemcpy(*dev, h.data(), n * 2, hipMemcpyHostToDevice)); }
    else HIPC(hipMemcpy(*dev, host, n * 4, hipMemcpyHostToDevice));
    return 0;
}
static int download_act(mi_ctx* c, const void* dev, float* host, size_t n) {
    if (c->bf) { std::vector<uint16_t> h(n); HIPC(hipMemcpy(h.data(), dev, n * 2, hipMemcpyDeviceToHost)); host_from_bf16(h.data(), host, n); }
    else HIPC(hipMemcpy(host, dev, n * 4, hipMemcpyDeviceToHost));
    return 0;
}

int mi_op_conv3x3(mi_ctx* c, int32_t mode, int32_t cin, int32_t cout, int32_t hw, int32_t n, const void* in, int32_t in_is_u8,
                  int32_t relu_in, const float* w_ref, const float* bias, const float* res, const float* mask, const float* dout,
                  float* out, float* dbias_out) {
    ARG(c && w_ref && out && n >= 1, "null"); JOIN(c);
    ConvShape s;
    if (shape_of(cin, cout, hw, &s)) return -1;
    ARG((s == CS_3_16_64) == (in_is_u8 != 0) || mode == 1, "block1.conv takes uint8 frames");
    const size_t px = (size_t)n * hw * hw;
    TensorDesc td{"w", 0, 0, (int64_t)cout * cin * 9, K_CONVW, cout, cin};
    std::vector<float> wdev(td.n);
    to_device_layout(td, w_ref, wdev.data());
    float *dw = nullptr, *db = nullptr;
    void *din = nullptr, *dres = nullptr, *dmask = nullptr, *ddout = nullptr, *dout_buf = nullptr;
    HIPC(dalloc(&dw, td.n)); HIPC(hipMemcpy(dw, wdev.data(), td.n * 4, hipMemcpyHostToDevice));
    if (bias) { HIPC(dalloc(&db, cout)); HIPC(hipMemcpy(db, bias, cout * 4, hipMemcpyHostToDevice)); }
    const int out_ch = (mode == 1) ? cin : cout;
    if (mode != 1) {
        if (in_is_u8) { HIPC(hipMalloc(&din, px * 3 + 256)); HIPC(hipMemcpy(din, in, px * 3, hipMemcpyHostToDevice)); }
        else if (int r = upload_act(c, (const float*)in, px * cin, &din)) return r;
    }
    if (mode >= 3) {        // a block's first conv fused with the block's max pool (bf16): 3 = forward -> pooled map,
                            // 4 = weight gradient, 5 = data gradient -- both from the POOLED gradient + the forward's arg-max bytes
        ARG(c->bf && (s == CS_3_16_64 || s == CS_16_32_32 || s == CS_32_32_16), "fused conv+pool modes: block1/2/3.conv in bf16 precision only");
        ARG(mode <= 7 && !(mode >= 5 && s == CS_3_16_64) && in, "mode");
        const size_t pp = (size_t)n * (hw / 2) * (hw / 2) * cout;
        void *dp = nullptr, *dgi = nullptr; uint8_t* di = nullptr; unsigned short* dbank = nullptr; BankDesc* ddesc = nullptr;
        HIPC(hipMalloc(&dp, pp * 2 + 256)); HIPC(dalloc(&di, pp));
        ConvArgs a{};
        a.in = din; a.w = dw; a.bias = db; a.n = n; a.bf16 = 1; a.lut16 = c->lut16;
        if (s == CS_3_16_64) launch_conv1_pool_fwd_bf16(a, c->lut16, dp, di, c->stream);
        else {
            const long long bf_len = (long long)cout * bank_ws(cin), bd_len = (long long)cin * bank_ws(cout);
            BankDesc d[2] = {{0, 0, cout, cin, cout, cin, 0, bank_ws(cin), cin == 32 ? 9 : 5}, {0, bf_len, cin, cout, cout, cin, 1, bank_ws(cout), cout == 32 ? 9 : 5}};
            HIPC(dalloc(&dbank, (size_t)(bf_len + bd_len))); HIPC(hipMalloc((void**)&ddesc, sizeof d)); HIPC(hipMemcpy(ddesc, d, sizeof d, hipMemcpyHostToDevice));
            launch_pack_banks(dw, dbank, ddesc, 2, c->stream);
            a.wbank = dbank;
            ARG(launch_conv_pool_fwd_bf16(s, a, dp, di, c->stream), "no fused kernel");
        }
        HIPC(hipGetLastError()); NETCHK(c);
        HIPC(hipStreamSynchronize(c->stream));
        if (mode == 3) { if (int r = download_act(c, dp, out, pp)) return r; }
        else if (mode == 4) {
            ARG(dout && c->slabs, "dout");
            if (int r = upload_act(c, dout, pp, &ddout)) return r;
            float* g = nullptr;
            HIPC(dalloc(&g, td.n + cout));
            WgradArgs wa{};
            wa.in = din; wa.dout = ddout; wa.partial = c->slabs; wa.n = n; wa.bf16 = 1; wa.lut16 = c->lut16; wa.pool_arg = di;
            const int grid = wgrad_grid_for(s, n, 1);
            launch_conv_wgrad(s, wa, c->stream);
            launch_reduce_slabs(c->slabs, grid, (int)td.n + cout, g, (int)td.n, g + td.n, cout, c->stream);
            HIPC(hipGetLastError()); NETCHK(c);
            HIPC(hipStreamSynchronize(c->stream));
            std::vector<float> hg(td.n + cout);
            HIPC(hipMemcpy(hg.data(), g, hg.size() * 4, hipMemcpyDeviceToHost));
            to_ref_layout(td, hg.data(), out);
            if (dbias_out) memcpy(dbias_out, hg.data() + td.n, cout * 4);
            hipFree(g);
        } else {            // 5: data gradient; 6 / 7: the fused data + weight gradient launch (block2.conv), returning dW (+ db) / dx
            ARG(dout, "dout");
            ARG(mode == 5 || (conv_bwd_fused_grid(s, n) > 0 && c->slabs), "modes 6 / 7: block2.conv in an IMPALA context");
            if (int r = upload_act(c, dout, pp, &ddout)) return r;
            HIPC(hipMalloc(&dgi, px * cin * 2 + 256));
            ConvArgs g{};
            g.in = ddout; g.pool_arg = di; g.w = dw; g.out = dgi; g.n = n; g.bf16 = 1; g.wbank = dbank + (long long)cout * bank_ws(cin);
            float* gw = nullptr;
            if (mode >= 6) { g.wg_in = din; g.wg_partial = c->slabs; HIPC(dalloc(&gw, td.n + cout)); }
            launch_conv_dgrad(s, g, c->stream);
            if (mode >= 6) launch_reduce_slabs(c->slabs, conv_bwd_fused_grid(s, n), (int)td.n + cout, gw, (int)td.n, gw + td.n, cout, c->stream);
            HIPC(hipGetLastError()); NETCHK(c);
            HIPC(hipStreamSynchronize(c->stream));
            if (mode == 6) {
                std::vector<float> hg(td.n + cout);
                HIPC(hipMemcpy(hg.data(), gw, hg.size() * 4, hipMemcpyDeviceToHost));
                to_ref_layout(td, hg.data(), out);
                if (dbias_out) memcpy(dbias_out, hg.data() + td.n, cout * 4);
            } else if (int r = download_act(c, dgi, out, px * cin)) return r;
            if (gw) hipFree(gw);
        }
        void* fr[] = {dw, db, din, ddout, dp, di, dbank, ddesc, dgi};
        for (void* p : fr) if (p) hipFree(p);
        return 0;
    }
    if (mode >= 1) { ARG(dout, "dout"); if (int r = upload_act(c, dout, px * cout, &ddout)) return r; }
    if (res) { if (int r = upload_act(c, res, px * out_ch, &dres)) return r; }
    if (mask) { if (int r = upload_act(c, mask, px * out_ch, &dmask)) return r; }
    if (mode <= 1) {
        HIPC(hipMalloc(&dout_buf, px * out_ch * 4 + 256));
        ConvArgs a{};
        a.in = (mode == 0) ? din : ddout; a.idx = nullptr; a.in_base = 0; a.w = dw; a.bias = (mode == 0) ? db : nullptr;
        a.res = dres; a.mask = dmask; a.out = dout_buf; a.lut = c->lut; a.n = n; a.relu_in = (mode == 0) ? relu_in : 0; a.bf16 = c->bf;
        a.lut16 = c->bf ? c->lut16 : nullptr;
        if (mode == 0) launch_conv_fwd(s, a, c->stream); else launch_conv_dgrad(s, a, c->stream);
        HIPC(hipGetLastError()); NETCHK(c);
        HIPC(hipStreamSynchronize(c->stream));
        if (int r = download_act(c, dout_buf, out, px * out_ch)) return r;
    } else {
        ARG(c->slabs, "wgrad needs an IMPALA context");
        float* g = nullptr;
        HIPC(dalloc(&g, td.n + cout));
        WgradArgs a{};
        a.in = din; a.idx = nullptr; a.in_base = 0; a.dout = ddout; a.partial = c->slabs; a.lut = c->lut; a.n = n; a.relu_in = relu_in; a.bf16 = c->bf;
        a.lut16 = c->bf ? c->lut16 : nullptr;
        const int grid = wgrad_grid_for(s, n, c->bf);
        launch_conv_wgrad(s, a, c->stream);
        launch_reduce_slabs(c->slabs, grid, (int)td.n + cout, g, (int)td.n, g + td.n, cout, c->stream);
        HIPC(hipGetLastError()); NETCHK(c);
        HIPC(hipStreamSynchronize(c->stream));
        std::vector<float> hg(td.n + cout);
        HIPC(hipMemcpy(hg.data(), g, hg.size() * 4, hipMemcpyDeviceToHost));
        to_ref_layout(td, hg.data(), out);
        if (dbias_out) memcpy(dbias_out, hg.data() + td.n, cout * 4);
        hipFree(g);
    }
    void* fr[] = {dw, db, din, dres, dmask, ddout, dout_buf};
    for (void* p : fr) if (p) hipFree(p);
    return 0;
}

// Fused residual block of the bf16 mode, op level.  mode 0: (x, w1, b1, w2, b2) -> a = conv1(relu(x)) + b1, y = conv2(relu(a)) + b2 + x.
// mode 1: (dy = x, a_fwd, x_fwd, w1, w2) -> out_a = d a = convT2(dy) * (a_fwd > 0), out_y = d x = convT1(d a) * (x_fwd > 0) + dy.
int mi_op_resblock(mi_ctx* c, int32_t mode, int32_t ch, int32_t hw, int32_t n, const float* x, const float* w1_ref, const float* b1,
                   const float* w2_ref, const float* b2, const float* a_fwd, const float* x_fwd, float* out_a, float* out_y) {
    ARG(c && x && w1_ref && w2_ref && out_a && out_y && n >= 1, "null"); JOIN(c);
    ARG(c->bf, "the fused residual-block kernels exist in bf16 precision only");
    ARG((mode == 0 || mode == 3) ? (b1 && b2) : (a_fwd && x_fwd), "modes 0 / 3 need the biases, modes 1 / 2 the forward tensors");
    ConvShape s;
    if (shape_of(ch, ch, hw, &s)) return -1;
    const size_t X = (size_t)n * hw * hw * ch, wl = (size_t)ch * ch * 9;
    TensorDesc td{"w", 0, 0, (int64_t)wl, K_CONVW, ch, ch};
    std::vector<float> wdev(2 * wl + 2 * ch, 0.f);
    to_device_layout(td, w1_ref, wdev.data()); to_device_layout(td, w2_ref, wdev.data() + wl);
    if (b1) memcpy(wdev.data() + 2 * wl, b1, ch * 4);
    if (b2) memcpy(wdev.data() + 2 * wl + ch, b2, ch * 4);
    float* dparams = nullptr; unsigned short* dbanks = nullptr; BankDesc* ddesc = nullptr;
    void *dx = nullptr, *da = nullptr, *dxf = nullptr, *doa = nullptr, *doy = nullptr;
    HIPC(dalloc(&dparams, wdev.size())); HIPC(hipMemcpy(dparams, wdev.data(), wdev.size() * 4, hipMemcpyHostToDevice));
    const int ws = bank_ws(ch), nk = ch == 32 ? 9 : 5;
    const long long bl = (long long)ch * ws;
    // bank 0 feeds the kernel's first conv, bank 1 its second: forward (w1, w2) as stored; backward (w2^T, w1^T)
    const bool tr = (mode == 1 || mode == 2);
    BankDesc d[2] = {{tr ? (long long)wl : 0, 0, ch, ch, ch, ch, tr ? 1 : 0, ws, nk}, {tr ? 0 : (long long)wl, bl, ch, ch, ch, ch, tr ? 1 : 0, ws, nk}};
    HIPC(dalloc(&dbanks, (size_t)2 * bl)); HIPC(hipMalloc((void**)&ddesc, sizeof d)); HIPC(hipMemcpy(ddesc, d, sizeof d, hipMemcpyHostToDevice));
    launch_pack_banks(dparams, dbanks, ddesc, 2, c->stream);
    if (int r = upload_act(c, x, X, &dx)) return r;
    HIPC(hipMalloc(&doa, X * 2 + 256)); HIPC(hipMalloc(&doy, X * 2 + 256));
    if (mode == 2) {        // whole backward of a 16-channel block: out_y = dx, out_a[0 .. 2*(9*ch*ch + ch)) = {dW1, db1, dW2, db2} (reference layout)
        ARG((s == CS_16_16_32 || s == CS_32_32_16 || s == CS_32_32_8) && c->slabs, "the whole-backward kernels exist for the 16-channel @32x32 and 32-channel @16x16 blocks (IMPALA context)");
        if (int r = upload_act(c, a_fwd, X, &da)) return r;
        if (int r = upload_act(c, x_fwd, X, &dxf)) return r;
        const int grid = s == CS_16_16_32 ? resblock_bwd_full_grid(n) : resblock_bwd_full32_grid(s, n), slab = (int)wl + ch;
        float* g = nullptr;
        HIPC(dalloc(&g, (size_t)2 * slab));
        float* sl2 = c->slabs; float* sl1 = c->slabs + (size_t)1024 * slab;
        if (s == CS_16_16_32) launch_resblock_bwd_full_bf16(dx, da, dxf, doy, nullptr, n, dbanks, dbanks + bl, sl2, sl1, c->stream);
        else launch_resblock_bwd_full32_bf16(s, dx, da, dxf, doy, nullptr, n, dbanks, dbanks + bl, sl2, sl1, c->stream);
        launch_reduce_slabs(sl1, grid, slab, g, (int)wl, g + wl, ch, c->stream);
        launch_reduce_slabs(sl2, grid, slab, g + slab, (int)wl, g + slab + wl, ch, c->stream);
        HIPC(hipGetLastError()); NETCHK(c);
        HIPC(hipStreamSynchronize(c->stream));
        std::vector<float> hg(2 * slab);
        HIPC(hipMemcpy(hg.data(), g, hg.size() * 4, hipMemcpyDeviceToHost));
        to_ref_layout(td, hg.data(), out_a); memcpy(out_a + wl, hg.data() + wl, ch * 4);
        to_ref_layout(td, hg.data() + slab, out_a + slab); memcpy(out_a + slab + wl, hg.data() + slab + wl, ch * 4);
        if (int r = download_act(c, doy, out_y, X)) return r;
        void* fr2[] = {dparams, dbanks, ddesc, dx, da, dxf, doa, doy, g};
        for (void* q : fr2) if (q) hipFree(q);
        return 0;
    }
    if (mode == 3) {        // res1 + res2 in one launch, both with (w1, b1, w2, b2): out_a = second conv1 output, out_y = second block output
        const float* bb[4] = {dparams + 2 * wl, dparams + 2 * wl + ch, dparams + 2 * wl, dparams + 2 * wl + ch};
        const unsigned short* bk[4] = {dbanks, dbanks + bl, dbanks, dbanks + bl};
        launch_resblock_pair_bf16(s, dx, bb, nullptr, nullptr, doa, doy, n, bk, c->stream);
    } else if (mode == 0) {
        launch_resblock_bf16(s, dx, dparams + 2 * wl, dparams + 2 * wl + ch, doa, doy, n, dbanks, dbanks + bl, c->stream);
    } else {
        if (int r = upload_act(c, a_fwd, X, &da)) return r;
        if (int r = upload_act(c, x_fwd, X, &dxf)) return r;
        launch_resblock_bwd_bf16(s, dx, da, dxf, doa, doy, n, dbanks, dbanks + bl, c->stream);
    }
    HIPC(hipGetLastError()); NETCHK(c);
    HIPC(hipStreamSynchronize(c->stream));
    if (int r = download_act(c, doa, out_a, X)) return r;
    if (int r = download_act(c, doy, out_y, X)) return r;
    void* fr[] = {dparams, dbanks, ddesc, dx, da, dxf, doa, doy};
    for (void* q : fr) if (q) hipFree(q);
    return 0;
}

// bit 0: run rollout-sized bf16 inference passes on the separate block-2 / block-3 kernels instead of the fused launch (parity A/B)
int mi_debug_flags(mi_ctx* c, int32_t flags) { ARG(c, "null"); JOIN(c); c->rollout_tail = !(flags & 1); c->no_pull = (flags & 4) != 0; c->side_on = !(flags & 16); return 0; }

// Philox4x32-10 known-answer hook: n x {c0,c1,c2,c3,k0,k1} in, n x 4 output words and the n uniforms the sampler would draw
// for (seed = k0 | k1 << 32, counter = c0 | c1 << 32) out.
int mi_debug_philox(mi_ctx* c, const uint32_t* ctr_key6, int32_t n, uint32_t* out4, float* u_out) {
    ARG(c && ctr_key6 && out4 && u_out, "null"); JOIN(c); ARG(n >= 1 && n <= (1 << 24), "n");
    uint32_t *din = nullptr, *dout = nullptr; float* du = nullptr;
    HIPC(hipMalloc((void**)&din, (size_t)n * 24)); HIPC(hipMalloc((void**)&dout, (size_t)n * 16)); HIPC(hipMalloc((void**)&du, (size_t)n * 4));
    HIPC(hipMemcpy(din, ctr_key6, (size_t)n * 24, hipMemcpyHostToDevice));
    launch_philox_debug(din, n, dout, du, c->stream);
    HIPC(hipGetLastError());
    HIPC(hipStreamSynchronize(c->stream));
    HIPC(hipMemcpy(out4, dout, (size_t)n * 16, hipMemcpyDeviceToHost));
    HIPC(hipMemcpy(u_out, du, (size_t)n * 4, hipMemcpyDeviceToHost));
    hipFree(din); hipFree(dout); hipFree(du);
    return 0;
}

// Measurement hook ("hipGraph"): wall-clock microseconds per policy step of slot t -- the step's launches (conv stack,
// embedder.fc, fused heads + sample: 5 kernels in bf16 mode) followed by a stream wait, `iters` times back to back -- issued eagerly
// (mode 0) or as ONE replay of a graph captured from the same launches (mode 1).  What a captured group step could save on the
// rollout's dependency chain, without touching the production path (whose per-step arguments change: slot, counters, ticket).
int mi_debug_step_latency(mi_ctx* c, int32_t t, int32_t iters, int32_t mode, float* us_out) {
    ARG(c && us_out, "null"); JOIN(c); ARG(t >= 0 && t <= c->T && iters >= 1 && (mode == 0 || mode == 1), "t / iters / mode");
    const int E = c->E;
    InputSrc src{c->frames ? (const void*)c->frames : (const void*)c->obsf, nullptr, (long long)t * E};
    auto issue = [&]() {
        c->prof.phase = 0;
        net_forward(c, src, E, true, false);
        launch_heads_sample(c->feat, c->params + c->wh_off, c->params + c->bh_off, E, c->H, c->A, nullptr, 1234ull, (unsigned long long)t * E,
                            nullptr, nullptr, c->value + (size_t)t * E, c->h_pack, nullptr, nullptr, nullptr, nullptr, c->stream);
    };
    const bool prof_on = c->prof.on; c->prof.on = false;       // (no event records inside a capture)
    issue();                                                   // warm: packed banks in place, lazy function attributes set
    HIPC(hipGetLastError()); NETCHK(c);
    HIPC(hipStreamSynchronize(c->stream));
    hipGraph_t graph = nullptr; hipGraphExec_t exec = nullptr;
    if (mode == 1) {
        HIPC(hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
        issue();
        HIPC(hipStreamEndCapture(c->stream, &graph));
        HIPC(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
        HIPC(hipGraphLaunch(exec, c->stream)); HIPC(hipStreamSynchronize(c->stream));
    }
    const auto t0 = std::chrono::steady_clock::now();
    for (int k = 0; k < iters; ++k) {
        if (mode == 1) HIPC(hipGraphLaunch(exec, c->stream)); else issue();
        HIPC(hipStreamSynchronize(c->stream));
    }
    const auto t1 = std::chrono::steady_clock::now();
    *us_out = (float)(std::chrono::duration_cast<std::chrono::nanoseconds>(t1 - t0).count() / 1e3 / iters);
    if (exec) hipGraphExecDestroy(exec);
    if (graph) hipGraphDestroy(graph);
    c->prof.on = prof_on;
    HIPC(hipGetLastError()); NETCHK(c);
    return 0;
}

// Read back what the last training-mode pass (mi_minibatch) left in the activation buffers, as fp32 NHWC: which = 8 * block + k with
// k = 0 P0 (pooled map), 1 A1, 2 P1, 3 A2, 4 P2 (res1.conv1 out, res1 out, res2.conv1 out, block out), 5 the max-pool arg-max bytes
// (window position ky*3+kx as float); which = 100: the 256 features.  For teacher-forced backward parity tests.
int mi_debug_read(mi_ctx* c, int32_t which, int32_t n, float* out) {
    ARG(c && out, "null"); JOIN(c); ARG(n >= 1 && n <= c->NB, "n must be in [1, max_batch]");
    if (which == 100) {
        HIPC(hipMemcpyAsync(out, c->feat, (size_t)n * c->H * 4, hipMemcpyDeviceToHost, c->stream));
        HIPC(hipStreamSynchronize(c->stream));
        return 0;
    }
    ARG(c->cfg.arch == MI_ARCH_IMPALA && which >= 0 && which < 24 && (which & 7) <= 5, "which");
    const Block& k = c->blk[which >> 3];
    const size_t pe = (size_t)n * (k.hin / 2) * (k.hin / 2) * k.cout;
    HIPC(hipStreamSynchronize(c->stream));
    if ((which & 7) == 5) {
        std::vector<uint8_t> h(pe);
        HIPC(hipMemcpy(h.data(), k.PI, pe, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < pe; ++i) out[i] = (float)h[i];
        return 0;
    }
    const float* src[5] = {k.P0, k.A1, k.P1, k.A2, k.P2};
    return download_act(c, src[which & 7], out, pe);
}

int mi_op_maxpool(mi_ctx* c, int32_t mode, int32_t n, int32_t hw, int32_t ch, const float* in, const float* dout, float* out) {
    ARG(c && in && out, "null"); JOIN(c); ARG(n >= 1, "n");
    ARG((hw == 64 && ch == 16) || (hw == 32 && ch == 32) || (hw == 16 && ch == 32), "max pool shapes of the IMPALA blocks only: (64,16), (32,32), (16,32)");
    const size_t X = (size_t)n * hw * hw * ch, p = X / 4;
    void *din = nullptr, *dp = nullptr, *dd = nullptr, *dg = nullptr; uint8_t* di = nullptr;
    if (int r = upload_act(c, in, X, &din)) return r;
    HIPC(hipMalloc(&dp, p * 4 + 256)); HIPC(dalloc(&di, p));
    if (c->bf) launch_maxpool_fwd_bf16(din, dp, di, n, hw, ch, c->stream); else launch_maxpool_fwd((const float*)din, (float*)dp, di, n, hw, ch, c->stream);
    if (mode == 0) { HIPC(hipStreamSynchronize(c->stream)); if (int r = download_act(c, dp, out, p)) return r; }
    else {
        ARG(dout, "dout");
        if (int r = upload_act(c, dout, p, &dd)) return r;
        HIPC(hipMalloc(&dg, X * 4 + 256));
        if (c->bf) launch_maxpool_bwd_bf16(dd, di, dg, n, hw, ch, c->stream); else launch_maxpool_bwd((const float*)dd, di, (float*)dg, n, hw, ch, c->stream);
        HIPC(hipStreamSynchronize(c->stream));
        if (int r = download_act(c, dg, out, X)) return r;
    }
    HIPC(hipGetLastError()); NETCHK(c);
    hipFree(din); hipFree(dp); hipFree(di); if (dd) hipFree(dd); if (dg) hipFree(dg);
    return 0;
}

int mi_op_gemm(mi_ctx* c, int32_t M, int32_t N, int32_t K, const float* A, int64_t sam, int64_t sak, const float* B, int64_t sbk,
               int64_t sbn, float* C) {
    ARG(c && A && B && C, "null"); JOIN(c);
    const size_t na = (size_t)((M - 1) * sam + (K - 1) * sak + 1), nb = (size_t)((K - 1) * sbk + (N - 1) * sbn + 1);
    float *da = nullptr, *db = nullptr, *dc = nullptr;
    HIPC(dalloc(&da, na)); HIPC(dalloc(&db, nb)); HIPC(dalloc(&dc, (size_t)M * N));
    HIPC(hipMemcpy(da, A, na * 4, hipMemcpyHostToDevice)); HIPC(hipMemcpy(db, B, nb * 4, hipMemcpyHostToDevice));
    GemmArgs g{};
    g.ws = c->gemm_ws; g.ws_floats = c->gemm_ws_floats;
    g.A = da; g.B = db; g.C = dc; g.M = M; g.N = N; g.K = K; g.sam = sam; g.sak = sak; g.sbk = sbk; g.sbn = sbn; g.ldc = N;
    launch_gemm(g, c->stream);
    HIPC(hipGetLastError()); NETCHK(c);
    HIPC(hipStreamSynchronize(c->stream));
    HIPC(hipMemcpy(C, dc, (size_t)M * N * 4, hipMemcpyDeviceToHost));
    hipFree(da); hipFree(db); hipFree(dc);
    return 0;
}

// D = A(16x4) * B(4x16) with asymmetric integer data through the operand maps the kernels assume:
// A[i = lane&15][k = lane>>4], B[k = lane>>4][j = lane&15], D[row = (lane>>4)*4 + r][col = lane&15]
__global__ void mfma_selftest_kernel(float* d) {
    const int lane = threadIdx.x, i = lane & 15, q = lane >> 4;
    const float a = (float)(i * 7 + q * 3 + 1), b = (float)(q * 5 - i * 2 + 11);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) d[(q * 4 + r) * 16 + i] = acc[r];
}
int mi_selftest_mfma(mi_ctx* c, float* max_err) {
    ARG(c && max_err, "null"); JOIN(c);
    float* d = nullptr;
    HIPC(dalloc(&d, 256));
    hipLaunchKernelGGL(mfma_selftest_kernel, dim3(1), dim3(64), 0, c->stream, d);
    HIPC(hipStreamSynchronize(c->stream));
    float h[256];
    HIPC(hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost));
    hipFree(d);
    float worst = 0.f;
    for (int m = 0; m < 16; ++m)
        for (int n = 0; n < 16; ++n) {
            float ref = 0.f;
            for (int k = 0; k < 4; ++k) ref += (float)(m * 7 + k * 3 + 1) * (float)(k * 5 - n * 2 + 11);
            worst = fmaxf(worst, fabsf(ref - h[m * 16 + n]));
        }
    *max_err = worst;
    return 0;
}
