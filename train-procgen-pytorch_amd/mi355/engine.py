"""Thin ctypes layer over the C ABI in include/mi355ppo.h.

There is deliberately NO fallback: if libmi355ppo.so is missing or no MI355X is visible, creating an
Engine raises.  (The CPU restatement under oracle/ is test infrastructure and is never imported here.)
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

ARCH_IMPALA, ARCH_MLP = 0, 1
F_REW, F_DONE, F_VALUE, F_LOGP, F_ADV, F_RET, F_ACT = range(7)
PTR_GRADS, PTR_LOSS_STATS, PTR_PARAMS, PTR_STATS_RING, PTR_FS_KEYS = 0, 1, 2, 3, 4
LOSS_FIELDS = ("pi_loss", "value_loss", "entropy", "x_ent", "total", "fs", "marg", "_pad")


class EngineError(RuntimeError):
    pass


class _Config(C.Structure):
    _fields_ = [("arch", C.c_int32), ("n_steps", C.c_int32), ("n_envs", C.c_int32), ("n_actions", C.c_int32),
                ("obs_dim", C.c_int32), ("mlp_depth", C.c_int32), ("mlp_width", C.c_int32), ("out_dim", C.c_int32),
                ("max_batch", C.c_int32), ("device", C.c_int32), ("precision", C.c_int32), ("reserved", C.c_int32 * 5),
                ("stream", C.c_void_p)]


class _HParams(C.Structure):
    _fields_ = [(n, C.c_float) for n in ("eps_clip", "value_coef", "entropy_coef", "x_entropy_coef",
                                         "entropy_multiplier", "fs_coef")]


def lib_path():
    return os.path.join(_HERE, "libmi355ppo.so")


EXPORTS = ("mi_last_error mi_create mi_destroy mi_sync mi_host_alloc mi_host_free mi_host_register mi_host_unregister mi_param_count mi_set_params "
           "mi_get_params mi_copy_params mi_get_grads mi_set_adam_state mi_get_adam_state mi_put_obs mi_get_obs mi_put_step "
           "mi_put_policy_outputs mi_read_field mi_write_field mi_policy_step mi_rollout_step mi_rollout_groups mi_rollout_submit mi_rollout_wait mi_predict_staged mi_value_saliency mi_commit_staged mi_set_gru mi_rec_state mi_get_hidden mi_forward_rec mi_forward mi_compute_estimates "
           "mi_adv_stats mi_adv_apply mi_minibatch mi_minibatch_multi mi_optimizer_step mi_loss_log_read mi_device_ptr "
           "mi_set_multirank mi_minibatch_finish mi_loss_log_finalize mi_profile_enable mi_profile_read mi_profile_class_name mi_op_conv3x3 mi_op_resblock mi_op_maxpool mi_op_gemm mi_selftest_mfma mi_debug_read mi_debug_flags mi_comm_unique_id mi_comm_init mi_comm_destroy mi_allreduce_arm mi_allreduce_grads mi_allreduce_buffer mi_adv_normalize_global mi_minibatch_positions mi_debug_philox mi_debug_step_latency").split()


def load_library():
    """dlopen the in-tree library (built by `make -C csrc` / __graft_entry__.build())."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = lib_path()
    if not os.path.exists(path):
        raise EngineError(f"{path} not found: build it with `make -C train-procgen-pytorch_amd/csrc` "
                          "(python __graft_entry__.py build).  There is no CPU fallback.")
    lib = C.CDLL(path)
    lib.mi_last_error.restype = C.c_char_p
    lib.mi_param_count.restype = C.c_int64
    lib.mi_host_alloc.restype = C.c_void_p
    lib.mi_host_alloc.argtypes = [C.c_size_t]
    lib.mi_host_free.argtypes = [C.c_void_p]
    lib.mi_host_free.restype = None
    lib.mi_profile_class_name.restype = C.c_char_p
    lib.mi_host_register.argtypes = [C.c_void_p, C.c_size_t]
    lib.mi_host_unregister.argtypes = [C.c_void_p]
    # the per-env-step entry point is called T+1 times per iteration: declared argtypes let plain ints / addresses through
    # without building ctypes wrapper objects on every call
    lib.mi_rollout_step.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.mi_rollout_submit.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]
    lib.mi_rollout_wait.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
    _LIB = lib
    return lib


def _fp(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


class Engine:
    """One device context (one GPU).  Methods mirror the C entry points one to one."""

    def __init__(self, arch, n_steps, n_envs, n_actions, max_batch, obs_dim=0, mlp_depth=0, mlp_width=0, out_dim=256,
                 device=0, stream=None, precision="fp32"):
        self.lib = load_library()
        self.arch = ARCH_IMPALA if arch in ("impala", ARCH_IMPALA) else ARCH_MLP
        cfg = _Config(arch=self.arch, n_steps=n_steps, n_envs=n_envs, n_actions=n_actions, obs_dim=obs_dim,
                      mlp_depth=mlp_depth, mlp_width=mlp_width, out_dim=out_dim, max_batch=max_batch, device=device,
                      precision={"fp32": 0, "bf16": 1}[precision], stream=stream)
        self.precision = precision
        self.T, self.E, self.A, self.max_batch = n_steps, n_envs, n_actions, max(max_batch, n_envs)
        self.H = 256 if self.arch == ARCH_IMPALA else out_dim
        self.obs_dim = obs_dim
        self._ctx = C.c_void_p()
        self._chk(self.lib.mi_create(C.byref(cfg), C.byref(self._ctx)))
        self.n_params = int(self.lib.mi_param_count(self._ctx))
        self._pinned = []
        self._registered = {}          # address -> (nbytes, array kept alive): caller buffers page-locked in place (mi_host_register)
        self._register_refused = 0

    # ------------------------------------------------------------------ plumbing
    def _chk(self, rc):
        if rc != 0:
            raise EngineError(f"libmi355ppo error {rc}: {self.lib.mi_last_error().decode()}")

    def close(self):
        if getattr(self, "_ctx", None) and self._ctx.value:
            for p in self._pinned:
                self.lib.mi_host_free(C.c_void_p(p))
            self._pinned = []
            for p in list(getattr(self, "_registered", {})):
                self.lib.mi_host_unregister(p)
            self._registered = {}
            self.lib.mi_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def sync(self):
        self._chk(self.lib.mi_sync(self._ctx))

    def pinned(self, shape, dtype):
        """numpy array over page-locked host memory owned by this engine."""
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        p = self.lib.mi_host_alloc(n)
        if not p:
            raise EngineError("mi_host_alloc failed")
        self._pinned.append(p)
        self._pinned_n = getattr(self, "_pinned_n", {})
        self._pinned_n[p] = n
        buf = (C.c_char * n).from_address(p)
        return np.frombuffer(buf, dtype=dtype).reshape(shape)

    MAX_REGISTERED = 64

    def dma_ready(self, arr):
        """True if `arr` (C-contiguous numpy) can be handed to rollout_submit / put_obs as it is: memory from Engine.pinned, or
        caller memory page-locked in place now (mi_host_register, once per buffer: an env that hands its frames out in the same few
        buffers step after step -- Procgen's rgb buffer, a pool -- is uploaded from where it lies, with no staging copy).  False when
        the runtime refuses the range or the env keeps producing new buffers (more than MAX_REGISTERED distinct ones): stage it."""
        addr = arr.__array_interface__['data'][0]
        hit = self._registered.get(addr)
        if hit is not None:
            return hit[0] >= arr.nbytes
        if getattr(self, "_pinned_n", {}).get(addr, -1) >= arr.nbytes:
            return True
        if len(self._registered) >= self.MAX_REGISTERED or self._register_refused >= 8 or not arr.flags.c_contiguous:
            return False
        base = arr
        while isinstance(getattr(base, "base", None), np.ndarray):
            base = base.base
        if self.lib.mi_host_register(addr, arr.nbytes) != 0:
            self._register_refused += 1
            return False
        self._registered[addr] = (arr.nbytes, base)
        return True

    # ------------------------------------------------------------------ parameters
    def set_params(self, flat):
        flat = _f32(flat)
        self._chk(self.lib.mi_set_params(self._ctx, _fp(flat), C.c_int64(flat.size)))

    def get_params(self):
        out = np.empty(self.n_params, np.float32)
        self._chk(self.lib.mi_get_params(self._ctx, _fp(out), C.c_int64(out.size)))
        return out

    def copy_params_from(self, src):
        """This context's parameters := src's (another Engine on the same GPU), device to device, ordered on both streams."""
        self._chk(self.lib.mi_copy_params(self._ctx, src._ctx))

    def get_grads(self):
        out = np.empty(self.n_params, np.float32)
        self._chk(self.lib.mi_get_grads(self._ctx, _fp(out), C.c_int64(out.size)))
        return out

    def set_adam_state(self, m, v):
        m, v = _f32(m), _f32(v)
        self._chk(self.lib.mi_set_adam_state(self._ctx, _fp(m), _fp(v), C.c_int64(m.size)))

    def get_adam_state(self):
        m, v = np.empty(self.n_params, np.float32), np.empty(self.n_params, np.float32)
        self._chk(self.lib.mi_get_adam_state(self._ctx, _fp(m), _fp(v), C.c_int64(m.size)))
        return m, v

    # ------------------------------------------------------------------ rollout storage
    def put_obs(self, t, obs):
        want = np.uint8 if self.arch == ARCH_IMPALA else np.float32
        obs = np.ascontiguousarray(obs, dtype=want)
        self._chk(self.lib.mi_put_obs(self._ctx, C.c_int32(t), _fp(obs), C.c_size_t(obs.nbytes)))
        return obs      # caller keeps it alive until the next sync

    def get_obs(self, t):
        if self.arch == ARCH_IMPALA:
            out = np.empty((self.E, 64, 64, 3), np.uint8)
        else:
            out = np.empty((self.E, self.obs_dim), np.float32)
        self._chk(self.lib.mi_get_obs(self._ctx, C.c_int32(t), _fp(out), C.c_size_t(out.nbytes)))
        return out

    def put_step(self, t, rew, done):
        rew, done = _f32(rew), _f32(done)
        self._chk(self.lib.mi_put_step(self._ctx, C.c_int32(t), _fp(rew), _fp(done)))
        return rew, done

    def put_policy_outputs(self, t, act=None, logp=None, value=None):
        act = None if act is None else np.ascontiguousarray(act, dtype=np.int32)
        logp = None if logp is None else _f32(logp)
        value = None if value is None else _f32(value)
        self._chk(self.lib.mi_put_policy_outputs(self._ctx, C.c_int32(t), _fp(act), _fp(logp), _fp(value)))

    def read_field(self, field):
        n = (self.T + 1) * self.E if field == F_VALUE else self.T * self.E
        out = np.empty(n, np.float32)
        self._chk(self.lib.mi_read_field(self._ctx, C.c_int32(field), _fp(out), C.c_int64(n)))
        return out.reshape(-1, self.E)

    def write_field(self, field, arr):
        arr = _f32(arr).reshape(-1)
        self._chk(self.lib.mi_write_field(self._ctx, C.c_int32(field), _fp(arr), C.c_int64(arr.size)))

    # ------------------------------------------------------------------ policy
    def policy_step(self, t, seed=0, u=None, want_outputs=True):
        u = None if u is None else _f32(u)
        if not want_outputs:
            self._chk(self.lib.mi_policy_step(self._ctx, C.c_int32(t), C.c_uint64(seed), _fp(u), None, None, None))
            return None
        act = np.empty(self.E, np.int64)
        logp = np.empty(self.E, np.float32)
        val = np.empty(self.E, np.float32)
        self._chk(self.lib.mi_policy_step(self._ctx, C.c_int32(t), C.c_uint64(seed), _fp(u), _fp(act), _fp(logp), _fp(val)))
        return act, logp, val

    def rollout_step(self, t, rew_prev=None, done_prev=None, seed=0, u=None):
        """store(t-1)'s reward/done + policy step on slot t in one call (one packed upload, one packed read-back).
        (Called 257 times per iteration between GPU steps: one result buffer, raw pointers instead of .ctypes -- 2 us per call.)"""
        u = None if u is None else _f32(u)
        if rew_prev is not None and (not isinstance(rew_prev, np.ndarray) or rew_prev.dtype != np.float32 or not rew_prev.flags.c_contiguous):
            rew_prev = _f32(rew_prev)
        if done_prev is not None and (not isinstance(done_prev, np.ndarray) or done_prev.dtype != np.float32 or not done_prev.flags.c_contiguous):
            done_prev = _f32(done_prev)
        E = self.E
        out = np.empty(4 * E, np.float32)                       # [act as int64 | logp | value]
        p = out.__array_interface__['data'][0]
        rc = self.lib.mi_rollout_step(self._ctx, t, None if rew_prev is None else rew_prev.__array_interface__['data'][0],
                                      None if done_prev is None else done_prev.__array_interface__['data'][0], seed,
                                      None if u is None else u.__array_interface__['data'][0], p, p + 8 * E, p + 12 * E)
        if rc:
            self._chk(rc)
        return out[:2 * E].view(np.int64), out[2 * E:3 * E], out[3 * E:]

    # pipelined rollout: env groups, submit / wait (see include/mi355ppo.h)
    def rollout_groups(self, n_groups):
        self._chk(self.lib.mi_rollout_groups(self._ctx, C.c_int32(n_groups)))
        self.n_groups = n_groups

    def rollout_submit(self, t, group, frames=None, rew_prev=None, done_prev=None, seed=0, u=None):
        """frames: this group's observations in PINNED memory (Engine.pinned), kept unchanged until rollout_wait(group); None = slot t holds them."""
        addr = lambda a: None if a is None else a.__array_interface__['data'][0]
        if rew_prev is not None and (rew_prev.dtype != np.float32 or not rew_prev.flags.c_contiguous):
            rew_prev = _f32(rew_prev)
        if done_prev is not None and (done_prev.dtype != np.float32 or not done_prev.flags.c_contiguous):
            done_prev = _f32(done_prev)
        u = None if u is None else _f32(u)
        if frames is not None and not frames.flags.c_contiguous:
            raise EngineError("rollout_submit needs a contiguous (pinned) frame buffer")
        rc = self.lib.mi_rollout_submit(self._ctx, t, group, addr(frames), 0 if frames is None else frames.nbytes, addr(rew_prev), addr(done_prev), seed, addr(u))
        if rc:
            self._chk(rc)
        self._keep_alive = (frames, rew_prev, done_prev, u)

    def rollout_wait(self, group):
        n = self.E // getattr(self, "n_groups", 1)
        out = np.empty(4 * n, np.float32)                       # [act as int64 | logp | value]
        p = out.__array_interface__['data'][0]
        rc = self.lib.mi_rollout_wait(self._ctx, group, p, p + 8 * n, p + 12 * n)
        if rc:
            self._chk(rc)
        return out[:2 * n].view(np.int64), out[2 * n:3 * n], out[3 * n:]

    def rollout_wait_into(self, group, act_addr, logp_addr, val_addr):
        """rollout_wait writing into caller arrays (raw addresses of this group's int64 / float32 / float32 slices): the collector's
        per-group-step call, no allocation."""
        rc = self.lib.mi_rollout_wait(self._ctx, group, act_addr, logp_addr, val_addr)
        if rc:
            self._chk(rc)

    def predict_staged(self, obs, seed=0, counter=0, u=None):
        want = np.uint8 if self.arch == ARCH_IMPALA else np.float32
        obs = np.ascontiguousarray(obs, dtype=want)
        u = None if u is None else _f32(u)
        act, logp, val = np.empty(self.E, np.int64), np.empty(self.E, np.float32), np.empty(self.E, np.float32)
        self._chk(self.lib.mi_predict_staged(self._ctx, _fp(obs), C.c_size_t(obs.nbytes), C.c_uint64(seed),
                                             C.c_uint64(counter), _fp(u), _fp(act), _fp(logp), _fp(val)))
        return act, logp, val

    def value_saliency(self, obs, seed=0, counter=0, u=None):
        """predict_staged + d value / d observation: (act, logp, value, grad) with grad (E,64,64,3) NHWC for IMPALA / (E,obs_dim) for the MLP."""
        want = np.uint8 if self.arch == ARCH_IMPALA else np.float32
        obs = np.ascontiguousarray(obs, dtype=want)
        u = None if u is None else _f32(u)
        act, logp, val = np.empty(self.E, np.int64), np.empty(self.E, np.float32), np.empty(self.E, np.float32)
        grad = np.empty((self.E, 64, 64, 3) if self.arch == ARCH_IMPALA else (self.E, self.obs_dim), np.float32)
        self._chk(self.lib.mi_value_saliency(self._ctx, _fp(obs), C.c_size_t(obs.nbytes), C.c_uint64(seed), C.c_uint64(counter), _fp(u),
                                             _fp(act), _fp(logp), _fp(val), _fp(grad)))
        return act, logp, val, grad

    def commit_staged(self, t):
        self._chk(self.lib.mi_commit_staged(self._ctx, C.c_int32(t)))

    def set_gru(self, w_ih, w_hh, b_ih, b_hh):
        a = [_f32(x) for x in (w_ih, w_hh, b_ih, b_hh)]
        self._chk(self.lib.mi_set_gru(self._ctx, *[_fp(x) for x in a]))

    def rec_state(self, hidden=None, done=None):
        hidden = None if hidden is None else _f32(hidden)
        done = None if done is None else _f32(done)
        self._chk(self.lib.mi_rec_state(self._ctx, _fp(hidden), _fp(done)))

    def get_hidden(self):
        out = np.empty((self.E, self.H), np.float32)
        self._chk(self.lib.mi_get_hidden(self._ctx, _fp(out)))
        return out

    def forward_rec(self, obs):
        want = np.uint8 if self.arch == ARCH_IMPALA else np.float32
        obs = np.ascontiguousarray(obs, dtype=want)
        lp, val, hid = np.empty((self.E, self.A), np.float32), np.empty(self.E, np.float32), np.empty((self.E, self.H), np.float32)
        self._chk(self.lib.mi_forward_rec(self._ctx, _fp(obs), _fp(lp), _fp(val), _fp(hid)))
        return lp, val, hid

    def forward(self, obs, want_feat=False):
        want = np.uint8 if self.arch == ARCH_IMPALA else np.float32
        obs = np.ascontiguousarray(obs, dtype=want)
        n = obs.shape[0]
        lp = np.empty((n, self.A), np.float32)
        val = np.empty(n, np.float32)
        feat = np.empty((n, self.H), np.float32) if want_feat else None
        self._chk(self.lib.mi_forward(self._ctx, _fp(obs), C.c_int32(n), _fp(lp), _fp(val), _fp(feat)))
        return (lp, val, feat) if want_feat else (lp, val)

    # ------------------------------------------------------------------ estimates
    def compute_estimates(self, gamma, lmbda, use_gae=True, normalize_adv=True):
        self._chk(self.lib.mi_compute_estimates(self._ctx, C.c_float(gamma), C.c_float(lmbda), C.c_int32(int(use_gae)),
                                                C.c_int32(int(normalize_adv))))

    def adv_stats(self):
        s = (C.c_double * 3)()
        self._chk(self.lib.mi_adv_stats(self._ctx, s))
        return np.array(list(s), dtype=np.float64)

    def adv_apply(self, stats3):
        s = (C.c_double * 3)(*[float(x) for x in stats3])
        self._chk(self.lib.mi_adv_apply(self._ctx, s))

    # ------------------------------------------------------------------ optimisation
    @staticmethod
    def hparams(eps_clip=0.2, value_coef=0.5, entropy_coef=0.01, x_entropy_coef=0.0, entropy_multiplier=1.0, fs_coef=0.0):
        return _HParams(eps_clip, value_coef, entropy_coef, x_entropy_coef, entropy_multiplier, fs_coef)

    def minibatch(self, idx, n_global, hp):
        idx = np.ascontiguousarray(idx, dtype=np.int64)
        self._chk(self.lib.mi_minibatch(self._ctx, _fp(idx), C.c_int32(idx.size), C.c_int32(n_global), C.byref(hp)))

    def minibatch_multi(self, idx, seg_n, n_global, hp):
        idx = np.ascontiguousarray(idx, dtype=np.int64)
        seg = np.ascontiguousarray(seg_n, dtype=np.int32)
        self._chk(self.lib.mi_minibatch_multi(self._ctx, _fp(idx), C.c_int32(idx.size), seg.ctypes.data_as(C.c_void_p), C.c_int32(seg.size),
                                              C.c_int32(n_global), C.byref(hp)))

    def minibatch_finish(self):
        self._chk(self.lib.mi_minibatch_finish(self._ctx))

    def loss_log_finalize(self):
        self._chk(self.lib.mi_loss_log_finalize(self._ctx))

    def set_multirank(self, enabled):
        self._chk(self.lib.mi_set_multirank(self._ctx, C.c_int32(int(enabled))))

    def optimizer_step(self, lr, max_grad_norm, adam_step, want_norm=False):
        g = C.c_float(0)
        self._chk(self.lib.mi_optimizer_step(self._ctx, C.c_float(lr), C.c_float(max_grad_norm), C.c_int32(adam_step),
                                             C.byref(g) if want_norm else None))
        return g.value if want_norm else None

    def loss_log(self, reset=True, max_records=4096):
        out = np.empty((max_records, 8), np.float32)
        n = C.c_int32(0)
        self._chk(self.lib.mi_loss_log_read(self._ctx, _fp(out), C.c_int32(max_records), C.byref(n), C.c_int32(int(reset))))
        return out[:n.value].copy()

    # ------------------------------------------------------------------ RCCL collectives (inside the library)
    def comm_unique_id(self):
        buf = (C.c_char * 128)()
        self._chk(self.lib.mi_comm_unique_id(buf, C.c_size_t(128)))
        return bytes(buf)

    def comm_init(self, id_bytes, rank, world):
        buf = (C.c_char * 128).from_buffer_copy(id_bytes)
        self._chk(self.lib.mi_comm_init(self._ctx, buf, C.c_size_t(128), C.c_int32(rank), C.c_int32(world)))
        self.comm_world = world

    def allreduce_arm(self):
        self._chk(self.lib.mi_allreduce_arm(self._ctx))

    def allreduce_grads(self):
        self._chk(self.lib.mi_allreduce_grads(self._ctx))

    def allreduce_buffer(self, which, n):
        self._chk(self.lib.mi_allreduce_buffer(self._ctx, C.c_int32(which), C.c_int64(n)))

    def adv_normalize_global(self):
        self._chk(self.lib.mi_adv_normalize_global(self._ctx))

    def device_ptr(self, which):
        p, n = C.c_void_p(), C.c_int64()
        self._chk(self.lib.mi_device_ptr(self._ctx, C.c_int32(which), C.byref(p), C.byref(n)))
        return p.value, n.value

    # ------------------------------------------------------------------ live kernel timing
    def profile_enable(self, mode=1):
        """0 off, 1 update-phase kernels only, 2 rollout + update."""
        self._chk(self.lib.mi_profile_enable(self._ctx, C.c_int32(int(mode))))

    def profile_read(self, reset=True):
        rows = np.zeros((64, 7), np.float64)
        n = C.c_int32(0)
        self._chk(self.lib.mi_profile_read(self._ctx, _fp(rows), C.c_int32(64), C.byref(n), C.c_int32(int(reset))))
        out = []
        for r in rows[:n.value]:
            out.append(dict(kernel=self.lib.mi_profile_class_name(C.c_int32(int(r[0]))).decode(), phase="rollout" if r[1] == 0 else "update",
                            launches=int(r[2]), ms=float(r[3]), samples=int(r[4]), bytes=float(r[5]), flops=float(r[6])))
        return out

    # ------------------------------------------------------------------ op-level (tests)
    def op_conv3x3(self, mode, cin, cout, hw, w_ref, inp=None, relu_in=False, bias=None, res=None, mask=None, dout=None):
        is_u8 = inp is not None and inp.dtype == np.uint8
        n = (inp if inp is not None else dout).shape[0]
        # modes: 0 forward, 1 dgrad, 2 wgrad; a block's first conv in bf16 precision also 3 = conv+maxpool forward,
        # 4 / 5 = weight / data gradient from the POOLED gradient (dout) and the forward's arg-max (inp is needed for it);
        # 6 / 7 = the fused data + weight gradient launch of block2.conv, returning (dW, db) / dx
        inp = None if inp is None else np.ascontiguousarray(inp)
        w_ref = _f32(w_ref)
        bias, res, mask, dout = (None if a is None else _f32(a) for a in (bias, res, mask, dout))
        if mode in (2, 4, 6):
            out, db = np.empty((cout, cin, 3, 3), np.float32), np.empty(cout, np.float32)
        elif mode == 3:
            out, db = np.empty((n, hw // 2, hw // 2, cout), np.float32), None
        elif mode in (5, 7):
            out, db = np.empty((n, hw, hw, cin), np.float32), None
        else:
            out, db = np.empty((n, hw, hw, cin if mode == 1 else cout), np.float32), None
        self._chk(self.lib.mi_op_conv3x3(self._ctx, C.c_int32(mode), C.c_int32(cin), C.c_int32(cout), C.c_int32(hw),
                                         C.c_int32(n), _fp(inp), C.c_int32(int(is_u8)), C.c_int32(int(relu_in)), _fp(w_ref),
                                         _fp(bias), _fp(res), _fp(mask), _fp(dout), _fp(out), _fp(db)))
        return (out, db) if mode in (2, 4, 6) else out

    def op_resblock(self, mode, x, w1, w2, b1=None, b2=None, a_fwd=None, x_fwd=None):
        """bf16 precision: fused residual block.  mode 0 -> (conv1 output, block output); mode 1 (x = dy) -> (d conv1-output, d block-input);
        mode 2 (16 channels @32x32, x = dy) -> (flat {dW1, db1, dW2, db2} in the head of the first array, d block-input)."""
        x = _f32(x)
        n, hw, _, ch = x.shape
        w1, w2 = _f32(w1), _f32(w2)
        b1, b2, a_fwd, x_fwd = (None if a is None else _f32(a) for a in (b1, b2, a_fwd, x_fwd))
        oa, oy = np.empty(max(x.size, 2 * (9 * ch * ch + ch)), np.float32), np.empty_like(x)      # mode 2 packs the weight gradients into oa
        self._chk(self.lib.mi_op_resblock(self._ctx, C.c_int32(mode), C.c_int32(ch), C.c_int32(hw), C.c_int32(n), _fp(x), _fp(w1), _fp(b1),
                                          _fp(w2), _fp(b2), _fp(a_fwd), _fp(x_fwd), _fp(oa), _fp(oy)))
        return (oa if mode == 2 else oa[:x.size].reshape(x.shape)), oy

    def op_maxpool(self, mode, x, dout=None):
        x = _f32(x)
        n, hw, _, c = x.shape
        dout = None if dout is None else _f32(dout)
        out = np.empty((n, hw // 2, hw // 2, c) if mode == 0 else x.shape, np.float32)
        self._chk(self.lib.mi_op_maxpool(self._ctx, C.c_int32(mode), C.c_int32(n), C.c_int32(hw), C.c_int32(c), _fp(x),
                                         _fp(dout), _fp(out)))
        return out

    def op_gemm(self, A, B, transpose_a=False, transpose_b=False):
        """C = op(A) @ op(B) with the operands left in place (strided access inside the kernel)."""
        A, B = _f32(A), _f32(B)
        M, K = (A.shape[1], A.shape[0]) if transpose_a else A.shape
        N = B.shape[0] if transpose_b else B.shape[1]
        sam, sak = (1, A.shape[1]) if transpose_a else (A.shape[1], 1)
        sbk, sbn = (1, B.shape[1]) if transpose_b else (B.shape[1], 1)
        out = np.empty((M, N), np.float32)
        self._chk(self.lib.mi_op_gemm(self._ctx, C.c_int32(M), C.c_int32(N), C.c_int32(K), _fp(A), C.c_int64(sam),
                                      C.c_int64(sak), _fp(B), C.c_int64(sbk), C.c_int64(sbn), _fp(out)))
        return out

    def debug_read(self, which, n):
        """Activation tensor `which` (see include/mi355ppo.h mi_debug_read) of the last minibatch pass, first n samples, fp32 NHWC."""
        if which == 100:
            out = np.empty((n, self.H), np.float32)
        else:
            b = which >> 3
            hw, ch = (32, 16, 8)[b], (16, 32, 32)[b]
            out = np.empty((n, hw, hw, ch), np.float32)
        self._chk(self.lib.mi_debug_read(self._ctx, C.c_int32(which), C.c_int32(n), _fp(out)))
        return out

    def minibatch_positions(self, gpos):
        """Global minibatch positions of the next minibatch()'s rows (fs_coef != 0 on several ranks, include/mi355ppo.h)."""
        g = np.ascontiguousarray(gpos, dtype=np.int32)
        self._chk(self.lib.mi_minibatch_positions(self._ctx, g.ctypes.data_as(C.c_void_p), C.c_int32(g.size)))

    def debug_philox(self, ctr_key6):
        """(n,6) uint32 {c0..c3,k0,k1} -> ((n,4) uint32 Philox4x32-10 output words, (n,) the sampler's uniforms for seed k0|k1<<32, counter c0|c1<<32)."""
        a = np.ascontiguousarray(ctr_key6, dtype=np.uint32).reshape(-1, 6)
        out, u = np.empty((a.shape[0], 4), np.uint32), np.empty(a.shape[0], np.float32)
        self._chk(self.lib.mi_debug_philox(self._ctx, _fp(a), C.c_int32(a.shape[0]), _fp(out), _fp(u)))
        return out, u

    def debug_step_latency(self, t, iters=200, graph=False):
        """microseconds per policy step of slot t (launches + stream wait), eager or as a replayed hipGraph of the same launches"""
        us = C.c_float(0)
        self._chk(self.lib.mi_debug_step_latency(self._ctx, C.c_int32(t), C.c_int32(iters), C.c_int32(int(graph)), C.byref(us)))
        return us.value

    def debug_flags(self, flags):
        self._chk(self.lib.mi_debug_flags(self._ctx, C.c_int32(flags)))

    def selftest_mfma(self):
        e = C.c_float(0)
        self._chk(self.lib.mi_selftest_mfma(self._ctx, C.byref(e)))
        return e.value
