"""PPO agent (reference: agents/ppo.py:9-279) on the MI355X engine.

Same constructor and public methods; what differs is where things live.  The rollout ring, the
parameters, Adam and every intermediate of the update are device-resident; per rollout step the host
uploads E uint8 frames (pinned, async) and downloads E actions; per minibatch it sends an index
vector.  No ``.item()`` per minibatch: loss records are appended to a device log and read once per
``optimize()``.  Reference quirks kept on purpose: unscaled gradient accumulation (ppo.py:170-177),
sign-flipped 'Loss/pi' / 'Loss/v' (:178-179), ``mini_batch_size`` shrunk to T*E/n_minibatch (:108-110),
validation rollouts that are never trained on (:241-252).
"""
import numpy as np
import torch

from common.misc_util import adjust_lr, adjust_lr_grok
from common.model import as_device_obs
from mi355.dist import Collective, DevicePointerTensor, update_plan
from mi355.engine import Engine, PTR_FS_KEYS, PTR_GRADS, PTR_LOSS_STATS, PTR_STATS_RING


def _with_next(it):
    """(item, next item or None) pairs: the update schedule needs one operation of look-ahead to arm the overlapped gradient exchange."""
    it = iter(it)
    prev = next(it, None)
    for cur in it:
        yield prev, cur
        prev = cur
    if prev is not None:
        yield prev, None
from mi355.optim import DeviceAdam
from .base_agent import BaseAgent


class PPO(BaseAgent):
    def __init__(self, env, policy, logger, storage, device, n_checkpoints, env_valid=None, storage_valid=None,
                 n_steps=128, n_envs=8, epoch=3, n_minibatch=8, mini_batch_size=32 * 8, gamma=0.99, lmbda=0.95,
                 learning_rate=2.5e-4, grad_clip_norm=0.5, eps_clip=0.2, value_coef=0.5, entropy_coef=0.01,
                 x_entropy_coef=0., normalize_adv=True, normalize_rew=True, use_gae=True, entropy_scaling=None,
                 increasing_lr=False, sparsity_coef=0., fs_coef=0., **kwargs):
        super().__init__(env, policy, logger, storage, device, n_checkpoints, env_valid, storage_valid)
        self.fs_coef = fs_coef
        self.total_timesteps = 0
        self.entropy_scaling = entropy_scaling
        self.entropy_multiplier = 1.
        self.s_loss_coef = sparsity_coef
        self.min_rew, self.max_rew = -1., 11.
        self.n_steps, self.n_envs = n_steps, n_envs
        self.epoch, self.n_minibatch, self.mini_batch_size = epoch, n_minibatch, mini_batch_size
        self.gamma, self.lmbda = gamma, lmbda
        self.learning_rate = learning_rate
        self.grad_clip_norm = grad_clip_norm
        self.eps_clip, self.value_coef, self.entropy_coef = eps_clip, value_coef, entropy_coef
        self.x_entropy_coef = x_entropy_coef
        self.normalize_adv, self.normalize_rew, self.use_gae = normalize_adv, normalize_rew, use_gae
        self.adjust_lr = adjust_lr_grok if increasing_lr else adjust_lr
        self.seed = int(kwargs.get("seed", 0))
        self.merge_accumulation = bool(kwargs.get("merge_accumulation", True))      # (new) see optimize()
        self.joint_rollouts = bool(kwargs.get("joint_rollouts", True))              # (new) training + validation rollout in one host loop (train())
        # train.py --detect_nan (reference: autograd anomaly mode + a NaN hook on every module's output, train.py:123-124,234-250):
        # here the policy outputs of every rollout step, every minibatch's loss record and every gradient norm are checked
        self.detect_nan = bool(kwargs.get("detect_nan", False))
        # activation storage of the IMPALA path: "fp32" (parity mode) or "bf16" (BASELINE config 3)
        self.precision = kwargs.get("precision", "fp32") if policy.arch == "impala" else "fp32"

        # ---- data parallel over n_envs: this process owns n_envs envs; the global minibatch spans all ranks
        self.coll = Collective()
        if fs_coef != 0. and self.coll.active and policy.arch != "impala":
            fs_coef = self.fs_coef = 0.                            # (no feature-sparsity term without the IMPALA feature map, model.py:976-977)
        self.n_envs_global = n_envs * self.coll.world
        n_total = n_steps * self.n_envs_global
        batch_size = n_total // n_minibatch
        max_local = min(mini_batch_size, batch_size)               # worst case for ONE minibatch: all of it in this rank's shard
        if self.merge_accumulation and batch_size > max_local and x_entropy_coef == 0 and fs_coef == 0:
            # accumulated minibatches are processed together (optimize()): room for this rank's expected share of the samples
            # between two optimizer steps (+7 % + 64: > 6 sigma of the split of a random permutation over the ranks), capped
            share = -(-batch_size // self.coll.world)
            max_local = max(max_local, min(16384 if self.precision == "bf16" else 8192, int(share * 1.07) + 64, n_steps * n_envs))
        dev_index = device.index if isinstance(device, torch.device) and device.index is not None else 0
        arch = policy.arch
        emb = policy.embedder
        # with >1 rank the engine issues on a torch-owned stream, so that torch.distributed collectives on the
        # aliased gradient buffer are ordered against the kernels that produce / consume it
        self._tstream = torch.cuda.Stream(device=dev_index) if self.coll.active else None
        self.engine = Engine(arch, n_steps, n_envs, policy.action_size, max_batch=max(max_local, n_envs),
                             obs_dim=getattr(emb, "input_size", 0), mlp_depth=getattr(emb, "depth", 0),
                             mlp_width=getattr(emb, "mid_weight", 0), out_dim=emb.output_dim, device=dev_index,
                             stream=self._tstream.cuda_stream if self._tstream is not None else None,
                             precision=self.precision)
        policy.attach_engine(self.engine)
        storage.attach_engine(self.engine)
        self.engine_valid = None
        if storage_valid is not None:
            # inference-only twin for the validation rollouts (shares nothing but the weights it is handed)
            self.engine_valid = Engine(arch, n_steps, n_envs, policy.action_size, max_batch=n_envs,
                                       obs_dim=getattr(emb, "input_size", 0), mlp_depth=getattr(emb, "depth", 0),
                                       mlp_width=getattr(emb, "mid_weight", 0), out_dim=emb.output_dim, device=dev_index,
                                       precision=self.precision)
            storage_valid.attach_engine(self.engine_valid)
            policy.attach_aux_engine(self.engine_valid)          # frozen GRU weights, now and after load_state_dict
        self.optimizer = DeviceAdam(policy, self.engine, learning_rate, eps=1e-5)
        self._grads_t = self._stats_t = self._ring_t = None
        # collectives: RCCL inside the library when the process group is "nccl" (one GPU per rank); torch.distributed on aliased
        # buffers otherwise (gloo: CPU tests / rehearsal)
        self._native = self.coll.attach_native(self.engine) if self.coll.active else False
        import os
        self._arm = self._native and os.environ.get("MI355_NATIVE_ARM", "1") != "0"      # (A/B switch of the overlapped exchange, tests)
        if self.coll.active:
            self.engine.set_multirank(True)
            gp, gn = self.engine.device_ptr(PTR_GRADS)
            sp, sn = self.engine.device_ptr(PTR_LOSS_STATS)
            self._grads_t = DevicePointerTensor(gp, gn).tensor(dev_index)
            self._stats_t = DevicePointerTensor(sp, sn).tensor(dev_index)
            rp, rn = self.engine.device_ptr(PTR_STATS_RING)
            self._ring_t = DevicePointerTensor(rp, rn).tensor(dev_index)
            self._fskeys_t = None
            if fs_coef != 0. and arch == "impala":                 # per-column candidates, max-all-reduced before the backward pass
                kp, kn = self.engine.device_ptr(PTR_FS_KEYS)
                self._fskeys_t = DevicePointerTensor(kp, kn, "<i8").tensor(dev_index)
        self._stage = [self.engine.pinned((n_envs,) + self._obs_stage_shape(arch, emb), self._obs_dtype(arch)) for _ in range(2)]
        self._stage_i = 0
        self._gstage = {}
        self._iter = 0

    @staticmethod
    def _obs_stage_shape(arch, emb):
        return (64, 64, 3) if arch == "impala" else (emb.input_size,)

    @staticmethod
    def _obs_dtype(arch):
        return np.uint8 if arch == "impala" else np.float32

    # ------------------------------------------------------------------ predict
    def _predict_into(self, engine, storage, t, obs, hidden_state=None, done=None):
        """Upload obs into ring slot t (pinned double buffer), forward (+ GRU cell) + sample on the device."""
        buf = self._stage[self._stage_i]
        self._stage_i ^= 1
        buf[...] = as_device_obs(obs, self.policy.arch)
        engine.put_obs(t, buf)
        rec = self.policy.is_recurrent()
        if rec:
            engine.rec_state(hidden_state, done)
        act, logp, value = engine.rollout_step(t, seed=self.seed * 1000003 + self._iter)
        if self.detect_nan and not (np.isfinite(value).all() and np.isfinite(logp).all()):
            raise RuntimeError(f"Found NaN / Inf in the policy outputs of rollout step {t}")
        storage.note_predicted(t, obs, act, logp, value)
        return act, logp, value, (engine.get_hidden() if rec else hidden_state)

    def predict(self, obs, hidden_state, done):
        """agents/ppo.py:72-81.  The observation is staged on the device; the ``Storage.store`` /
        ``store_last`` call that follows with the same array commits it into its slot on the device, so the
        frames cross PCIe once."""
        self._predict_calls = getattr(self, "_predict_calls", 0) + 1
        rec = self.policy.is_recurrent()
        if rec:
            self.engine.rec_state(hidden_state, done)
        act, logp, value = self.engine.predict_staged(as_device_obs(obs, self.policy.arch),
                                                      seed=self.seed * 1000003 + self._iter,
                                                      counter=self._predict_calls * self.n_envs)
        self.storage.note_predicted(-1, obs, act, logp, value)
        return act, logp, value, (self.engine.get_hidden() if rec else np.asarray(hidden_state))

    def predict_w_value_saliency(self, obs, hidden_state, done):
        """agents/ppo.py:83-94: predict + the gradient of the value with respect to the observation, in the observation's
        layout ((E,3,64,64) for frames).  Each env's value depends on its own observation only, so this is the gradient of
        sum_e value_e (the reference's value.backward() needs n_envs = 1, as render.py uses it)."""
        rec = self.policy.is_recurrent()
        if rec:                                                   # the value then depends on obs through the GRU cell as well (model.py:219-225)
            self.engine.rec_state(hidden_state, done)
        self._predict_calls = getattr(self, "_predict_calls", 0) + 1
        act, logp, value, grad = self.engine.value_saliency(as_device_obs(obs, self.policy.arch),
                                                            seed=self.seed * 1000003 + self._iter,
                                                            counter=self._predict_calls * self.n_envs)
        self.storage.note_predicted(-1, obs, act, logp, value)
        if self.policy.arch == "impala":
            grad = np.ascontiguousarray(grad.transpose(0, 3, 1, 2))          # NHWC -> the reference's (E,3,64,64)
        return act, logp, value, (self.engine.get_hidden() if rec else np.asarray(hidden_state)), grad

    # ------------------------------------------------------------------ optimize
    def _hparams(self):
        return self.engine.hparams(self.eps_clip, self.value_coef, self.entropy_coef, self.x_entropy_coef,
                                   self.entropy_multiplier, self.fs_coef)

    def optimize(self):
        if self.entropy_scaling == "reward_based":
            mean_rew = np.mean(self.logger.episode_reward_buffer)
            self.entropy_multiplier = 1 - ((mean_rew - self.min_rew) / (self.max_rew - self.min_rew))
        elif self.entropy_scaling == "time_based":
            self.entropy_multiplier = 1 - (self.t / self.total_timesteps)

        n_total = self.n_steps * self.n_envs_global
        batch_size = n_total // self.n_minibatch
        if batch_size < self.mini_batch_size:
            self.mini_batch_size = batch_size
        grad_accumulation_steps = batch_size / self.mini_batch_size
        eng, coll, hp = self.engine, self.coll, self._hparams()
        recurrent = self.policy.is_recurrent()
        # Multi-rank: the backward pass needs a cross-rank statistic (the batch-mean action distribution) only for the x-entropy
        # term; without it (and without the feature-sparsity term) every rank runs its minibatches straight through and the
        # logged loss sums are reduced over the ranks ONCE per optimize() instead of once per minibatch.
        no_batch_terms = self.x_entropy_coef == 0 and self.fs_coef == 0
        deferred = coll.active and no_batch_terms
        # Gradient accumulation (grad_accumulation_steps > 1: the gradients of that many minibatches are summed before one optimizer
        # step, :170-177) without batch-level loss terms is one sum over all their samples: this rank's shares of the accumulated
        # minibatches go through the network in ONE pass (mi_minibatch_multi; losses still taken and logged per minibatch).  On R
        # ranks a share is ~1/R of a minibatch, so this keeps every launch at single-GPU size instead of R-fold smaller.
        merge = self.merge_accumulation and no_batch_terms and grad_accumulation_steps > 1
        if coll.active:
            eng.set_multirank(2 if deferred else 1)

        def chunks():
            for _ in range(self.epoch):
                yield from self.storage.minibatch_index_stream(self.mini_batch_size, recurrent, self.n_envs_global)

        native = self._native
        # fs_coef != 0 on > 1 rank: the rows' positions in the global minibatch travel with every pass (ties between equal column maxima go
        # to the globally first row) and the per-column candidates are max-all-reduced next to the loss statistics
        fs_global = coll.active and self.fs_coef != 0 and self.policy.arch == "impala"
        plan = update_plan(chunks(), coll.rank, coll.world, self.n_envs_global, grad_accumulation_steps, merge,
                           coll.active and not deferred, eng.max_batch, with_positions=fs_global)
        for op, nxt in _with_next(plan):
            if op[0] == "minibatch":
                _, local, seg_n, n_global = op[:4]
                if fs_global:
                    eng.minibatch_positions(op[4])
                # the LAST pass before an optimizer step hands its gradient regions to the side stream as they become final
                # (mi_allreduce_arm); with a per-minibatch statistics exchange the backward pass runs inside minibatch_finish (below)
                if self._arm and nxt is not None and nxt[0] == "step":
                    eng.allreduce_arm()
                if len(seg_n) > 1 or merge:
                    eng.minibatch_multi(local, seg_n, n_global, hp)
                else:
                    eng.minibatch(local, n_global, hp)
            elif op[0] == "stats":
                if native:
                    eng.allreduce_buffer(PTR_LOSS_STATS, 32)
                    if fs_global:
                        eng.allreduce_buffer(PTR_FS_KEYS, 2048)
                else:
                    with torch.cuda.stream(self._tstream):
                        coll.allreduce_sum_(self._stats_t)       # 32 floats: loss sums + mean action probabilities
                        if fs_global:
                            coll.allreduce_max_(self._fskeys_t)  # 2048 int64 (value bits, position) candidates
                if self._arm and nxt is not None and nxt[0] == "step":
                    eng.allreduce_arm()
                eng.minibatch_finish()
            elif op[0] == "step":
                if native:
                    eng.allreduce_grads()                    # whatever the armed pass has not sent yet (normally nothing)
                elif coll.active:
                    with torch.cuda.stream(self._tstream):
                        coll.allreduce_sum_(self._grads_t)   # ONE collective per optimizer step: the flat gradient
                gn = self.optimizer.step(self.grad_clip_norm, want_norm=self.detect_nan)
                if self.detect_nan and not np.isfinite(gn):
                    raise RuntimeError(f"Found NaN / Inf in the gradient norm of optimizer step {self.optimizer.step_count}: {gn}")
            elif deferred:                                   # ("log", n): the statistics ring, once per optimize()
                if native:
                    eng.allreduce_buffer(PTR_STATS_RING, 32 * op[1])
                else:
                    with torch.cuda.stream(self._tstream):
                        coll.allreduce_sum_(self._ring_t[:32 * op[1]])
                eng.loss_log_finalize()
        log = eng.loss_log(reset=True)
        if self.detect_nan and not np.isfinite(log[:, :5]).all():
            bad = np.argwhere(~np.isfinite(log[:, :5]))
            raise RuntimeError(f"Found NaN / Inf in the loss terms (minibatch, term) {bad[:8].tolist()} of this update")
        nan = float("nan")
        fs = float(np.mean(log[:, 5])) if self.policy.arch == "impala" else nan
        return {'Loss/pi': float(np.mean(-log[:, 0])), 'Loss/v': float(np.mean(-log[:, 1])),
                'Loss/entropy': float(np.mean(log[:, 2])), 'Loss/x_entropy': float(np.mean(log[:, 3])),
                'Loss/atn_entropy': nan, 'Loss/atn_entropy2': nan, 'Loss/sparsity': nan,
                'Loss/feature_sparsity': fs, 'Loss/total': float(np.mean(log[:, 4]))}

    def draw_permutation_ahead(self):
        """The first epoch's index permutation of the NEXT optimize() -- torch.randperm(T * E_global), or (E_global) for a recurrent policy
        (common/storage.py:86-110) -- drawn by a helper thread from now on; same numbers as the in-place draw (Storage.draw_permutation_ahead)."""
        n = self.n_envs_global if self.policy.is_recurrent() else self.n_steps * self.n_envs_global
        self.storage.draw_permutation_ahead(n)

    # ------------------------------------------------------------------ rollout + train
    def _collect(self, env, engine, storage, obs, hidden_state, done):
        if self._can_pipeline(env):
            return self._collect_pipelined(env, engine, storage, obs, hidden_state, done)
        for _ in range(self.n_steps):
            t = storage.step
            act, logp, value, next_hidden = self._predict_into(engine, storage, t, obs, hidden_state, done)
            next_obs, rew, done, info = env.step(act)
            storage.store(obs, hidden_state, act, rew, done, info, logp, value)
            obs, hidden_state = next_obs, next_hidden
        _, _, last_val, hidden_state = self._predict_into(engine, storage, self.n_steps, obs, hidden_state, done)
        storage.store_last(obs, hidden_state, last_val)
        return obs, hidden_state, done

    def _collect_pipelined(self, env, engine, storage, obs, hidden_state, done):
        """The same T steps + bootstrap step with the env groups of `env.env_groups` as independent chains (mi_rollout_submit /
        mi_rollout_wait): while the host waits for / steps group g, the other groups' frame uploads and forward passes run on
        their own streams.  Device ring and returned arrays are those of `_collect` (same kernels, same sampling counters).

        Host work per group step is kept to what the chain needs: a group's frames go up FROM THE ENV'S OWN BUFFER when that can be
        page-locked in place (Engine.dma_ready: uint8 NHWC arrays an env hands out again and again -- Procgen's rgb buffer, a pool),
        and only otherwise through a copy into a pinned staging buffer (786 KB per group step at E = 256: tens of microseconds of one
        core inside a ~105 us dependency chain); results are written into this call's arrays through raw addresses; a step's infos
        stay per-group objects (StepInfo.join), nothing is walked entry by entry."""
        return self._collect_lanes([(env, engine, storage, obs, hidden_state, done)])[0]

    def _collect_lanes(self, lanes):
        """Several rollouts side by side -- `lanes` = [(env with .env_groups, engine, storage, obs, hidden_state, done), ...], each on its own
        engine -- in ONE host loop: every (lane, group) is an independent chain.  One lane is the pipelined collector of PPO.train; two
        lanes are the training and the validation rollout of an iteration (agents/ppo.py:225-252 runs them one after the other on the same
        policy; neither depends on the other): both are latency chains that leave the GPU mostly idle, so interleaved they take little
        longer than one alone."""
        from common.env.vec_envs import StepInfo
        T, arch = self.n_steps, self.policy.arch
        want_dtype = self._obs_dtype(arch)

        class Lane:
            pass
        Ls = []
        for k, (env, engine, storage, obs, hidden_state, done) in enumerate(lanes):
            L = Lane()
            L.groups, L.engine, L.storage, L.hidden = env.env_groups, engine, storage, hidden_state
            L.G, E = len(L.groups), self.n_envs
            ng = E // L.G
            if getattr(engine, "n_groups", 1) != L.G:
                engine.rollout_groups(L.G)
            L.st = self._gstage.setdefault((id(engine), L.G), [[engine.pinned((ng,) + self._obs_stage_shape(arch, self.policy.embedder), want_dtype)
                                                                for _ in range(2)] for _ in range(L.G)])
            L.obs_g = [obs[g * ng:(g + 1) * ng] for g in range(L.G)]
            L.rew = np.zeros(E, np.float32); L.dn = np.zeros(E, np.float32)
            L.act = np.zeros(E, np.int64); L.logp = np.zeros(E, np.float32); L.val = np.zeros(E, np.float32)
            L.pa, L.pl, L.pv = (a.__array_interface__['data'][0] for a in (L.act, L.logp, L.val))
            L.sls = [slice(g * ng, (g + 1) * ng) for g in range(L.G)]
            L.infos = [None] * L.G
            # the validation lane draws from its own Philox stream (the reference's two rollouts share torch's generator, not its numbers)
            L.seed = self.seed * 1000003 + self._iter + (k << 40)
            L.wait, L.submit, L.ready = engine.rollout_wait_into, engine.rollout_submit, engine.dma_ready
            Ls.append(L)
        for t in range(T + 1):
            for g in range(max(L.G for L in Ls)):
                for L in Ls:
                    if g >= L.G:
                        continue
                    sl = L.sls[g]
                    if t:
                        L.wait(g, L.pa + 8 * sl.start, L.pl + 4 * sl.start, L.pv + 4 * sl.start)
                        o, r, d, L.infos[g] = L.groups[g].step(L.act[sl])
                        L.obs_g[g] = o
                        L.rew[sl] = r; L.dn[sl] = d
                    else:
                        o = L.obs_g[g]
                    # upload in place when the env's array already is what the ring stores, else convert / copy into the pinned stage
                    if t and isinstance(o, np.ndarray) and o.dtype == want_dtype and o.flags.c_contiguous and (arch != "impala" or o.shape[-1] == 3) and L.ready(o):
                        buf = o
                    else:
                        buf = L.st[g][t & 1]
                        buf[...] = as_device_obs(o, arch)
                    L.submit(t, g, buf, L.rew[sl] if t else None, L.dn[sl] if t else None, seed=L.seed)
            if t:
                for L in Ls:
                    if self.detect_nan and not (np.isfinite(L.val).all() and np.isfinite(L.logp).all()):
                        raise RuntimeError(f"Found NaN / Inf in the policy outputs of rollout step {t - 1}")
                    L.storage.note_stored(L.rew, L.dn, StepInfo.join(L.infos))       # (hidden-state mirror: all zeros for a non-recurrent policy, left as is)
        out = []
        for L in Ls:
            for g in range(L.G):
                L.wait(g, L.pa + 8 * L.sls[g].start, L.pl + 4 * L.sls[g].start, L.pv + 4 * L.sls[g].start)
            L.storage._hidden[T] = L.hidden                  # store_last: value[T] and the frames are already in the device ring
            out.append((np.concatenate(L.obs_g), L.hidden, L.dn.copy()))
        return out

    def _can_pipeline(self, env):
        return len(getattr(env, "env_groups", ())) > 1 and not self.policy.is_recurrent() and self.n_envs % len(env.env_groups) == 0

    def train(self, num_timesteps):
        self.total_timesteps = num_timesteps
        save_every = num_timesteps // self.num_checkpoints
        checkpoints = sorted((i + 1) * save_every for i in range(self.num_checkpoints))
        checkpoint_cnt = 0
        obs = self.env.reset()
        hidden_state = np.zeros((self.n_envs, self.storage.hidden_state_size))
        done = np.zeros(self.n_envs)
        if self.env_valid is not None:
            obs_v = self.env_valid.reset()
            hidden_state_v = np.zeros((self.n_envs, self.storage.hidden_state_size))
            done_v = np.zeros(self.n_envs)
        steps_per_iter = self.n_steps * self.n_envs_global

        while self.t < num_timesteps:
            self._iter += 1
            if self.env_valid is not None:
                self.engine_valid.copy_params_from(self.engine)          # device to device (mi_copy_params): both rollouts run the current policy
            if (self.env_valid is not None and self.joint_rollouts and self._can_pipeline(self.env) and self._can_pipeline(self.env_valid)
                    and len(self.env.env_groups) + len(self.env_valid.env_groups) <= 4):     # (more than four busy streams serialise: 8 chains 86-94 ms)
                # training and validation rollout as chains of ONE host loop (neither depends on the other; ppo.py:225-252 runs them in turn):
                # at E = 256 with 2 + 2 env groups 49.5 ms for both, against 28.4 + 28.5 ms one after the other with 4 groups each
                # (scratch/valid_rollout_time.py)
                (obs, hidden_state, done), (obs_v, hidden_state_v, done_v) = self._collect_lanes(
                    [(self.env, self.engine, self.storage, obs, hidden_state, done),
                     (self.env_valid, self.engine_valid, self.storage_valid, obs_v, hidden_state_v, done_v)])
                self.storage.compute_estimates(self.gamma, self.lmbda, self.use_gae, self.normalize_adv, self.coll)
                self.storage_valid.compute_estimates(self.gamma, self.lmbda, self.use_gae, self.normalize_adv)
            else:
                obs, hidden_state, done = self._collect(self.env, self.engine, self.storage, obs, hidden_state, done)
                self.storage.compute_estimates(self.gamma, self.lmbda, self.use_gae, self.normalize_adv, self.coll)
                if self.env_valid is not None:
                    obs_v, hidden_state_v, done_v = self._collect(self.env_valid, self.engine_valid, self.storage_valid,
                                                                  obs_v, hidden_state_v, done_v)
                    self.storage_valid.compute_estimates(self.gamma, self.lmbda, self.use_gae, self.normalize_adv)
            summary = self.optimize()
            self.t += steps_per_iter
            rew_batch, done_batch, true_average_reward = self.storage.fetch_log_data()
            if (done_batch > 0).any():
                print(f"Mean Reward:{np.mean(rew_batch[done_batch > 0]):.2f}")
            if self.storage_valid is not None:
                rew_batch_v, done_batch_v, true_average_reward_v = self.storage_valid.fetch_log_data()
            else:
                rew_batch_v = done_batch_v = true_average_reward_v = None
            self.logger.feed(rew_batch, done_batch, true_average_reward, rew_batch_v, done_batch_v, true_average_reward_v)
            self.optimizer, lr = self.adjust_lr(self.optimizer, self.learning_rate, self.t, num_timesteps)
            self.logger.dump(summary, lr)
            self.draw_permutation_ahead()                    # the next update's first permutation, behind the next rollout
            if checkpoint_cnt < len(checkpoints) and self.t > checkpoints[checkpoint_cnt]:
                if self.coll.rank == 0:
                    print("Saving model.")
                    # the reference's two keys (agents/ppo.py:271-276) + what it forgets and a resume needs: the step counter, the
                    # learning rate and the reward normaliser's running variance (extra keys are ignored by the reference's loaders)
                    rs = getattr(self.env, "reward_state", None)
                    torch.save({'model_state_dict': self.policy.state_dict(),
                                'optimizer_state_dict': self.optimizer.state_dict(),
                                't': int(self.t), 'learning_rate': float(lr), 'reward_norm': rs() if callable(rs) else None},
                               self.logger.logdir + '/model_' + str(self.t) + '.pth')
                checkpoint_cnt += 1
        self.env.close()
        if self.env_valid is not None:
            self.env_valid.close()
