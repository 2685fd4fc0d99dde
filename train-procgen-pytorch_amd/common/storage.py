"""Storage (reference: common/storage.py:7-162) with the rollout resident in HBM.

The reference keeps (T+1,E,3,64,64) fp32 observations (3.2 GB at hard-500) plus the scalars on the
host and gathers + uploads every minibatch.  Here the ring lives in the engine: uint8 NHWC frames
(12 288 B per frame, 4x smaller than fp32) and fp32/int32 scalars; a minibatch is only an index
vector.  The public surface (constructor, store/store_last/compute_estimates/fetch_train_generator/
collate_data/fetch_log_data, the *_batch attributes) is the reference's; the *_batch attributes read
back from the device on access.
"""
from collections import deque

import numpy as np
import torch

from mi355 import engine as M
from .model import as_device_obs


def _randperm_serial(n):
    """torch.randperm(n) on the global CPU generator, drawn with ONE intra-op thread: the permutation is a serial shuffle whose
    result does not depend on the thread count, but with the default pool (every core the host shows, whatever the cgroup grants)
    the call took milliseconds (60 ms in an 8-CPU container) instead of 0.6 ms for n = 65536 - with the GPU idle behind it at the
    head of every optimize()."""
    k = torch.get_num_threads()
    torch.set_num_threads(1)
    try:
        return torch.randperm(n).numpy()
    finally:
        torch.set_num_threads(k)


class _PermutationAhead:
    """The first epoch's torch.randperm(n) of the NEXT update, drawn by a helper thread while the rollout runs (PPO.train starts it at
    the end of an iteration).  At the head of optimize() the GPU is idle behind that draw: 0.6 ms for n = 65 536, 8 ms for the 524 288
    indices of an 8-GPU run (9 % of an iteration).  The numbers are the ones an in-place draw would give -- same generator, same
    position -- provided nothing touched the generator in between; that is checked (state after the draw == state at use), and a
    stale permutation (a caller re-seeded, as the tests do) is dropped in favour of a fresh draw."""

    def __init__(self, n):
        import threading
        self.n, self.perm, self.state_after = n, None, None
        self._t = threading.Thread(target=self._run, daemon=True)
        self._t.start()

    def _run(self):
        self.perm = _randperm_serial(self.n)
        self.state_after = torch.get_rng_state()

    def take(self, n):
        self._t.join()
        if n == self.n and self.perm is not None and torch.equal(self.state_after, torch.get_rng_state()):
            return self.perm
        return None


class Storage:
    def __init__(self, obs_shape, hidden_state_size, num_steps, num_envs, device, continuous_actions=False,
                 act_shape=None):
        if continuous_actions:
            raise NotImplementedError("continuous actions are not part of the accelerated PPO path")
        self.continuous_actions = False
        self.performance_track = {}
        self.obs_shape = tuple(obs_shape)
        self.act_shape = act_shape
        self.hidden_state_size = hidden_state_size
        self.num_steps = num_steps
        self.num_envs = num_envs
        self.device = device
        self.engine = None
        self._perm_ahead = None
        self.arch = "impala" if len(self.obs_shape) == 3 else "mlp"
        # host mirrors of reward / done: what Logger and fetch_log_data read (pinned once an engine is attached)
        self._rew = np.zeros((num_steps, num_envs), np.float32)
        self._done = np.zeros((num_steps, num_envs), np.float32)
        self.reset()

    # ------------------------------------------------------------------ engine plumbing
    def attach_engine(self, engine):
        if (engine.T, engine.E) != (self.num_steps, self.num_envs):
            raise ValueError("engine (T,E) does not match the storage")
        self.engine = engine
        # pinned host mirrors: async DMA source for rew/done and the data Logger / fetch_log_data read
        self._rew = engine.pinned((self.num_steps, self.num_envs), np.float32)
        self._done = engine.pinned((self.num_steps, self.num_envs), np.float32)
        self._rew[:] = 0
        self._done[:] = 0

    def _eng(self):
        if self.engine is None:
            raise RuntimeError("Storage is not attached to an engine yet: construct agents.ppo.PPO with it first")
        return self.engine

    def reset(self):
        self.info_batch = deque(maxlen=self.num_steps)
        self.step = 0
        self._pending = None
        self._hidden = np.zeros((self.num_steps + 1, self.num_envs, self.hidden_state_size), np.float32)

    def note_predicted(self, t, obs, act, logp, value):
        """Called by the agent after a policy step.  t >= 0: ring slot t already holds obs and the policy outputs
        (fast path of PPO.train).  t == -1: they are staged on the device (public PPO.predict) and the next
        store / store_last commits them into its slot."""
        self._pending = (t, obs, act, logp, value)

    def _claim(self, t, obs):
        """-> None (nothing usable), 'slot' (already in place) or 'staged' (committed into slot t now)."""
        p = self._pending
        if p is None or p[1] is not obs:
            return None
        if p[0] == t:
            return "slot"
        if p[0] == -1:
            self.engine.commit_staged(t)
            return "staged"
        return None

    # ------------------------------------------------------------------ reference API: writes
    def store(self, obs, hidden_state, act, rew, done, info, log_prob_act, value):
        eng, t = self._eng(), self.step
        pend = self._pending if self._claim(t, obs) else None
        if pend is None:
            self._keep = eng.put_obs(t, as_device_obs(obs, self.arch))
        if pend is None or pend[2] is not act or pend[3] is not log_prob_act or pend[4] is not value:
            eng.put_policy_outputs(t, np.asarray(act), np.asarray(log_prob_act), np.asarray(value))
        self._rew[t] = rew
        self._done[t] = done
        eng.put_step(t, self._rew[t], self._done[t])
        self._hidden[t] = hidden_state
        self.info_batch.append(info)
        self._pending = None
        self.step = (self.step + 1) % self.num_steps

    def note_stored(self, rew, done, info, hidden_state=None):
        """Host half of store() for the pipelined collector: the device ring already holds this step's frames (uploaded by
        mi_rollout_submit), policy outputs (written by the step's kernels) and reward / done (handed over with the next submit); only
        the host mirrors that Logger / fetch_log_data read are filled in."""
        t = self.step
        self._rew[t] = rew
        self._done[t] = done
        if hidden_state is not None:
            self._hidden[t] = hidden_state
        self.info_batch.append(info)
        self._pending = None
        self.step = (self.step + 1) % self.num_steps

    def store_last(self, last_obs, last_hidden_state, last_value):
        eng, T = self._eng(), self.num_steps
        pend = self._pending if self._claim(T, last_obs) else None
        if pend is None:
            self._keep = eng.put_obs(T, as_device_obs(last_obs, self.arch))
        if pend is None or pend[4] is not last_value:
            eng.put_policy_outputs(T, None, None, np.asarray(last_value))
        self._hidden[T] = last_hidden_state
        self._pending = None

    def compute_estimates(self, gamma=0.99, lmbda=0.95, use_gae=True, normalize_adv=True, collective=None):
        eng = self._eng()
        if collective is None or not collective.active or not normalize_adv:
            eng.compute_estimates(gamma, lmbda, use_gae, normalize_adv)
            return
        from mi355.dist import merge_adv_stats
        eng.compute_estimates(gamma, lmbda, use_gae, False)
        if getattr(eng, "comm_world", 0) > 1:
            eng.adv_normalize_global()                       # RCCL inside the library: all-gather + Chan merge + apply on the device
        else:
            eng.adv_apply(merge_adv_stats(collective.allgather_f64(eng.adv_stats())))

    # ------------------------------------------------------------------ index streams (bit-exact with the reference)
    def minibatch_index_stream(self, mini_batch_size=None, recurrent=False, n_envs_global=None):
        """Flat indices i = t*E + e of each minibatch of one epoch, consuming the global torch CPU generator
        exactly as the reference does: ONE torch.randperm(T*E) (SubsetRandomSampler inside BatchSampler with
        drop_last, storage.py:86-91), or torch.randperm(E) + env groups when recurrent (storage.py:93-110)."""
        E = self.num_envs if n_envs_global is None else n_envs_global
        T = self.num_steps
        N = T * E
        B = N if mini_batch_size is None else mini_batch_size
        ahead, self._perm_ahead = self._perm_ahead, None
        draw = lambda n: (ahead.take(n) if ahead is not None else None)
        if not recurrent:
            perm = draw(N)
            if perm is None:
                perm = _randperm_serial(N)
            for k in range(N // B):
                yield perm[k * B:(k + 1) * B].astype(np.int64)
        else:
            per = E // (N // B)
            perm = draw(E)
            if perm is None:
                perm = _randperm_serial(E)
            for s in range(0, E, per):
                envs = perm[s:s + per].astype(np.int64)
                yield (np.arange(T, dtype=np.int64)[:, None] * E + envs[None, :]).reshape(-1)

    def draw_permutation_ahead(self, n):
        """Start drawing the next update's first torch.randperm(n) now (see _PermutationAhead): call when nothing else will use torch's
        CPU generator until that update -- PPO.train does, right after an iteration's logging."""
        self._perm_ahead = _PermutationAhead(int(n))

    # ------------------------------------------------------------------ reference API: reads (compat, not the hot path)
    def _field(self, f):
        return torch.from_numpy(self._eng().read_field(f))

    obs_batch = property(lambda self: torch.from_numpy(np.stack([self._obs_as_ref(t) for t in range(self.num_steps + 1)])))
    hidden_states_batch = property(lambda self: torch.from_numpy(self._hidden))
    act_batch = property(lambda self: self._field(M.F_ACT))
    rew_batch = property(lambda self: self._field(M.F_REW))
    done_batch = property(lambda self: self._field(M.F_DONE))
    log_prob_act_batch = property(lambda self: self._field(M.F_LOGP))
    value_batch = property(lambda self: self._field(M.F_VALUE))
    return_batch = property(lambda self: self._field(M.F_RET))
    adv_batch = property(lambda self: self._field(M.F_ADV))

    def _obs_as_ref(self, t):
        o = self._eng().get_obs(t)
        if self.arch == "impala":
            return (o.transpose(0, 3, 1, 2) / 255.0).astype(np.float32)
        return o

    def collate_data(self, indices):
        """The reference's 8-tuple for explicit indices (storage.py:112-128), read back from the device.
        Quirk kept: hidden_state_batch is the whole (N,H) tensor, not indexed."""
        idx = np.asarray(indices, dtype=np.int64)
        T, E = self.num_steps, self.num_envs
        t, e = idx // E, idx % E
        frames = {tt: self._obs_as_ref(int(tt)) for tt in np.unique(t)}
        obs = torch.from_numpy(np.stack([frames[int(a)][int(b)] for a, b in zip(t, e)]))
        hid = torch.from_numpy(self._hidden[:-1].reshape(T * E, -1))
        g = lambda f, last=False: self._field(f)[:T].reshape(-1)[idx]
        return (obs, hid, g(M.F_ACT), g(M.F_DONE), g(M.F_LOGP), g(M.F_VALUE), g(M.F_RET), g(M.F_ADV))

    def fetch_train_generator(self, mini_batch_size=None, recurrent=False):
        T, E = self.num_steps, self.num_envs
        for idx in self.minibatch_index_stream(mini_batch_size, recurrent):
            sample = list(self.collate_data(idx))
            if recurrent:
                envs = idx[:len(idx) // T] % E
                sample[1] = torch.from_numpy(self._hidden[0:1][:, envs].reshape(-1, self.hidden_state_size))
            yield tuple(sample)

    def fetch_log_data(self):
        """storage.py:130-162 on the host mirrors (no device read-back).  Column-wise: a step's `info` is read as one array per key
        (common/env/vec_envs.py StepInfo, or gathered once from a list of dicts) and only the envs whose episode ended are looked at
        one by one -- the reference walks all T*E dicts three times."""
        T = self.num_steps
        infos = list(self.info_batch)

        def has(info, key):
            if hasattr(info, "has"):
                return info.has(key)
            return len(info) > 0 and key in info[0]

        def col(info, key):
            return info.column(key) if hasattr(info, "column") else np.array([i[key] for i in info])

        first = infos[0] if infos else ()
        rew_batch = np.array([col(i, 'env_reward') for i in infos[:T]]) if has(first, 'env_reward') else np.array(self._rew)
        done_batch = np.array([col(i, 'env_done') for i in infos[:T]]) if has(first, 'env_done') else np.array(self._done)
        if has(first, 'prev_level_seed'):
            for s, e in zip(*np.nonzero(done_batch > 0)):                  # row-major: step by step, env by env, as the reference
                info = infos[s][e]
                self.performance_track.setdefault(info["prev_level_seed"], deque(maxlen=10)).append(info["env_reward"])
        rewards = [r for dq in self.performance_track.values() for r in dq]
        true_average_reward = np.mean(rewards) if rewards else np.nan
        return rew_batch, done_batch, true_average_reward
