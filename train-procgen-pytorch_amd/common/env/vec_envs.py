"""Host-side vectorised environments that speak the baselines VecEnv protocol the agent drives
(reset() -> obs ; step(act) -> (obs, rew, done, info) ; close()), reference:
common/env/procgen_wrappers.py:45-124.  Rollout collection stays on the host (north_star).

* CartPoleVec   -- a from-scratch numpy cart-pole (classic Barto/Sutton dynamics, Euler, tau 0.02) with
                   auto-reset, the plumbing config C1 (MLP policy).
* SyntheticFrames -- random uint8 64x64x3 frames, N(0,1) rewards, Bernoulli(0.01) dones: the synthetic
                   learner-side workload of SURVEY 8(d) / bench.py.
* create_procgen_env -- the real engine if the `procgen` package is importable (it is not in the build image).
"""
import numpy as np


class _Space:
    def __init__(self, shape=None, n=None):
        self.shape, self.n = shape, n


class StepInfo:
    """One vector-env step's `info`, kept column-wise.

    The reference's protocol hands the agent a list of E dicts per step and walks them entry by entry afterwards
    (`Storage.fetch_log_data`, common/storage.py:130-162: T*E dict look-ups per iteration; `VecNormalize.step_wait`,
    procgen_wrappers.py:336-337: E dict writes per step).  A StepInfo answers to the same uses -- len(), info[i] -> dict,
    iteration, `'key' in info[0]` -- but holds what the wrappers add (`env_reward`, ...) as ONE array per key and leaves the
    env's own per-env dicts (`rows`, e.g. Procgen's prev_level_seed / level_complete) untouched; `column(key)` is what the logging
    path reads.  `StepInfo.join(parts)` shows the infos of several env groups as one step's info without copying."""
    __slots__ = ("n", "columns", "rows", "parts")

    def __init__(self, n, columns=None, rows=None, parts=None):
        self.n, self.columns, self.rows, self.parts = n, columns or {}, rows, parts

    @staticmethod
    def join(parts):
        parts = [p if isinstance(p, StepInfo) else StepInfo(len(p), rows=p) for p in parts]
        return parts[0] if len(parts) == 1 else StepInfo(sum(p.n for p in parts), parts=parts)

    def __len__(self):
        return self.n

    def has(self, key):
        if self.parts is not None:
            return bool(self.parts) and self.parts[0].has(key)
        return key in self.columns or bool(self.rows) and key in self.rows[0]

    def column(self, key):
        if self.parts is not None:
            return np.concatenate([p.column(key) for p in self.parts])
        if key in self.columns:
            return np.asarray(self.columns[key])
        return np.array([r[key] for r in self.rows])

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[j] for j in range(*i.indices(self.n))]
        if i < 0:
            i += self.n
        if self.parts is not None:
            for p in self.parts:
                if i < p.n:
                    return p[i]
                i -= p.n
            raise IndexError(i)
        d = dict(self.rows[i]) if self.rows else {}
        for k, v in self.columns.items():
            d[k] = v[i]
        return d

    def __iter__(self):
        return (self[i] for i in range(self.n))


class CartPoleVec:
    gravity, masscart, masspole, length, force_mag, tau = 9.8, 1.0, 0.1, 0.5, 10.0, 0.02
    x_threshold, theta_threshold = 2.4, 12 * 2 * np.pi / 360

    def __init__(self, n_envs, max_steps=500, seed=0):
        self.n_envs, self.max_steps = n_envs, max_steps
        self.rng = np.random.default_rng(seed)
        self.observation_space = _Space(shape=(4,))
        self.action_space = _Space(n=2)
        self.state = np.zeros((n_envs, 4))
        self.steps = np.zeros(n_envs, dtype=np.int64)

    def _fresh(self, k):
        return self.rng.uniform(-0.05, 0.05, size=(k, 4))

    def reset(self):
        self.state = self._fresh(self.n_envs)
        self.steps[:] = 0
        return self.state.astype(np.float32)

    def step(self, act):
        x, xd, th, thd = self.state.T
        force = np.where(np.asarray(act) == 1, self.force_mag, -self.force_mag)
        total_m, pml = self.masscart + self.masspole, self.masspole * self.length
        ct, st = np.cos(th), np.sin(th)
        temp = (force + pml * thd ** 2 * st) / total_m
        thacc = (self.gravity * st - ct * temp) / (self.length * (4.0 / 3.0 - self.masspole * ct ** 2 / total_m))
        xacc = temp - pml * thacc * ct / total_m
        self.state = np.stack([x + self.tau * xd, xd + self.tau * xacc, th + self.tau * thd, thd + self.tau * thacc], axis=1)
        self.steps += 1
        fell = (np.abs(self.state[:, 0]) > self.x_threshold) | (np.abs(self.state[:, 2]) > self.theta_threshold)
        done = fell | (self.steps >= self.max_steps)
        rew = np.ones(self.n_envs, dtype=np.float32)
        info = [{} for _ in range(self.n_envs)]
        if done.any():
            k = int(done.sum())
            self.state[done] = self._fresh(k)
            self.steps[done] = 0
        return self.state.astype(np.float32), rew, done, info

    def close(self):
        pass


class SyntheticFrames:
    def __init__(self, n_envs, n_actions=15, seed=0, pool=8):
        self.n_envs = n_envs
        self.rng = np.random.default_rng(seed)
        self.observation_space = _Space(shape=(3, 64, 64))
        self.action_space = _Space(n=n_actions)
        self._pool = [self.rng.integers(0, 256, size=(n_envs, 64, 64, 3), dtype=np.uint8) for _ in range(pool)]
        self._k = 0
        self._level = np.arange(n_envs) % 500

    def _obs(self):
        self._k = (self._k + 1) % len(self._pool)
        return self._pool[self._k]                       # uint8 NHWC: what Procgen's 'rgb' delivers

    def reset(self):
        return self._obs()

    def step(self, act):
        rew = self.rng.standard_normal(self.n_envs).astype(np.float32)
        done = self.rng.random(self.n_envs) < 0.01
        return self._obs(), rew, done, StepInfo(self.n_envs, {"env_reward": rew, "prev_level_seed": self._level})

    def close(self):
        pass


class SyntheticTape(SyntheticFrames):
    """SyntheticFrames with every random draw made up front (a tape of `length` steps, replayed in a loop): step() is a few
    microseconds of host work whatever n_envs is.  bench.py's stand-in for the Procgen engine -- the metric excludes env.step's own
    cost, and a group's env.step sits INSIDE the rollout's dependency chain, so the stand-in must not be what the chain waits for.
    Frames are handed out from a small pool of ordinary (pageable) numpy arrays, as an engine would from its own buffers."""

    def __init__(self, n_envs, n_actions=15, seed=0, pool=4, length=256):
        super().__init__(n_envs, n_actions, seed, pool)
        self._rew = self.rng.standard_normal((length, n_envs)).astype(np.float32)
        self._done = self.rng.random((length, n_envs)) < 0.01
        self._info = [StepInfo(n_envs, {"env_reward": self._rew[i], "prev_level_seed": self._level}) for i in range(length)]
        self._i = -1

    def step(self, act):
        self._i = i = (self._i + 1) % len(self._info)
        return self._obs(), self._rew[i], self._done[i], self._info[i]


class EnvGroups:
    """G independent vector envs shown as ONE VecEnv of sum(n_envs) environments (reset / step concatenate in group order), plus
    `.env_groups` for the agent's pipelined collector (agents/ppo.py `_collect_pipelined`): an env's next observation depends only
    on its own action, so the agent steps group g on the host while the device works on the other groups' frames."""

    def __init__(self, envs):
        self.env_groups = list(envs)
        e0 = self.env_groups[0]
        self.n_envs = self.num_envs = sum(getattr(e, "n_envs", getattr(e, "num_envs", 0)) for e in self.env_groups)
        self.observation_space, self.action_space = e0.observation_space, e0.action_space
        for k in ("combos", "unique_actions"):
            if hasattr(e0, k):
                setattr(self, k, getattr(e0, k))

    def reset(self):
        return np.concatenate([e.reset() for e in self.env_groups])

    def step(self, act):
        act = np.asarray(act)
        outs, o = [], 0
        for e in self.env_groups:
            n = getattr(e, "n_envs", getattr(e, "num_envs", 0))
            outs.append(e.step(act[o:o + n])); o += n
        return (np.concatenate([x[0] for x in outs]), np.concatenate([x[1] for x in outs]), np.concatenate([x[2] for x in outs]),
                StepInfo.join([x[3] for x in outs]))

    def reward_state(self):
        rs = getattr(self.env_groups[0], "reward_state", None)
        return rs() if callable(rs) else None

    def close(self):
        for e in self.env_groups:
            e.close()


def create_procgen_env(**kwargs):
    """The Procgen engine behind the one-object wrapper chain of common/env/procgen_pipeline.py (uint8 NHWC frames out)."""
    from common.env.procgen_pipeline import create_procgen_env as make
    return make(**kwargs)
