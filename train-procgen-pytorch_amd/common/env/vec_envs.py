"""Host-side vectorised environments that speak the baselines VecEnv protocol the agent drives
(reset() -> obs ; step(act) -> (obs, rew, done, info) ; close()), reference:
common/env/procgen_wrappers.py:45-124.  Rollout collection stays on the host (north_star).

* CartPoleVec   -- the reference's 9-observation pre-vectorised cart-pole (discrete_env/cartpole_pre_vec.py) restated in numpy,
                   the plumbing config C1 (MLPModel(9, ...)).
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
    """The reference's pre-vectorised cart-pole (discrete_env/cartpole_pre_vec.py:20-205 on discrete_env/pre_vec_env.py:21-125),
    restated in numpy: BASELINE config 1's env.  Observation = the 9-column state (cartpole_pre_vec.py:136-149, 197-208)

        [x, x_dot, theta, theta_dot, gravity, pole_length, cart_mass, pole_mass, force_mag]

    -- the four dynamic variables plus the five physics parameters an episode was drawn with, which is why the reference builds
    MLPModel(9, ...) for `--env_name cartpole`.  Every episode start draws all nine uniformly from [low, high] (start_space,
    :163-171: the dynamic variables from +-0.05, the parameters from their min/max range); Florian's equations with the per-env
    parameters, Euler steps of tau = 0.02 (:214-245); termination beyond +-h_range or +-degrees (:254-262), truncation at max_steps
    (pre_vec_env.py:86-87); reward 1 every step; info[i] = {'env_reward': 1.0} (:112-113).  Kept from the reference: when ANY env ends,
    a whole (n_envs, 9) block is drawn and only the ended rows take their values (pre_vec_env.py:111-118), and `done` is the
    `terminated` array AFTER the reset.  The generator is numpy's default_rng(seed) -- what gymnasium's seeding.np_random(seed)
    constructs (PCG64 over SeedSequence(seed)).  Parity with the reference's class is UNPINNED: it cannot be imported here without
    stand-ins for gymnasium code it executes (spaces, Env, seeding) and the reference holds no trajectory fixture;
    tests/test_env_pipeline.py checks the dynamics against a scalar restatement of the same equations."""
    tau = 0.02

    def __init__(self, n_envs, degrees=12, h_range=2.4, min_gravity=9.8, max_gravity=10.4, min_pole_length=0.5, max_pole_length=1.0,
                 min_cart_mass=1.0, max_cart_mass=1.5, min_pole_mass=0.1, max_pole_mass=0.2, min_force_mag=10., max_force_mag=10.,
                 max_steps=500, seed=0, **_ignored):
        if n_envs < 2:
            raise Exception("n_envs must be greater than or equal to 2")
        self.n_envs = self.num_envs = n_envs
        self.max_steps = max_steps
        self.theta_threshold, self.x_threshold = degrees * 2 * np.pi / 360, h_range
        self.low = np.array([-0.05] * 4 + [min_gravity, min_pole_length, min_cart_mass, min_pole_mass, min_force_mag])
        self.high = np.array([0.05] * 4 + [max_gravity, max_pole_length, max_cart_mass, max_pole_mass, max_force_mag])
        self.rng = np.random.default_rng(seed)
        self.observation_space = _Space(shape=(9,))
        self.action_space = _Space(n=2)
        self.reward = np.ones(n_envs)
        self.info = StepInfo(n_envs, {"env_reward": self.reward})
        self.state = np.zeros((n_envs, 9))
        self.terminated = np.full(n_envs, True)
        self.n_steps = np.zeros(n_envs)

    def _set(self):
        fresh = self.rng.uniform(low=self.low, high=self.high, size=(self.n_envs, 9))
        self.state[self.terminated] = fresh[self.terminated]
        self.n_steps[self.terminated] = 0
        return self.state

    def reset(self):
        self.terminated = np.full(self.n_envs, True)
        self.n_steps = np.zeros(self.n_envs)
        return self._set()

    def step(self, act):
        act = np.asarray(act)
        assert act.size == self.n_envs and np.all(act < 2)
        x, xd, th, thd, g, length, m_cart, m_pole, f_mag = self.state.T
        force = np.where(act.reshape(-1) == 0, -1.0, 1.0) * f_mag
        ct, st = np.cos(th), np.sin(th)
        pml, total = m_pole * length, m_pole + m_cart
        temp = (force + pml * thd ** 2 * st) / total
        thacc = (g * st - ct * temp) / (length * (4.0 / 3.0 - m_pole * ct ** 2 / total))
        xacc = temp - pml * thacc * ct / total
        self.state = np.vstack((x + self.tau * xd, xd + self.tau * xacc, th + self.tau * thd, thd + self.tau * thacc,
                                g, length, m_cart, m_pole, f_mag)).T
        nx, nth = self.state[:, 0], self.state[:, 2]
        self.terminated = (nx < -self.x_threshold) | (nx > self.x_threshold) | (nth < -self.theta_threshold) | (nth > self.theta_threshold)
        self.n_steps += 1
        self.terminated[self.n_steps >= self.max_steps] = True
        if np.any(self.terminated):
            self._set()
        return self.state, self.reward, self.terminated, self.info

    def close(self):
        pass


# create_cartpole (discrete_env/cartpole_pre_vec.py:397-412): [training value, validation value] of every constructor argument
CARTPOLE_PARAM_RANGE = {"degrees": [12], "h_range": [2.4], "min_gravity": [9.8, 10.4], "max_gravity": [10.4, 24.8],
                        "min_pole_length": [0.5, 1.0], "max_pole_length": [1.0, 2.0], "min_cart_mass": [1.0, 2.], "max_cart_mass": [1.5, 3.],
                        "min_pole_mass": [0.1, .2], "max_pole_mass": [0.2, .4], "min_force_mag": [10.], "max_force_mag": [10.]}


def create_cartpole(hyperparameters, is_valid=False, seed=0, n_envs=None):
    """create_cartpole / create_pre_vec / assign_env_vars (cartpole_pre_vec.py:397-412, pre_vec_env.py:207-223, helper_pre_vec.py:52-66):
    the validation env draws its physics from the SECOND value of every range (heavier, longer, stronger gravity); a hyper-parameter
    `<name>` (training) or `<name>_v` (validation) overrides a range entry (config.yml `cartpole`: degrees_v 9, h_range_v 1.8);
    max_steps may come from the hyper-parameters too."""
    suffix, i = ("_v", -1) if is_valid else ("", 0)
    kw = {k: hyperparameters.get(k + suffix, v[i]) for k, v in CARTPOLE_PARAM_RANGE.items()}
    if "max_steps" in hyperparameters:
        kw["max_steps"] = hyperparameters["max_steps"]
    return CartPoleVec(n_envs if n_envs is not None else hyperparameters.get("n_envs", 32), seed=seed, **kw)


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
