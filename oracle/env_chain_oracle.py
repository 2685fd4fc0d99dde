"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's Procgen wrapper chain, stage by stage and
with the reference's data types (float64 NCHW observations), to check `common/env/procgen_pipeline.py::ProcgenFrameSource`.

PARITY UNPINNED: the reference's `common/env/procgen_wrappers.py` cannot be imported here (it needs gym3, procgen and, through
helper_local, wandb/moviepy -- none installed, none stubbed) and none of the reference's tests holds an env fixture, so this
restatement is checked only against first principles in tests/test_env_pipeline.py (the running variance against numpy's variance
of the concatenated stream; the action tables against the name lists written out by hand).

Stages, in the order of procgen_wrappers.py:565-587:
  extract   VecExtractDictObs("rgb")            :272-279   obs = obs_dict["rgb"]                      uint8 (E,64,64,3)
  normalise VecNormalize(ob=False)              :316-355   info.env_reward = r; R = 0.99 R + r; rms.update(R); r' = clip(r/sqrt(var+1e-8), +-10); R[done]=0
  mirror    MirrorFrame        (mirror_env)     :358-388   odd envs: frame flipped along W; their actions swapped LEFT<->RIGHT
  transpose TransposeFrame                      :391-404   (E,H,W,C) -> (E,C,H,W)
  scale     ScaledFloatFrame                    :407-419   obs / 255.0  (float64)
  actions   ActionWrapper      (reduce dups)    :422-446   a -> first Procgen index whose (aliased) name is the a-th unique name
"""
import numpy as np

COMBOS_0_10_7 = [("LEFT", "DOWN"), ("LEFT",), ("LEFT", "UP"), ("DOWN",), (), ("UP",), ("RIGHT", "DOWN"), ("RIGHT",),
                 ("RIGHT", "UP"), ("D",), ("A",), ("W",), ("S",), ("Q",), ("E",)]


def names_of(combos):
    """helper_local.py:198-204"""
    out = []
    for c in combos:
        name = c[0] + "_" + c[1] if len(c) == 2 else (c[0] if len(c) == 1 else "")
        out.append({"D": "RIGHT", "A": "LEFT", "W": "UP", "S": "DOWN", "Q": "LEFT_UP", "E": "RIGHT_UP"}.get(name, name))
    return out


def chan_merge(mean, var, count, b_mean, b_var, b_count):
    """procgen_wrappers.py:300-313"""
    delta = b_mean - mean
    tot = count + b_count
    m2 = var * count + b_var * b_count + delta ** 2 * count * b_count / tot
    return mean + delta * b_count / tot, m2 / tot, tot


class ReferenceChain:
    def __init__(self, venv, normalize_rew=True, mirror_env=False, reduce_duplicate_actions=True, combos=None):
        self.venv, self.E = venv, venv.num_envs
        self.normalize_rew, self.mirror_env, self.reduce = normalize_rew, mirror_env, reduce_duplicate_actions
        names = names_of(combos or COMBOS_0_10_7)
        self.names = names
        self.unique = sorted(set(names))
        self.reduce_table = [names.index(u) for u in self.unique]
        swapped = [n.replace("LEFT", "RIGHT") if "LEFT" in n else n.replace("RIGHT", "LEFT") for n in names]
        self.mirror_table = [names.index(s) for s in swapped]
        self.n_actions = len(self.unique) if reduce_duplicate_actions else len(names)
        self.mean, self.var, self.count = np.float64(0), np.float64(1), 1e-4
        self.ret = np.zeros(self.E)

    def _observe(self, obs_dict):
        x = np.array(obs_dict["rgb"])                                         # extract (copy: mirror writes in place)
        if self.mirror_env:
            for e in range(1, self.E, 2):
                x[e] = x[e][:, ::-1, :]
        return x.transpose(0, 3, 1, 2) / 255.0                                # transpose, scale -> float64

    def reset(self):
        self.ret = np.zeros(self.E)
        return self._observe(self.venv.reset())

    def step(self, actions):
        a = np.array(actions).copy()
        if self.reduce:
            a = np.array([self.reduce_table[k] for k in a], dtype=np.int32)       # table dtype = venv.action_space.dtype (:434)
        if self.mirror_env:
            for e in range(1, self.E, 2):
                a[e] = self.mirror_table[a[e]]
        self.venv.step_async(a)
        obs, rews, news, infos = self.venv.step_wait()
        if self.normalize_rew:
            for e in range(self.E):
                infos[e]["env_reward"] = rews[e]
            self.ret = self.ret * 0.99 + rews
            self.mean, self.var, self.count = chan_merge(self.mean, self.var, self.count, np.mean(self.ret), np.var(self.ret), self.E)
            rews = np.clip(rews / np.sqrt(self.var + 1e-8), -10.0, 10.0)
            self.ret[news] = 0.0
        return self._observe(obs), rews, news, infos
