"""Host side of the rollout, the step right before the accelerated path.

The reference stacks five vector-env wrappers between Procgen and the agent (common/env/procgen_wrappers.py:565-587):
    VecExtractDictObs("rgb") -> VecNormalize(ob=False) -> [MirrorFrame] -> TransposeFrame -> ScaledFloatFrame -> [ActionWrapper]
which turn Procgen's uint8 NHWC frame buffer into a float64 NCHW array in [0,1] -- 8x the bytes, two full passes over them
per step -- that the agent then casts to fp32 and uploads (12.6 MB per step at E = 256).

`ProcgenFrameSource` is that whole chain as ONE object with the same outward behaviour (reset/step/step_async/step_wait,
observation_space (3,64,64), action_space.n, `info[i]['env_reward']`, `.combos`), except that the observation it hands out stays
what Procgen produced: uint8 (E,64,64,3).  The engine stores frames as uint8 NHWC and converts k -> bf16/fp32(k/255) inside the
first conv kernel, so TransposeFrame and ScaledFloatFrame have no work left on the host; `frames.transpose(0,3,1,2) / 255.0` of
what this returns IS the reference's observation, bit for bit (tests/test_env_pipeline.py).  Reward normalisation keeps the
reference's float64 running-variance arithmetic on the host; action remapping is one table lookup.

Procgen itself (procgen==0.10.7, gym3==0.3.3; third-party C++) is not part of this repository: anything with the
baselines-VecEnv surface (`num_envs`, `reset()`, `step_async(a)`, `step_wait()`, dict observations with an 'rgb' entry) works."""
import re

import numpy as np

# procgen 0.10.7, procgen/env.py get_combos(): the 15 key combinations an action index stands for (third-party, published;
# used only when the wrapped env does not carry `.combos` itself)
PROCGEN_COMBOS = [("LEFT", "DOWN"), ("LEFT",), ("LEFT", "UP"), ("DOWN",), (), ("UP",), ("RIGHT", "DOWN"), ("RIGHT",),
                  ("RIGHT", "UP"), ("D",), ("A",), ("W",), ("S",), ("Q",), ("E",)]
_KEY_ALIASES = {"D": "RIGHT", "A": "LEFT", "W": "UP", "S": "DOWN", "Q": "LEFT_UP", "E": "RIGHT_UP"}


def action_names(combos):
    """helper_local.py:198-204: combo -> name, the six single-letter keys renamed to the direction they duplicate."""
    names = ["_".join(c) if len(c) <= 2 else "" for c in combos]
    return np.array([_KEY_ALIASES.get(n, n) for n in names])


def first_index_of(wanted, names):
    """helper_local.py:65-70 (`match`): for every entry of `wanted` present in `names`, the index of its first occurrence."""
    names = list(names)
    return np.array([names.index(w) for w in wanted if w in names], dtype=np.int64)


def reduced_action_table(combos):
    """ActionWrapper (procgen_wrappers.py:422-446): agent action a in [0, n_unique) -> Procgen action index.
    Returns (table, unique_names, reduced_combos)."""
    names = action_names(combos)
    uniq = np.unique(names)
    table = first_index_of(uniq, names)
    reduced = [(x[0],) if len(x) == 1 else (x[0], x[1]) for x in (re.split("_", u) for u in uniq)]
    return table, uniq, reduced


def mirror_action_table(combos):
    """MirrorFrame (procgen_wrappers.py:362-373): Procgen action index -> the index of the left/right-swapped action."""
    names = action_names(combos)
    swapped = [re.sub("LEFT", "RIGHT", n) if re.search("LEFT", n) else re.sub("RIGHT", "LEFT", n) for n in names]
    return first_index_of(swapped, names)


class RunningMoments:
    """RunningMeanStd for a scalar stream (procgen_wrappers.py:282-313): Chan's parallel update in float64, count starts at 1e-4."""

    def __init__(self, epsilon=1e-4):
        self.mean, self.var, self.count = np.float64(0.0), np.float64(1.0), epsilon

    def update(self, x):
        b_mean, b_var, b_count = np.mean(x, axis=0), np.var(x, axis=0), x.shape[0]
        delta = b_mean - self.mean
        tot = self.count + b_count
        m2 = self.var * self.count + b_var * b_count + np.square(delta) * self.count * b_count / tot
        self.mean, self.var, self.count = self.mean + delta * b_count / tot, m2 / tot, tot


class RewardNormalizer:
    """VecNormalize(ob=False) (procgen_wrappers.py:316-355): rewards divided by the running std of the discounted return
    (gamma 0.99), clipped to +-10; the discounted return of an env restarts after its episode ends."""

    def __init__(self, n_envs, cliprew=10.0, gamma=0.99, epsilon=1e-8, ret_rms=None):
        self.ret_rms = ret_rms if ret_rms is not None else RunningMoments()      # env groups of one run share ONE running variance
        self.ret = np.zeros(n_envs)
        self.cliprew, self.gamma, self.epsilon = cliprew, gamma, epsilon

    def reset(self):
        self.ret = np.zeros(len(self.ret))

    def __call__(self, rews, news):
        self.ret = self.ret * self.gamma + rews
        self.ret_rms.update(self.ret)
        out = np.clip(rews / np.sqrt(self.ret_rms.var + self.epsilon), -self.cliprew, self.cliprew)
        self.ret[news] = 0.0
        return out

    def state(self):
        """(new) what a checkpoint needs to resume with the same reward scale (the reference loses it)."""
        return {"mean": float(self.ret_rms.mean), "var": float(self.ret_rms.var), "count": float(self.ret_rms.count)}

    def load_state(self, s):
        self.ret_rms.mean, self.ret_rms.var, self.ret_rms.count = np.float64(s["mean"]), np.float64(s["var"]), s["count"]


class _Space:
    def __init__(self, shape=None, n=None):
        self.shape, self.n = shape, n


class ProcgenFrameSource:
    def __init__(self, venv, normalize_rew=True, mirror_env=False, reduce_duplicate_actions=True, key="rgb", ret_rms=None):
        self.venv = venv
        self.num_envs = self.n_envs = venv.num_envs
        self.key = key
        combos = _find_combos(venv)
        self._to_procgen = None
        self.combos = combos
        n_actions = len(combos)
        if reduce_duplicate_actions:
            self._to_procgen, self.unique_actions, self.combos = reduced_action_table(combos)
            n_actions = len(self._to_procgen)
        self._mirror = mirror_action_table(combos) if mirror_env else None
        self._flip = (np.arange(self.num_envs) % 2 == 1)                    # odd envs are mirrored (procgen_wrappers.py:374)
        self._flipped = None
        self._rew = RewardNormalizer(self.num_envs, ret_rms=ret_rms) if normalize_rew else None
        self._act_dtype = getattr(getattr(venv, "action_space", None), "dtype", None) or np.int32
        self.observation_space = _Space(shape=(3, 64, 64))                  # what TransposeFrame advertises; the DATA stays NHWC uint8
        self.action_space = _Space(n=n_actions)

    # ---- observations: Procgen's own uint8 buffer, mirrored copies for the odd envs if asked for
    def _frames(self, obs):
        frames = obs[self.key] if isinstance(obs, dict) else obs
        if frames.dtype != np.uint8 or frames.ndim != 4 or frames.shape[-1] != 3:
            raise ValueError(f"expected uint8 (E,H,W,3) frames from the env, got {frames.dtype} {frames.shape}")
        if self._mirror is None:
            return frames
        if self._flipped is None:
            self._flipped = np.empty_like(frames)
        np.copyto(self._flipped, frames)
        self._flipped[self._flip] = frames[self._flip, :, ::-1]
        return self._flipped

    def reset(self):
        if self._rew is not None:
            self._rew.reset()
        return self._frames(self.venv.reset())

    # ---- actions: reduced index -> Procgen index -> mirrored index for the flipped envs
    def step_async(self, actions):
        a = np.asarray(actions)
        if self._to_procgen is not None:
            a = self._to_procgen[a]
        a = a.astype(self._act_dtype, copy=True)
        if self._mirror is not None:
            a[self._flip] = self._mirror[a[self._flip]]
        self.venv.step_async(a)

    def step_wait(self):
        obs, rews, news, infos = self.venv.step_wait()
        if self._rew is not None:
            # VecNormalize.step_wait writes infos[i]['env_reward'] = rews[i] into every env's dict (procgen_wrappers.py:336-337): here one
            # column beside the env's own dicts (StepInfo: info[i]['env_reward'] reads the same value)
            from common.env.vec_envs import StepInfo
            infos = StepInfo(len(infos), {"env_reward": rews}, rows=infos)
            rews = self._rew(rews, news)
        return self._frames(obs), rews, news, infos

    def step(self, actions):
        self.step_async(actions)
        return self.step_wait()

    def close(self):
        return self.venv.close()

    def reward_state(self):
        return self._rew.state() if self._rew is not None else None


def _find_combos(env):
    """helper_local.py:163-168, extended over `.venv` chains; Procgen's own table when the env does not say."""
    seen = 0
    while env is not None and seen < 16:
        if "combos" in getattr(env, "__dict__", {}) or hasattr(type(env), "combos"):
            return list(env.combos)
        env = getattr(env, "__dict__", {}).get("env") or getattr(env, "__dict__", {}).get("venv")
        seen += 1
    return list(PROCGEN_COMBOS)


def create_procgen_env(env_name="coinrun", n_envs=256, is_valid=False, val_env_name=None, start_level=0, num_levels=500,
                       distribution_mode="hard", num_threads=8, paint_vel_info=True, normalize_rew=True, mirror_env=False,
                       reduce_duplicate_actions=True, start_level_val=None, ret_rms=None):
    """procgen_wrappers.py:524-587 for the real-Procgen branch: the validation env draws from all levels (`num_levels=0`)
    starting at a random level in [500, 9999]."""
    try:
        from procgen import ProcgenEnv
    except ImportError as e:
        raise NotImplementedError("the Procgen engine (procgen==0.10.7, gym3==0.3.3) is not installed in this image; "
                                  "use --env_name synthetic or cartpole") from e
    import random
    if start_level_val is None:
        start_level_val = random.randint(500, 9999)
    if start_level == start_level_val:
        raise ValueError("Seeds for training and validation envs are equal.")
    venv = ProcgenEnv(num_envs=n_envs, env_name=(val_env_name or env_name) if is_valid else env_name,
                      num_levels=0 if is_valid else num_levels, start_level=start_level_val if is_valid else start_level,
                      paint_vel_info=paint_vel_info, distribution_mode=distribution_mode, num_threads=num_threads)
    return ProcgenFrameSource(venv, normalize_rew=normalize_rew, mirror_env=mirror_env,
                              reduce_duplicate_actions=reduce_duplicate_actions, ret_rms=ret_rms)
