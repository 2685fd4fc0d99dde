"""Host env pipeline: ProcgenFrameSource against the stage-by-stage restatement of the reference's wrapper
chain (oracle/env_chain_oracle.py -- parity unpinned at this boundary: no Procgen, no reference fixture) and against first
principles.  A deterministic fake stands in for ProcgenEnv (dict observations, async step, int32 actions)."""
import os
import sys
import time

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "train-procgen-pytorch_amd")
sys.path[:0] = [ROOT, PKG]

from common.env import procgen_pipeline as P          # noqa: E402
from oracle import env_chain_oracle as O                # noqa: E402


class FakeProcgen:
    """What gym3.ToBaselinesVecEnv(ProcgenGym3Env) looks like from outside."""

    class _ActSpace:
        dtype = np.int32
        n = 15

    def __init__(self, n_envs, seed=0, done_rate=0.05):
        self.num_envs, self.rng, self.done_rate = n_envs, np.random.default_rng(seed), done_rate
        self.action_space = self._ActSpace()
        self.received = []
        self._a = None

    def _obs(self):
        f = self.rng.integers(0, 256, size=(self.num_envs, 64, 64, 3), dtype=np.uint8)
        f.setflags(write=False)                                       # the engine's buffer is not ours to write
        return {"rgb": f}

    def reset(self):
        return self._obs()

    def step_async(self, a):
        assert a.dtype.kind == 'i' and a.shape == (self.num_envs,) and a.min() >= 0 and a.max() < 15
        self._a = a
        self.received.append(a.copy())

    def step_wait(self):
        rew = (self.rng.standard_normal(self.num_envs) * (1 + self._a % 3)).astype(np.float32)
        done = self.rng.random(self.num_envs) < self.done_rate
        return self._obs(), rew, done, [{"prev_level_seed": int(e)} for e in range(self.num_envs)]

    def close(self):
        pass


def test_action_tables_match_the_names_written_out():
    names = list(P.action_names(P.PROCGEN_COMBOS))
    assert names == ["LEFT_DOWN", "LEFT", "LEFT_UP", "DOWN", "", "UP", "RIGHT_DOWN", "RIGHT", "RIGHT_UP",
                     "RIGHT", "LEFT", "UP", "DOWN", "LEFT_UP", "RIGHT_UP"]
    table, uniq, combos = P.reduced_action_table(P.PROCGEN_COMBOS)
    assert list(uniq) == ["", "DOWN", "LEFT", "LEFT_DOWN", "LEFT_UP", "RIGHT", "RIGHT_DOWN", "RIGHT_UP", "UP"]
    assert table.tolist() == [4, 3, 1, 0, 2, 7, 6, 8, 5]
    assert combos[0] == ("",) and combos[3] == ("LEFT", "DOWN") and len(combos) == 9
    mirror = P.mirror_action_table(P.PROCGEN_COMBOS)
    assert mirror.tolist() == [6, 7, 8, 3, 4, 5, 0, 1, 2, 1, 7, 5, 3, 8, 2]
    chain = O.ReferenceChain(FakeProcgen(2))
    assert chain.reduce_table == table.tolist() and chain.mirror_table == mirror.tolist()


@pytest.mark.parametrize("normalize_rew,mirror_env,reduce", [(True, False, True), (True, True, True), (False, False, False),
                                                             (True, True, False)])
def test_frame_source_equals_the_wrapper_chain(normalize_rew, mirror_env, reduce):
    E, steps = 6, 40
    src = P.ProcgenFrameSource(FakeProcgen(E, seed=3), normalize_rew, mirror_env, reduce)
    ref_env = FakeProcgen(E, seed=3)
    ref = O.ReferenceChain(ref_env, normalize_rew, mirror_env, reduce)
    assert src.action_space.n == ref.n_actions == (9 if reduce else 15) and src.observation_space.shape == (3, 64, 64)
    f, o = src.reset(), ref.reset()
    assert f.dtype == np.uint8 and f.shape == (E, 64, 64, 3) and o.dtype == np.float64 and o.shape == (E, 3, 64, 64)
    assert np.array_equal(f.transpose(0, 3, 1, 2) / 255.0, o)
    rng = np.random.default_rng(0)
    for _ in range(steps):
        act = rng.integers(0, src.action_space.n, E)
        keep = act.copy()
        f, r, d, info = src.step(act)
        o, r2, d2, info2 = ref.step(act)
        assert np.array_equal(act, keep)                                       # the caller's actions are not remapped in place
        assert np.array_equal(f.transpose(0, 3, 1, 2) / 255.0, o)               # bit-exact: k/255 in float64 on both sides
        assert r.dtype == r2.dtype and np.array_equal(r, r2) and np.array_equal(d, d2)
        assert ("env_reward" in info[0]) == normalize_rew
        if normalize_rew:
            assert [i["env_reward"] for i in info] == [i["env_reward"] for i in info2] and np.abs(r).max() <= 10.0
    assert all(np.array_equal(a, b) for a, b in zip(src.venv.received, ref_env.received))
    if normalize_rew:
        s = src.reward_state()
        assert s["var"] == float(ref.var) and s["count"] == pytest.approx(1e-4 + E * steps)


def test_running_variance_is_the_variance_of_the_whole_stream():
    rng = np.random.default_rng(1)
    m = P.RunningMoments()
    chunks = [rng.standard_normal(17) * 3 + 1 for _ in range(50)]
    for c in chunks:
        m.update(c)
    allx = np.concatenate(chunks)
    # the 1e-4 pseudo-count of (mean 0, var 1) the stream starts from moves the result by ~1e-7 relative
    assert m.count == pytest.approx(len(allx) + 1e-4)
    assert float(m.mean) == pytest.approx(allx.mean(), rel=1e-6) and float(m.var) == pytest.approx(allx.var(), rel=1e-5)


def test_reward_normaliser_restarts_the_return_of_finished_envs():
    n = P.RewardNormalizer(3)
    n(np.array([1.0, 2.0, 3.0], dtype=np.float32), np.array([False, True, False]))
    assert n.ret.tolist() == [1.0, 0.0, 3.0]
    out = n(np.array([100.0, 100.0, -100.0], dtype=np.float32), np.array([False, False, False]))
    assert n.ret.tolist() == [100.99, 100.0, -97.03] and out.max() <= 10 and out.min() >= -10
    st = n.state()
    m = P.RewardNormalizer(3)
    m.load_state(st)
    assert m.state() == st


def test_frames_reach_the_engine_format_without_a_copy_or_a_float():
    from common.model import as_device_obs
    src = P.ProcgenFrameSource(FakeProcgen(4), True, False, True)
    f = src.reset()
    assert as_device_obs(f, "impala") is f or np.shares_memory(as_device_obs(f, "impala"), f)
    ref = O.ReferenceChain(FakeProcgen(4))
    assert np.array_equal(as_device_obs(ref.reset(), "impala"), f)              # the reference's float64 NCHW converts back to the same bytes


def test_host_cost_per_step_against_the_wrapper_chain():
    """Not a parity check: records what the chain costs the host per step at E = 256 (the numbers quoted in DESIGN.md)."""
    E = 256
    src = P.ProcgenFrameSource(FakeProcgen(E, seed=1), True, False, True)
    ref = O.ReferenceChain(FakeProcgen(E, seed=1), True, False, True)
    src.reset(); ref.reset()
    act = np.zeros(E, dtype=np.int64)
    frames = src.venv._obs()
    src.venv._obs = lambda: frames                 # take the fake's random-number generation out of the timing
    ref.venv._obs = lambda: frames
    t0 = time.perf_counter()
    for _ in range(5):
        src.step(act)
    t1 = time.perf_counter()
    for _ in range(5):
        o, *_ = ref.step(act)
        o.astype(np.float32)                       # what agents/ppo.py:76 (torch.FloatTensor(obs)) adds on the reference side
    t2 = time.perf_counter()
    print(f"\nhost pipeline per step at E=256: frame source {(t1 - t0) / 5 * 1e3:.2f} ms, wrapper chain + fp32 cast {(t2 - t1) / 5 * 1e3:.2f} ms")
    assert (t1 - t0) < (t2 - t1)
