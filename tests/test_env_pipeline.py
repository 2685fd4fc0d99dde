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


# ------------------------------------------------------------------------------------------------ BASELINE config 1's env
def test_cartpole_nine_observation_layout_and_dynamics():
    """common/env/vec_envs.py CartPoleVec = the reference's pre-vectorised cart-pole (discrete_env/cartpole_pre_vec.py:136-149,197-262;
    pre_vec_env.py:78-118), PARITY UNPINNED (the reference class needs gymnasium code executed; no trajectory fixture exists).  Checked:
    the 9-column observation [x, x_dot, theta, theta_dot, gravity, pole_length, cart_mass, pole_mass, force_mag]; an env's five physics
    columns are constant within an episode and inside the constructor's ranges; every transition equals a scalar restatement of
    Florian's equations with that env's parameters; termination beyond +-2.4 / +-12 degrees and truncation at max_steps re-draw ALL nine
    columns of exactly the ended envs; reward 1, info[i]['env_reward'] == 1; train / validation ranges of create_cartpole and the
    `_v` overrides of the `cartpole` hyper-parameter set."""
    import math
    from common.env.vec_envs import CARTPOLE_PARAM_RANGE, CartPoleVec, create_cartpole
    E = 16
    env = CartPoleVec(E, seed=3, max_steps=40)
    obs = env.reset().copy()
    assert obs.shape == (E, 9) and env.observation_space.shape == (9,) and env.action_space.n == 2
    lo = np.array([-.05] * 4 + [9.8, 0.5, 1.0, 0.1, 10.0]); hi = np.array([.05] * 4 + [10.4, 1.0, 1.5, 0.2, 10.0])
    assert (obs >= lo).all() and (obs <= hi).all() and len(np.unique(obs[:, 4])) == E
    rng = np.random.default_rng(0)
    steps_alive = np.zeros(E)
    ended_by_fall = ended_by_time = 0
    for t in range(300):
        act = rng.integers(0, 2, E)
        nxt, rew, done, info = env.step(act)
        nxt = nxt.copy()
        steps_alive += 1
        for e in range(E):
            x, xd, th, thd, g, L, mc, mp, fm = obs[e]
            f = fm if act[e] == 1 else -fm
            tot = mp + mc
            temp = (f + mp * L * thd * thd * math.sin(th)) / tot
            thacc = (g * math.sin(th) - math.cos(th) * temp) / (L * (4.0 / 3.0 - mp * math.cos(th) ** 2 / tot))
            xacc = temp - mp * L * thacc * math.cos(th) / tot
            want = np.array([x + 0.02 * xd, xd + 0.02 * xacc, th + 0.02 * thd, thd + 0.02 * thacc, g, L, mc, mp, fm])
            fell = abs(want[0]) > 2.4 or abs(want[2]) > 12 * 2 * math.pi / 360
            timed = steps_alive[e] >= 40
            assert bool(done[e]) == (fell or timed), (t, e)
            if done[e]:
                ended_by_fall += fell; ended_by_time += (timed and not fell)
                assert (nxt[e] >= lo).all() and (nxt[e] <= hi).all()                   # a fresh episode: all nine columns re-drawn
                assert not np.array_equal(nxt[e, 4:], obs[e, 4:])
                steps_alive[e] = 0
            else:
                np.testing.assert_allclose(nxt[e], want, rtol=1e-12, atol=1e-14)
                assert np.array_equal(nxt[e, 4:], obs[e, 4:])                          # physics parameters fixed within the episode
        assert np.array_equal(rew, np.ones(E)) and len(info) == E and info[0]["env_reward"] == 1.0 and "env_reward" in info[0]
        obs = nxt
    assert ended_by_fall > 5 and ended_by_time > 0
    # same seed -> same trajectory (numpy default_rng(seed), what gymnasium's seeding.np_random(seed) builds)
    a, b = CartPoleVec(4, seed=11), CartPoleVec(4, seed=11)
    assert np.array_equal(a.reset(), b.reset()) and np.array_equal(a.reset(), np.random.default_rng(11).uniform(a.low, a.high, (2, 4, 9))[1])
    # create_cartpole: second entry of every range for the validation env; `_v` hyper-parameters override it (config.yml cartpole: degrees_v 9, h_range_v 1.8)
    hp = {"n_envs": 8, "degrees_v": 9, "h_range_v": 1.8}
    tr, va = create_cartpole(hp, False, seed=1), create_cartpole(hp, True, seed=1)
    assert tr.n_envs == 8 and tr.x_threshold == 2.4 and abs(tr.theta_threshold - 12 * 2 * math.pi / 360) < 1e-15
    assert va.x_threshold == 1.8 and abs(va.theta_threshold - 9 * 2 * math.pi / 360) < 1e-15
    assert tr.low[4] == CARTPOLE_PARAM_RANGE["min_gravity"][0] and va.low[4] == 10.4 and va.high[4] == 24.8 and va.high[5] == 2.0 and va.high[6] == 3.0
    o = va.reset()
    assert (o[:, 4] >= 10.4).all() and (o[:, 4] <= 24.8).all() and (o[:, 7] >= 0.2).all()


def test_step_info_behaves_like_the_list_of_dicts():
    """StepInfo (column-wise step info) under the uses the reference makes of a step's `info` list: len, indexing, iteration, key tests
    on an entry, np.array(info)[mask] is replaced by column(key) / info[i]; join() over env groups keeps env order."""
    from common.env.vec_envs import StepInfo
    rows = [{"prev_level_seed": 7 + i, "level_complete": i % 2} for i in range(3)]
    a = StepInfo(3, {"env_reward": np.array([1.0, 2.0, 3.0])}, rows=rows)
    b = StepInfo(2, {"env_reward": np.array([4.0, 5.0]), "prev_level_seed": np.array([1, 2])})
    assert len(a) == 3 and a[1] == {"prev_level_seed": 8, "level_complete": 1, "env_reward": 2.0} and "env_reward" in a[0]
    assert a.has("prev_level_seed") and a.has("env_reward") and not a.has("env_done")
    assert [d["env_reward"] for d in a] == [1.0, 2.0, 3.0] and a[-1]["prev_level_seed"] == 9
    j = StepInfo.join([a, b])
    assert len(j) == 5 and j[3]["env_reward"] == 4.0 and j[4]["prev_level_seed"] == 2 and j[0]["prev_level_seed"] == 7
    assert np.array_equal(j.column("env_reward"), [1, 2, 3, 4, 5]) and np.array_equal(j.column("prev_level_seed"), [7, 8, 9, 1, 2])
    plain = StepInfo.join([[{"k": 1}, {"k": 2}], [{"k": 3}]])                      # groups that speak the plain protocol
    assert len(plain) == 3 and plain[2] == {"k": 3} and np.array_equal(plain.column("k"), [1, 2, 3])
