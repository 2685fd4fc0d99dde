"""Episode statistics + CSV logging with the reference's column set (common/logger.py:13-174), with the
per-step Python double loop of Logger.feed (:119-147) replaced by per-env segment sums over `done`."""
import csv
import os
import time
from collections import deque

import numpy as np

EPISODE_KEYS = ["max_episode_rewards", "mean_episode_rewards", "median_episode_rewards", "min_episode_rewards",
                "max_episode_len", "mean_episode_len", "min_episode_len", "mean_timeouts",
                "mean_episode_len_pos_reward", "balanced_mean_rewards"]
LOSS_KEYS = ["loss_pi", "loss_v", "loss_entropy", "loss_x_entropy", "atn_entropy", "atn_entropy2", "loss_sparsity",
             "loss_feature_sparsity", "loss_total"]


class _EpisodeTracker:
    def __init__(self, n_envs):
        self.run_rew = np.zeros(n_envs, dtype=np.float64)
        self.run_len = np.zeros(n_envs, dtype=np.int64)
        self.rewards, self.lens, self.timeouts = deque(maxlen=40), deque(maxlen=40), deque(maxlen=40)
        self.true_mean = None

    def feed(self, rew, done, max_steps):
        """rew/done (T,E).  Episodes close in (env-major, time-minor) order like the reference's loops."""
        T, E = rew.shape
        n_new = 0
        for e in range(E):
            ends = np.nonzero(done[:, e])[0]
            start = 0
            for t in ends:
                total = self.run_rew[e] + rew[start:t + 1, e].sum()
                length = self.run_len[e] + (t + 1 - start)
                self.rewards.append(total); self.lens.append(int(length)); self.timeouts.append(1 if length == max_steps else 0)
                self.run_rew[e], self.run_len[e] = 0.0, 0
                start = t + 1
                n_new += 1
            self.run_rew[e] += rew[start:, e].sum()
            self.run_len[e] += T - start
        return n_new

    def stats(self):
        r, l = np.array(self.rewards, dtype=np.float64), np.array(self.lens, dtype=np.float64)
        mean = lambda a: float(np.mean(a)) if len(a) else float("nan")
        return [float(np.max(r, initial=0)), mean(r), float(np.median(r)) if len(r) else float("nan"), float(np.min(r, initial=0)),
                float(np.max(l, initial=0)), mean(l), float(np.min(l, initial=0)), mean(np.array(self.timeouts)),
                mean(l[r > 0]) if len(r) else float("nan"), self.true_mean]


class Logger(object):
    def __init__(self, n_envs, logdir, use_wandb=False, has_vq=False, algo="ppo", greedy=False):
        self.n_envs, self.logdir, self.use_wandb = n_envs, logdir, use_wandb
        self.start_time = time.time()
        self.max_steps = 10 ** 3
        self.train, self.valid = _EpisodeTracker(n_envs), _EpisodeTracker(n_envs)
        self.columns = (["timesteps", "wall_time", "num_episodes"] + EPISODE_KEYS + ["val_" + k for k in EPISODE_KEYS]
                        + ["ema_rewards"] + LOSS_KEYS + ["learning_rate"])
        self.rows = []
        self.timesteps = 0
        self.num_episodes = 0

    @property
    def episode_reward_buffer(self):
        return self.train.rewards

    def feed(self, rew_batch, done_batch, true_mean_reward, rew_batch_v=None, done_batch_v=None, true_mean_reward_v=None, *_):
        self.train.true_mean, self.valid.true_mean = true_mean_reward, true_mean_reward_v
        self.num_episodes += self.train.feed(np.asarray(rew_batch), np.asarray(done_batch) > 0, self.max_steps)
        if rew_batch_v is not None and done_batch_v is not None:
            self.valid.feed(np.asarray(rew_batch_v), np.asarray(done_batch_v) > 0, self.max_steps)
        self.timesteps += self.n_envs * np.asarray(rew_batch).shape[0]

    def dump(self, summary={}, lr=0.):
        wall = time.time() - self.start_time
        ts, vs = self.train.stats(), self.valid.stats()
        ema = ts[1]
        if self.rows:
            k = .99 / (1 + len(self.rows))
            ema = ema * k + self.rows[-1][self.columns.index("ema_rewards")] * (1 - k)
        row = [self.timesteps, wall, self.num_episodes] + ts + vs + [ema] + list(summary.values()) + [lr]
        self.rows.append(row)
        if self.logdir:
            path = os.path.join(self.logdir, "log-append.csv")
            new = not os.path.exists(path) or os.path.getsize(path) == 0
            with open(path, "a") as f:
                w = csv.writer(f)
                if new:
                    w.writerow(self.columns)
                w.writerow(row)
        print("  ".join(f"{c}={v:.4g}" if isinstance(v, (int, float, np.floating)) and v is not None else f"{c}={v}"
                        for c, v in zip(self.columns, row) if c in ("timesteps", "mean_episode_rewards", "mean_episode_len",
                                                                    "loss_pi", "loss_v", "loss_entropy", "loss_total", "learning_rate")))
        if self.use_wandb:
            import wandb
            wandb.log(dict(zip(self.columns, row)))
