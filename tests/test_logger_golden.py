"""common/logger.py (Logger.feed / dump) and Storage.fetch_log_data against what the REFERENCE's Logger and Storage produced on the
same reward / done / info streams (fixture G10, tests/golden/make_golden.py::g10_logger): every CSV column of every dumped row.
CPU only: the logger is host code either side of the accelerated path."""
import csv
import io
import json
import os

import numpy as np
import pytest

from conftest import load_npz


def _j(z, k):
    return json.loads(bytes(z[k]).decode())


@pytest.mark.parametrize("variant", ["info", "plain"])
def test_logger_rows_match_reference(variant, tmp_path):
    import torch
    from common.logger import Logger
    from common.storage import Storage
    z = load_npz("g10_logger.npz")
    T, E, K = 64, 8, 3
    cols, rows = _j(z, f"{variant}/columns"), _j(z, f"{variant}/rows")
    logger = Logger(E, str(tmp_path))
    logger.max_steps = 37
    assert logger.columns == cols
    st, stv = (Storage((9,), 64, T, E, torch.device("cpu")) for _ in range(2))
    for k in range(K):
        streams = []
        for which, storage in (("t", st), ("v", stv)):
            rew, raw, done, seed = (z[f"{variant}/{k}/{which}/{x}"] for x in ("rew", "raw", "done", "seed"))
            for t in range(T):
                info = [{"env_reward": raw[t, e], "prev_level_seed": int(seed[t, e])} for e in range(E)] if variant == "info" else [{} for _ in range(E)]
                storage.note_stored(rew[t], done[t], info)          # host half of Storage.store (the device half needs an engine)
            with np.errstate(all="ignore"):
                streams.append(storage.fetch_log_data())
        (rb, db, tm), (rbv, dbv, tmv) = streams
        np.testing.assert_array_equal(np.asarray(rb, np.float64), z[f"{variant}/{k}/fetch_t_rew"])
        np.testing.assert_array_equal(np.asarray(db, np.float64), z[f"{variant}/{k}/fetch_t_done"])
        logger.feed(rb, db, tm, rbv, dbv, tmv)
        summary = {'Loss/pi': 0.1 * k, 'Loss/v': -0.2, 'Loss/entropy': 2.7, 'Loss/x_entropy': 0.0, 'Loss/atn_entropy': float("nan"),
                   'Loss/atn_entropy2': float("nan"), 'Loss/sparsity': float("nan"), 'Loss/feature_sparsity': 0.8, 'Loss/total': 1.5 - k}
        with np.errstate(all="ignore"):
            logger.dump(summary, 5e-4 * (1 - k / K))
        mine, ref = logger.rows[-1], rows[k]
        for c, a, b in zip(cols, mine, ref):
            if c == "wall_time":
                continue
            a = float("nan") if a is None else float(a)
            b = float("nan") if b is None else float(b)
            if np.isnan(b):
                assert np.isnan(a), (k, c, a, b)
            elif c in ("timesteps", "num_episodes") or "_len" in c or "timeouts" in c:
                assert a == b, (k, c, a, b)                     # counts and lengths: exact
            else:
                # episode returns: the reference adds float32 rewards pairwise in float32, here they are added in float64
                assert abs(a - b) <= 2e-6 * max(1.0, abs(b)), (k, c, a, b)
    # the CSV on disk: same header, same number of rows, same values
    ref_csv = list(csv.reader(io.StringIO(bytes(z[f"{variant}/csv"]).decode())))
    my_csv = list(csv.reader(open(os.path.join(str(tmp_path), "log-append.csv"))))
    assert my_csv[0] == ref_csv[0] and len(my_csv) == len(ref_csv) == K + 1
    assert logger.episode_reward_buffer is logger.train.rewards and len(logger.episode_reward_buffer) == 40
