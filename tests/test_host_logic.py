"""CPU tests of the host side: the C-ABI library loads and exports every symbol include/mi355ppo.h declares,
fails loudly without a GPU, the parameter layout / init / index-stream logic is bit-exact with the reference
(golden vectors), and the data-parallel sharding helpers are consistent."""
import hashlib
import json
import os
import re
import zlib

import numpy as np
import pytest
import torch

from conftest import GOLD, ROOT, load_npz, npz_json, npz_params


def test_library_exports_every_declared_symbol():
    from mi355 import engine as M
    hdr = open(os.path.join(ROOT, "include", "mi355ppo.h")).read()
    declared = sorted(set(re.findall(r"\b(mi_[a-z0-9_]+)\s*\(", hdr)))
    lib = M.load_library()
    missing = [n for n in declared if not hasattr(lib, n)]
    assert not missing, missing
    assert sorted(set(M.EXPORTS)) == [d for d in declared if d in M.EXPORTS]
    assert set(declared) - set(M.EXPORTS) == set(), set(declared) - set(M.EXPORTS)


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU failure mode")
def test_no_cpu_fallback():
    from mi355.engine import Engine, EngineError
    with pytest.raises(EngineError, match="no CPU fallback"):
        Engine("impala", 4, 4, 15, 8)
    from common.model import ImpalaModel
    from common.policy import CategoricalPolicy
    pol = CategoricalPolicy(ImpalaModel(3), False, 15)
    with pytest.raises(NotImplementedError):
        pol(torch.zeros(2, 3, 64, 64), None, None)          # no engine attached -> no compute path


def test_init_is_bit_identical_to_reference():
    """Same seed => same weights as the reference constructors (sha256 of the flat parameter vector)."""
    from common.model import ImpalaModel, MLPModel
    from common.policy import CategoricalPolicy
    z = load_npz("g3_impala_forward.npz")
    shas = npz_json(z, "sha")
    for A in (15, 9):
        torch.manual_seed(6033)
        p = CategoricalPolicy(ImpalaModel(in_channels=3), False, A)
        flat = np.concatenate([x.detach().numpy().ravel() for x in p.parameters()]).astype(np.float32)
        assert hashlib.sha256(flat.tobytes()).hexdigest() == shas[f"A{A}"]
        assert sum(x.numel() for x in p.parameters()) == (626256 if A == 15 else 624714)
    assert list(p.state_dict().keys()) == [k[2:] for k in z.files if k.startswith("p/")]
    z7 = load_npz("g7_mlp_forward.npz")
    torch.manual_seed(6033)
    p = CategoricalPolicy(MLPModel(9, 4, 256, 64), False, 2)
    flat = np.concatenate([x.detach().numpy().ravel() for x in p.parameters()]).astype(np.float32)
    assert hashlib.sha256(flat.tobytes()).hexdigest() == bytes(z7["sha"]).decode()
    assert list(p.state_dict().keys()) == [k[2:] for k in z7.files if k.startswith("p/")]


def test_recurrent_policy_init_and_keys_match_reference():
    from common.model import ImpalaModel
    from common.policy import CategoricalPolicy
    z = load_npz("g9_recurrent_predict.npz")
    torch.manual_seed(6033)
    p = CategoricalPolicy(ImpalaModel(3), True, 15)
    flat = np.concatenate([x.detach().numpy().ravel() for x in p.parameters()]).astype(np.float32)
    assert hashlib.sha256(flat.tobytes()).hexdigest() == bytes(z["sha_all"]).decode()
    assert list(p.state_dict().keys()) == npz_json(z, "keys")
    assert len(p.param_shapes()) == 36                       # the frozen GRU is not part of the trained flat vector


def test_layout_roundtrip_and_order():
    from mi355 import layout
    z = load_npz("g3_impala_forward.npz")
    shapes = layout.impala_param_shapes(15)
    assert list(shapes.keys()) == [k[2:] for k in z.files if k.startswith("p/")]
    flat = layout.flatten(shapes, npz_params(z))
    assert flat.size == 626256
    back = layout.unflatten(shapes, flat)
    assert all(np.array_equal(back[k], npz_params(z)[k]) for k in shapes)
    with pytest.raises(ValueError):
        layout.unflatten(shapes, flat[:-1])
    assert sum(int(np.prod(s)) for s in layout.mlp_param_shapes(2, 9, 4, 256, 64).values()) == 150787


def test_numa_helper_parses_cpu_lists_and_is_a_no_op_without_a_gpu(monkeypatch):
    """mi355.numa: sysfs cpu lists -> CPU sets; no GPU / unreadable topology / MI355_NUMA_PIN=0 / too few local CPUs leave the
    affinity alone and return None."""
    from mi355 import numa
    assert numa._parse_cpulist("64-67,192-193\n") == {64, 65, 66, 67, 192, 193}
    assert numa._parse_cpulist("5") == {5} and numa._parse_cpulist("\n") == set()
    before = os.sched_getaffinity(0)
    assert numa.pin_to_gpu_node(0) is None or numa.pin_to_gpu_node(0) <= before          # (a GPU box pins; this container has no GPU)
    monkeypatch.setattr(numa, "gpu_numa_cpus", lambda i=0: {next(iter(before))})
    assert numa.pin_to_gpu_node(0, min_cpus=2) is None and os.sched_getaffinity(0) == before      # one local CPU < min_cpus
    monkeypatch.setenv("MI355_NUMA_PIN", "0")
    monkeypatch.setattr(numa, "gpu_numa_cpus", lambda i=0: set(before))
    assert numa.pin_to_gpu_node(0, min_cpus=1) is None and os.sched_getaffinity(0) == before


def test_single_threaded_permutation_draw_is_torch_randperm():
    """The permutation is drawn with one intra-op thread (3 ms of idle GPU per optimize() otherwise): same numbers as torch.randperm at
    the production size, the generator advanced identically, the caller's thread count put back."""
    from common.storage import _randperm_serial
    k = torch.get_num_threads()
    for n in (7, 4096, 65536):
        torch.manual_seed(123 + n)
        want = torch.randperm(n).numpy()
        after_want = torch.rand(1).item()
        torch.manual_seed(123 + n)
        got = _randperm_serial(n)
        assert np.array_equal(got, want) and torch.rand(1).item() == after_want
    assert torch.get_num_threads() == k


def test_index_streams_bit_exact_with_reference():
    """Storage.minibatch_index_stream consumes the torch CPU generator exactly like the reference's
    BatchSampler(SubsetRandomSampler) / randperm(E) (golden G2)."""
    from common.storage import Storage
    rec = json.load(open(os.path.join(GOLD, "g2_perm.json")))
    for key, r in rec.items():
        if key.startswith("rec_"):
            _, s, e = key.split("_")
            E = int(e[1:])
            st = Storage((9,), 4, 4, E, None)
            torch.manual_seed(int(s[1:]))
            groups = list(st.minibatch_index_stream(4 * E, recurrent=True))
            assert len(groups) == 1
            envs = groups[0][:E]                                   # first time step: t = 0 -> flat index == env
            assert envs[:16].tolist() == r["first16"]
            assert zlib.crc32(envs.astype(np.int64).tobytes()) == r["crc_all"]
            assert np.array_equal(groups[0].reshape(4, E), np.arange(4)[:, None] * E + envs[None, :])   # time-major
            continue
        s, T, E, B = key.split("_")
        seed, T, E, B = int(s[1:]), int(T[1:]), int(E[1:]), int(B[1:])
        st = Storage((9,), 4, T, E, None)
        torch.manual_seed(seed)
        chunks = list(st.minibatch_index_stream(B)) + list(st.minibatch_index_stream(B))
        assert len(chunks) == r["n_chunks"]
        allidx = np.concatenate(chunks)
        assert allidx.dtype == np.int64
        assert allidx[:16].tolist() == r["first16"] and allidx[-16:].tolist() == r["last16"]
        assert zlib.crc32(allidx.tobytes()) == r["crc_all"]
        assert [zlib.crc32(c.tobytes()) for c in chunks] == r["crc_chunks"]


def test_recurrent_groups_match_reference_generator():
    from common.storage import Storage
    z = load_npz("g8_recurrent.npz")
    meta = npz_json(z, "meta")
    T, E, B = meta["T"], meta["E"], meta["B"]
    st = Storage((5,), 6, T, E, None)
    torch.manual_seed(meta["seed"])
    groups = list(st.minibatch_index_stream(B, recurrent=True))
    assert len(groups) == len(meta["shapes"])
    val = z["val"][:-1].reshape(-1)
    act = z["act"].reshape(-1)
    for idx, sh, first in zip(groups, meta["shapes"], meta["first"]):
        assert len(idx) == sh[0][0]
        np.testing.assert_array_equal(act[idx], np.asarray(first["act"], dtype=np.float32))
        np.testing.assert_allclose(val[idx], first["val"], atol=1e-7)


def test_shard_indices_partition_and_order():
    from mi355.dist import env_range, merge_adv_stats, shard_indices
    T, E, W = 16, 24, 4
    rng = np.random.default_rng(0)
    chunk = rng.permutation(T * E)[:100]
    seen = []
    for r in range(W):
        loc = shard_indices(chunk, E, r, W)
        e0, e1 = env_range(E, r, W)
        El = e1 - e0
        t, e = loc // El, loc % El + e0
        glob = t * E + e
        # order preserved, every element belongs to the shard
        assert np.array_equal(glob, chunk[(chunk % E >= e0) & (chunk % E < e1)])
        seen.append(glob)
    assert sorted(np.concatenate(seen).tolist()) == sorted(chunk.tolist())
    assert np.array_equal(shard_indices(chunk, E, 0, 1), chunk)
    with pytest.raises(ValueError):
        env_range(10, 0, 4)
    # Chan merge == statistics of the concatenation
    parts = [rng.standard_normal(n) * (k + 1) + k for k, n in enumerate((5, 1000, 37))]
    stats = [(len(p), p.mean(), ((p - p.mean()) ** 2).sum()) for p in parts]
    n, mean, m2 = merge_adv_stats(stats)
    allp = np.concatenate(parts)
    assert n == len(allp) and abs(mean - allp.mean()) < 1e-12 and abs(m2 / (n - 1) - allp.var(ddof=1)) < 1e-10


def test_as_device_obs_is_lossless():
    from common.model import as_device_obs
    u8 = np.random.default_rng(1).integers(0, 256, (3, 64, 64, 3), dtype=np.uint8)
    ref_obs = u8.transpose(0, 3, 1, 2) / 255.0                       # what the reference's wrappers hand the agent
    assert np.array_equal(as_device_obs(ref_obs, "impala"), u8)
    assert np.array_equal(as_device_obs(torch.from_numpy(ref_obs.astype(np.float32)), "impala"), u8)
    assert np.array_equal(as_device_obs(u8, "impala"), u8)
    # and the table the kernels use reproduces the reference's float32 observation exactly
    lut = (np.arange(256) / 255.0).astype(np.float32)
    assert np.array_equal(lut[u8].transpose(0, 3, 1, 2), ref_obs.astype(np.float32))


def test_lr_schedule_and_config():
    import yaml
    from common.misc_util import adjust_lr

    class Opt:
        param_groups = [{"lr": 1.0}]
    o, lr = adjust_lr(Opt(), 5e-4, 65536, 200_000_000)
    assert lr == 5e-4 * (1 - 65536 / 200_000_000) and o.param_groups[0]["lr"] == lr
    cfg = yaml.safe_load(open(os.path.join(ROOT, "train-procgen-pytorch_amd", "hyperparams", "procgen", "config.yml")))
    hp = cfg["hard-500"]
    assert (hp["n_envs"], hp["n_steps"], hp["mini_batch_size"], hp["epoch"]) == (256, 256, 8192, 3)
    assert cfg["easy"]["mini_batch_size"] == 2048 and cfg["cartpole"]["architecture"] == "mlpmodel"


def test_frame_byte_to_bf16_needs_no_table():
    """block1.conv (bf16 mode) stages a frame byte k as bf16(k/255).  The kernel computes k * fp32(1/255) and rounds to
    bf16 (csrc/conv_bf16.hip c1_store); for all 256 bytes that equals rounding the correctly-rounded fp32 quotient
    (what the reference's ScaledFloatFrame produces before a bf16 cast)."""
    k = torch.arange(256, dtype=torch.float32)
    quotient = (k / 255.0).bfloat16()
    product = (k * torch.tensor(1.0 / 255.0, dtype=torch.float32)).bfloat16()
    assert torch.equal(quotient.view(torch.int16), product.view(torch.int16))


def test_update_plan_positions_are_the_rows_places_in_the_global_minibatch():
    """fs_coef != 0 on several ranks: a pass carries, for every local sample, its position in the global minibatch --
    ascending, disjoint between the ranks, together covering 0 .. B-1, and mapping the local indices back to the chunk's entries."""
    from mi355.dist import update_plan, shard_indices
    T, EG, B, world = 8, 16, 32, 4
    torch.manual_seed(3)
    chunks = [torch.randperm(T * EG).numpy()[k * B:(k + 1) * B] for k in range(4)]
    seen = [np.zeros(B, int) for _ in chunks]
    for rank in range(world):
        ops = list(update_plan(iter(chunks), rank, world, EG, 1, False, True, B, with_positions=True))
        mbs = [op for op in ops if op[0] == "minibatch"]
        assert len(mbs) == len(chunks) and [op[0] for op in ops].count("stats") == len(chunks)
        for k, (_, local, seg_n, n_global, pos) in enumerate(mbs):
            assert n_global == B and seg_n == [len(local)] and len(pos) == len(local) and pos.dtype == np.int32
            assert np.all(np.diff(pos) > 0)
            seen[k][pos] += 1
            e_per = EG // world
            glob = chunks[k][pos]                                  # the global flat indices at those positions ...
            t, e = glob // EG, glob % EG
            assert np.all(e // e_per == rank)                      # ... belong to this rank's env range
            np.testing.assert_array_equal(local, t * e_per + (e - rank * e_per))      # and are its local indices, in order
            np.testing.assert_array_equal(local, shard_indices(chunks[k], EG, rank, world))
    for s in seen:
        assert np.all(s == 1)


def test_permutation_drawn_ahead_is_the_in_place_permutation():
    """Storage.draw_permutation_ahead (PPO.train: the next update's first torch.randperm, drawn by a helper thread behind the rollout):
    the minibatch stream is bit-identical to the in-place draw -- same generator, same position, later epochs continue from it -- and a
    permutation made stale by a re-seed (what every parity test does right before optimize()) is dropped, not used."""
    import torch
    from common.storage import Storage
    st = Storage((3, 64, 64), 256, 16, 8, None)

    def epochs(k, B=32, rec=False):
        return [c.copy() for _ in range(k) for c in st.minibatch_index_stream(B, rec)]
    torch.manual_seed(123)
    want = epochs(3)
    torch.manual_seed(123)
    st.draw_permutation_ahead(16 * 8)
    got = epochs(3)
    assert len(got) == 12 and all(np.array_equal(a, b) for a, b in zip(want, got))
    # stale: drawn, then the caller re-seeds -> the stream is the re-seeded one
    torch.manual_seed(7)
    st.draw_permutation_ahead(16 * 8)
    st._perm_ahead._t.join()
    torch.manual_seed(123)
    got = epochs(3)
    assert all(np.array_equal(a, b) for a, b in zip(want, got))
    # wrong length (e.g. the policy's kind changed): dropped as well; recurrent streams take E-permutations
    torch.manual_seed(5); want_r = epochs(1, 32, True)
    torch.manual_seed(5); st.draw_permutation_ahead(8); got_r = epochs(1, 32, True)
    assert all(np.array_equal(a, b) for a, b in zip(want_r, got_r))
    torch.manual_seed(5); st.draw_permutation_ahead(999); st._perm_ahead._t.join(); torch.manual_seed(5)
    assert all(np.array_equal(a, b) for a, b in zip(want_r, epochs(1, 32, True)))


def test_update_plan_c4_shape_eight_ranks_in_process():
    """BASELINE config 4's update schedule (E = 2048 over 8 ranks, T = 256: N = 524 288, global minibatch 8192, 8 accumulated minibatches per
    optimizer step) as mi355/dist.py::update_plan emits it for each of the 8 ranks -- pure index logic, all ranks in this process:
    the same operation sequence on every rank (what keeps their collectives matched), 8 optimizer steps per epoch, every pass within
    max_batch and its segments adding up, and over the ranks every global sample visited exactly once per epoch, by the rank that
    owns its env."""
    import torch
    from mi355.dist import update_plan, env_range
    T, E, R, B = 256, 2048, 8, 8192
    N = T * E
    torch.manual_seed(0)
    perm = torch.randperm(N).numpy()
    chunks = [perm[k * B:(k + 1) * B].astype(np.int64) for k in range(N // B)]
    max_batch = int(B * 1.07) + 64
    seqs, seen = [], np.zeros(N, np.int32)
    for r in range(R):
        ops = list(update_plan(iter(chunks), r, R, E, (N // 8) / B, True, False, max_batch))
        seqs.append([(o[0], len(o[2]) if o[0] == "minibatch" else None, o[3] if o[0] == "minibatch" else None) for o in ops])
        e0, e1 = env_range(E, r, R)
        for o in ops:
            if o[0] != "minibatch":
                continue
            local, seg = o[1], o[2]
            assert len(local) == sum(seg) <= max_batch and o[3] == B
            t, e = local // (e1 - e0), local % (e1 - e0)
            seen[t * E + e0 + e] += 1
    assert all(s == seqs[0] for s in seqs)                       # rank-uniform schedule
    kinds = [k for k, _, _ in seqs[0]]
    assert kinds.count("step") == 8 and kinds.count("minibatch") == 8 and kinds[-1] == "log" and seqs[0][0] == ("minibatch", 8, B)
    assert (seen == 1).all()
