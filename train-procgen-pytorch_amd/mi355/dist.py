"""Data-parallel plumbing: the n_envs dimension is sharded over ranks (one process per GPU).

Every rank draws the SAME torch.randperm(N_global) (same seed, same stream position), forms the same
global minibatches as the single-process reference (common/storage.py:86-91) and keeps the indices
whose env falls in its shard.  Means are taken over the global minibatch (each rank scales its
partial sums by 1/B_global); gradients are summed with ONE all-reduce of the flat gradient buffer per
optimizer step (RCCL over xGMI via torch.distributed backend "nccl"; "gloo" in the CPU tests)."""
import numpy as np


def env_range(n_envs_global, rank, world):
    if n_envs_global % world != 0:
        raise ValueError(f"n_envs={n_envs_global} is not divisible by world_size={world}")
    per = n_envs_global // world
    return rank * per, (rank + 1) * per


def shard_indices(chunk, n_envs_global, rank, world, with_positions=False):
    """Global flat indices i = t*E + e of one minibatch -> this rank's LOCAL flat indices
    t*E_local + (e - e0), in the order they appear in the chunk (with_positions: also their positions in the chunk)."""
    chunk = np.asarray(chunk, dtype=np.int64)
    if world == 1:
        return (chunk, np.arange(len(chunk), dtype=np.int32)) if with_positions else chunk
    e0, e1 = env_range(n_envs_global, rank, world)
    t, e = chunk // n_envs_global, chunk % n_envs_global
    keep = (e >= e0) & (e < e1)
    local = t[keep] * (e1 - e0) + (e[keep] - e0)
    return (local, np.nonzero(keep)[0].astype(np.int32)) if with_positions else local


def merge_adv_stats(stats_list):
    """Chan et al. parallel merge of per-rank {count, mean, M2} (fp64) -> global triple, from which
    mean and the unbiased std of common/storage.py:79 follow."""
    n, mean, m2 = 0.0, 0.0, 0.0
    for c, mu, s in stats_list:
        if c == 0:
            continue
        tot = n + c
        d = mu - mean
        mean = mean + d * c / tot
        m2 = m2 + s + d * d * n * c / tot
        n = tot
    return np.array([n, mean, m2], dtype=np.float64)


def update_plan(chunks, rank, world, n_envs_global, grad_accumulation_steps, merge, stats_per_minibatch, max_batch, max_segments=16,
                with_positions=False):
    """The host schedule of PPO.optimize (agents/ppo.py:155-177) for one rank, as a stream of operations -- pure index logic, no
    device: agents/ppo.py executes it on the engine, tests/test_dist_gloo.py on two CPU ranks.

    chunks: the global minibatches (flat index arrays, all epochs back to back; every rank sees the same ones).  Yields
      ("minibatch", local_idx, seg_n, n_global)  one pass over this rank's samples of len(seg_n) global minibatches (1 unless `merge`:
                                                 accumulated minibatches whose gradients are summed anyway, <= max_batch samples and
                                                 <= max_segments minibatches per pass, never across an optimizer step)
      ("stats",)                                 only with stats_per_minibatch (x-entropy term on > 1 rank): all-reduce the loss
                                                 statistics of the pass just issued, then finish it (backward)
      ("step",)                                  after every grad_accumulation_steps-th minibatch (the reference's `cnt % steps == 0`
                                                 with a float): gradient all-reduce + optimizer step
      ("log", n_minibatches)                     once at the end: reduce / read the per-minibatch records
    with_positions (fs_coef != 0 on > 1 rank, not with `merge`): a "minibatch" op carries a fifth element, the positions of the local
    samples in their global minibatch (the feature-sparsity term's tie rule needs the global order).
    """
    assert not (merge and with_positions)
    assert not (merge and stats_per_minibatch)
    held, held_n, held_global, n_mb, cnt = [], 0, None, 0, 1

    def flush():
        nonlocal held, held_n
        if held:
            out = ("minibatch", np.concatenate(held), [len(h) for h in held], held_global)
            held, held_n = [], 0
            return out
        return None

    for chunk in chunks:
        if with_positions:
            local, pos = shard_indices(chunk, n_envs_global, rank, world, True)
        else:
            local = shard_indices(chunk, n_envs_global, rank, world)
        n_mb += 1
        if merge:
            if held_n + len(local) > max_batch or len(held) == max_segments or (held and len(chunk) != held_global):
                op = flush()
                if op:
                    yield op
            held.append(local)
            held_n += len(local)
            held_global = len(chunk)
        else:
            yield ("minibatch", local, [len(local)], len(chunk), pos) if with_positions else ("minibatch", local, [len(local)], len(chunk))
            if stats_per_minibatch:
                yield ("stats",)
        if cnt % grad_accumulation_steps == 0:
            op = flush()
            if op:
                yield op
            yield ("step",)
        cnt += 1
    op = flush()
    if op:
        yield op
    yield ("log", n_mb)


class Collective:
    """Minimal wrapper so that the agent code is identical for 1 rank, gloo (CPU tensors in tests) and
    nccl/RCCL (device tensors wrapping the engine's buffers)."""

    def __init__(self, group=None):
        import torch.distributed as td
        self.td = td
        self.active = td.is_available() and td.is_initialized() and td.get_world_size(group) > 1
        self.group = group
        self.rank = td.get_rank(group) if self.active else 0
        self.world = td.get_world_size(group) if self.active else 1

    def attach_native(self, engine):
        """RCCL inside the library (include/mi355ppo.h mi_comm_*): with the "nccl" backend the engine gets its own communicator
        (rank 0's 128-byte id reaches the others through this process group) and gradient / statistics collectives run behind the
        C ABI, the gradient exchange overlapped with the backward pass.  gloo (CPU tests, --rehearse-on-one-gpu) keeps the
        torch.distributed calls on aliased device buffers below."""
        # OPT-IN (MI355_NATIVE_COMM=1): the in-library path has run at world size 1 only (no multi-GPU node was available to any build
        # round), so the default multi-GPU route is torch.distributed's own RCCL process group on the aliased buffers -- the schedule
        # the two-rank tests execute.  tests/test_gpu_agent.py::test_native_rccl_equals_torch_distributed_on_two_gpus compares the two
        # (armed, unarmed, torch.distributed) bit for bit wherever two GPUs are visible.
        import os
        if not self.active or self.td.get_backend(self.group) != "nccl" or os.environ.get("MI355_NATIVE_COMM", "0") != "1":
            return False
        box = [engine.comm_unique_id() if self.rank == 0 else None]
        self.td.broadcast_object_list(box, src=0, group=self.group)
        engine.comm_init(box[0], self.rank, self.world)
        return True

    def allreduce_sum_(self, tensor):
        if self.active:
            self.td.all_reduce(tensor, op=self.td.ReduceOp.SUM, group=self.group)
        return tensor

    def allreduce_max_(self, tensor):
        if self.active:
            self.td.all_reduce(tensor, op=self.td.ReduceOp.MAX, group=self.group)
        return tensor

    def allgather_f64(self, vec):
        import torch
        if not self.active:
            return [np.asarray(vec, dtype=np.float64)]
        t = torch.as_tensor(np.asarray(vec, dtype=np.float64))
        dev = torch.device("cuda") if self.td.get_backend(self.group) == "nccl" else torch.device("cpu")
        t = t.to(dev)
        out = [torch.empty_like(t) for _ in range(self.world)]
        self.td.all_gather(out, t, group=self.group)
        return [o.cpu().numpy() for o in out]


class DevicePointerTensor:
    """torch view of a raw device buffer owned by the engine (for RCCL all-reduce through
    torch.distributed): exposes __cuda_array_interface__ and lets torch.as_tensor alias it."""

    def __init__(self, ptr, n_elems, typestr="<f4"):
        self.__cuda_array_interface__ = {"shape": (int(n_elems),), "typestr": typestr, "data": (int(ptr), False),
                                         "version": 2, "strides": None}

    def tensor(self, device_index=0):
        import torch
        return torch.as_tensor(self, device=torch.device("cuda", device_index))
