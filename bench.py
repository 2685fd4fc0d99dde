#!/usr/bin/env python3
"""Headline benchmark: env steps/sec of the PPO hot path, coinrun hard-500 shape (IMPALA-CNN, T=256,
256 envs per GPU, 3 epochs x 8 minibatches of 8192), learner side, synthetic frames in pinned host memory.

One "step" (--steps) = one PPO iteration = T*E env steps: T+1 policy steps, each one uploading its E uint8
frames from pinned host memory (H2D inside the timed region), forward, sample, actions read back to the host,
rewards/dones uploaded; GAE + advantage normalisation; then epoch x n_minibatch minibatch updates (index gather
inside the first conv, forward, fused loss, backward, clip + Adam) -- everything agents/ppo.py:225-255 does
except env.step and logging ("rollout inference + H2D + GAE + epoch passes of minibatch updates").

    python bench.py                      # 1 GPU
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W     # weak scaling: 256 envs per GPU, global minibatch 8192
"""
import argparse
import json
import os
import sys
import time

if "HSA_ENABLE_IPC_MODE_LEGACY" not in os.environ:       # before anything initialises HIP: RCCL needs dmabuf IPC on this driver
    os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
# the pipelined rollout runs up to 4 env-group streams beside the main stream: with the runtime's default of 4 hardware queues two
# of them share a queue and serialise (measured: 4 groups 194 us per step with 4 queues, 134 us with 8)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "train-procgen-pytorch_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np
import torch
import yaml

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_F32_PEAK_TF = 157.3     # MI355X_MICROARCH.md: matrix fp32 (v_mfma_f32_*_f32), dense
MFMA_BF16_PEAK_TF = 2500.0   # MI355X_MICROARCH.md: bf16 MFMA, dense (the 5 PF headline is 2:1 sparse)


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def cpu_baseline(hp, E, A, t_sample, threads):
    """The CPU oracle (oracle/ppo_oracle.py, the restatement of the reference's PyTorch path, SURVEY 8(d) "CPU baseline") on the
    host cores, with the REFERENCE's data path: observations converted to fp32 NCHW on the host and kept in a (T+1,E,3,64,64) fp32
    host storage, one forward + sample per rollout step on the stored step, GAE on the host, then a full epoch of minibatches each
    GATHERED BY INDEX from that storage (common/storage.py:112-128), forward / backward / clip / Adam.  Bounded sample: t_sample
    rollout steps of the E envs instead of T = 256 (everything scales linearly in T; minibatch = N_sample / 8 keeps 8 optimizer
    steps per epoch); one epoch is timed and counted `epoch` times."""
    from oracle import ppo_oracle as O
    from common.model import ImpalaModel
    from common.policy import CategoricalPolicy
    torch.set_num_threads(threads)
    torch.manual_seed(6033)
    pol = CategoricalPolicy(ImpalaModel(3), False, A)
    params = {k: v.detach().numpy().copy() for k, v in pol.state_dict().items()}
    rng = np.random.default_rng(0)
    T = t_sample
    frames = rng.integers(0, 256, size=(T + 1, E, 64, 64, 3), dtype=np.uint8)
    p = {k: torch.from_numpy(v) for k, v in params.items()}
    with torch.no_grad():
        O.policy_forward(p, "impala", O.frames_to_obs(frames[0, :32]))                      # warm-up (thread pool, allocator)
    obs_store = torch.zeros(T + 1, E, 3, 64, 64)                                            # Storage.obs_batch (common/storage.py:21)
    act, logp = torch.zeros(T, E), torch.zeros(T, E)
    val = torch.zeros(T + 1, E)
    rew = torch.from_numpy(rng.standard_normal((T, E)).astype(np.float32))
    done = torch.from_numpy((rng.random((T, E)) < 0.01).astype(np.float32))
    t0 = time.time()
    with torch.no_grad():
        for t in range(T + 1):                                                              # PPO.predict + Storage.store (agents/ppo.py:225-236)
            obs_store[t] = O.frames_to_obs(frames[t])
            lp, v, _ = O.policy_forward(p, "impala", obs_store[t])
            val[t] = v
            if t < T:
                a, l = O.sample_actions(lp, torch.from_numpy(rng.random(E).astype(np.float32)))
                act[t], logp[t] = a.float(), l
    t_roll = time.time() - t0
    t0 = time.time()
    adv, ret = O.compute_estimates(rew, done, val, hp["gamma"], hp["lmbda"], hp["use_gae"], hp["normalize_adv"])
    t_gae = time.time() - t0
    N = T * E
    ag = O.OraclePPO(params, "impala", T, E, epoch=1, n_minibatch=8, mini_batch_size=N // 8, learning_rate=hp["learning_rate"],
                     gamma=hp["gamma"], lmbda=hp["lmbda"])
    t0 = time.time()
    ag.optimize(dict(obs=obs_store, act=act, logp=logp, val=val, ret=ret, adv=adv))         # 1 epoch: randperm + 8 gathered minibatches
    t_epoch = time.time() - t0
    total = t_roll + t_gae + hp["epoch"] * t_epoch
    return dict(value=N / total, unit="env steps/s", cores=threads, kind="port", cpu=cpu_model(),
                sample=f"{T} rollout steps x {E} envs ({N} env steps of the {hp['n_steps']} x {E} iteration): host fp32 (T+1,E,3,64,64) storage, "
                       f"{T + 1} forward+sample passes {t_roll:.1f} s, GAE {t_gae * 1e3:.0f} ms, one epoch of 8 index-gathered minibatches of {N // 8} "
                       f"(forward, backward, clip, Adam) {t_epoch:.1f} s, counted {hp['epoch']}x")


def fp32_record(args, iters=3):
    """The same iteration in the fp32 PARITY mode (the mode the 1e-4 loss / return parity with the reference is stated and tested in;
    exact-fp32 matrix instructions), `iters` timed iterations after one warm-up, product collector included: a driver-run number for it
    beside the bf16 headline.  Run as a CHILD process of this script (a second engine in this process lands its env-group streams on
    hardware queues the first one already occupies and its rollout serialises: 218 instead of 88 ms)."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--precision", "fp32", "--steps", str(iters), "--warmup", "1", "--no-cpu-baseline",
           "--no-fp32-record", "--param_name", args.param_name, "--n-actions", str(args.n_actions), "--rollout-groups", str(args.rollout_groups)]
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
        d = json.loads(r.stdout.strip().splitlines()[-1])
    except Exception as e:                                   # the headline line must not depend on this leg
        return {"error": f"{type(e).__name__}: {e}"}
    return {"value": d["value"], "unit": d["unit"], "ms_per_step": d["ms_per_step"], "steps": d["steps"], "dtype": d["dtype"],
            "rollout_ms": d["phase_ms_per_step"]["rollout"], "update_ms": d["phase_ms_per_step"]["update"], "loss_total": d["loss_total"],
            "whole_step_roofline_mfma_frac": d["whole_step_roofline"]["mfma_frac"],
            "dominant_kernel": None if d["roofline"] is None else {k: d["roofline"][k] for k in ("kernel", "avg_launch_ms", "mfma_TFps", "mfma_frac", "hbm_GBps")}}


def rocprof_name(cls, precision):
    """kernel class of the live profiler -> substrings that identify its rocprofv3 kernel name (template arguments)."""
    parts = cls.split("_")
    kind, cin, cout, hw = parts[1], *[int(x) for x in parts[2:5]]
    if parts[0] == "resblock":                        # fused residual block (bf16 mode): RbCfg<C, HW, TH, NIMG>, BWD
        if kind == "dgrad" and cin == 16:             # 16-channel blocks: data + weight gradients in one kernel
            return ["resblock_bwd_full16d_bf16_kernel"]      # (round 3: two roles + LDS-DMA tiles; rounds 1-2: resblock_bwd_full_bf16_kernel)
        if kind == "dgrad" and hw == 16:              # 32-channel blocks @16x16: likewise (wave-specialised 512-thread workgroups)
            return ["resblock_bwd_full32"]
        if kind == "dgrad" and hw == 8:
            return ["resblock_bwd_full32_bf16_kernel<RbFull32T<8"]
        if kind == "fwd":                             # res1 + res2 of a block in one launch
            return [f"resblock_pair_bf16_kernel<RbCfg<{cin}, {hw},"]
        return [f"resblock_bf16_kernel<RbCfg<{cin}, {hw},", ", true>" if kind == "dgrad" else ", false>"]
    if precision == "bf16":
        if cin == 3:
            return ["conv1_wgrad_bf16_kernel<true>"] if kind == "wgrad" else ["conv1_pool_fwd_bf16_kernel"]
        if kind == "wgrad":                           # POOLED variant for a block's first conv (cin != cout or block3.conv; the
            return [f"conv3x3_wgrad_bf16_kernel<WbCfg<{cin}, {cout}, {hw},", ", true>" if cin != cout else ", false>"]      # class is dominated by the res convs)
        if kind == "fwd":                             # the block's first conv, fused with the max pool
            return [f"conv_pool_fwd_bf16_kernel<CpCfg<{cin}, {cout}, {hw}>"]
        if cin == 16 and cout == 32:                  # block2.conv: data + weight gradient from the pooled gradient, dedicated kernel
            return ["block2_conv_bwd_bf16_kernel"]
        return [f"conv3x3_bf16_kernel<BfCfg<{cout}, {cin}, {hw},", "true>, true>" if cin != cout else "true>, false>"]
    if cin == 3:
        return ["conv3x3_wgrad_kernel<WgCfg<3, 4, 16, 64"] if kind == "wgrad" else ["conv3x3_kernel<FwdCfg<3, 4, 16, 64"]
    if kind == "wgrad":
        return [f"conv3x3_wgrad_kernel<WgCfg<{cin}, {cin}, {cout}, {hw},"]
    # FwdCfg<CIN, CINP, COUT, HW, TH, TW, NIMG, IN_U8, TRANSW, BFIO>
    return [f"conv3x3_kernel<FwdCfg<{cin}, {cin}, {cout}, {hw},", "false, false>"] if kind == "fwd" \
        else [f"conv3x3_kernel<FwdCfg<{cout}, {cout}, {cin}, {hw},", "true, false>"]


def pmc_traffic(cls, precision):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc passes (FETCH_SIZE x2 per the
    gfx950 correction + WRITE_SIZE; profiles/r0N_pmc_<precision>.json, collected on 8192-sample launches by
    scratch/pmc_workload.py).  None when no committed counter summary matches."""
    keys = rocprof_name(cls, precision)
    for rnd in ("r03", "r02", "r01"):               # the newest committed counter summary that has this kernel
        try:
            tab = json.load(open(os.path.join(ROOT, "profiles", f"{rnd}_pmc_{precision}.json")))
        except Exception:
            continue
        for name, v in tab.items():
            if all(k in name for k in keys):
                return dict(bytes_per_launch=(v["hbm_read_MB_per_call"] + v["hbm_write_MB_per_call"]) * 1048576.0,
                            read_MB=v["hbm_read_MB_per_call"], write_MB=v["hbm_write_MB_per_call"],
                            source=f"profiles/{rnd}_pmc_{precision}.json", rocprof_kernel=name)
    return None


# Tensor passes the FUSED bf16 kernels really move, as a fraction of the SURVEY 8(d) layer-boundary bytes they are priced at
# (the model charges every conv / pool boundary; fusion keeps those tensors in LDS).  Used for the `own_*` fields only.
OWN_TRAFFIC = {
    "resblock_dgrad": 4.0 / 7.0,          # whole backward of a residual block: dy, a, x in; dx out   vs 2 x 3p + p   (halo rows re-read on top: PMC)
    "resblock_fwd": 5.0 / 10.0,           # res1 + res2: x in; a1, y1, a2, y2 out                    vs 2 x (2 x 2p + p)
    "conv_fwd_3_16_64": 61440.0 / 307200.0,           # frames in, pooled map + arg-max out             vs I + 2X + p
    "conv_wgrad_3_16_64": 61440.0 / 438272.0,         # frames, pooled gradient + arg-max in            vs I + 3X + p
    "conv_fwd_16_32_32": 57344.0 / 180224.0,
    "conv_dgrad_16_32_32": 90112.0 / 278528.0,        # block2.conv data + weight gradient from the pooled gradient
    "conv_fwd_32_32_16": 22528.0 / 53248.0,
}


def own_traffic_ratio(cls, precision):
    if precision != "bf16":
        return 1.0
    for k, v in OWN_TRAFFIC.items():
        if cls.startswith(k):
            return v
    return 1.0


def host_cores():
    """Cores this process may really use: affinity mask capped by the cgroup CPU quota (a 1-GPU box exposes all
    host CPUs but grants a 16-core share); oversubscribing the quota makes the CPU leg slower, not faster."""
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        n = os.cpu_count() or 1
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = max(1, min(n, int(int(q) / int(per))))
    except Exception:
        pass
    return min(n, 16)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--param_name", default="hard-500")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=32, help="rollout steps (of all E envs) the CPU baseline leg runs")
    ap.add_argument("--cpu-threads", type=int, default=0)
    ap.add_argument("--precision", default="bf16", choices=["fp32", "bf16"],
                    help="activation storage: fp32 = parity mode; bf16 = BASELINE config 3 (bf16 storage + bf16 MFMA fwd/dgrad)")
    ap.add_argument("--profile-rollout", action="store_true", help="bracket the rollout-phase launches with HIP events too")
    ap.add_argument("--no-kernel-profile", action="store_true", help="diagnostic: no HIP events around the launches (roofline = null)")
    ap.add_argument("--profile-period", type=int, default=8, help="bracket every P-th minibatch update with HIP events (1 = all)")
    ap.add_argument("--rank-share", type=int, default=1, help="diagnostic: on ONE GPU, run the launch sizes a rank of an R-GPU job sees "
                    "(minibatches of mini_batch_size/R, R-fold gradient accumulation, no collectives)")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true", help="diagnostic: run the N-rank code path (sharding, merged accumulation, "
                    "gradient / statistics all-reduce, barriers) with every rank on GPU 0 and the gloo backend; the number it prints is not a result")
    ap.add_argument("--no-merge", action="store_true", help="diagnostic (with --rank-share R): the R accumulated passes of a minibatch run one by one "
                    "instead of merged into one pass -- what splitting a minibatch into cache-sized pieces would cost / gain")
    ap.add_argument("--debug-flags", type=int, default=0, help="diagnostic: mi_debug_flags bits (1: unfused rollout tail, 4: frames always uploaded by DMA copy, never pulled by a kernel, 16: no side stream in the minibatch pass)")
    ap.add_argument("--no-h2d", action="store_true", help="diagnostic: policy steps read frames already resident in HBM (no per-step upload); "
                    "the default uploads every step's E frames from pinned host memory inside the timed region, as the real loop must")
    ap.add_argument("--engine-only", action="store_true", help="diagnostic: the rollout phase drives mi_rollout_submit / mi_rollout_wait directly with frames "
                    "that already lie in pinned buffers (rounds 1-2's loop) instead of the product's collector PPO._collect on an EnvGroups env")
    ap.add_argument("--no-fp32-record", action="store_true", help="skip the three extra iterations in the fp32 parity mode (the `fp32` sub-record)")
    ap.add_argument("--n-actions", type=int, default=9, help="action count A: 9 = the reference's default for Procgen (ActionWrapper merges the 15 key "
                    "combinations by name, helper_local.py:653), 15 = --no-reduce_duplicate_actions")
    ap.add_argument("--rollout-groups", type=int, default=0, help="env groups of the pipelined rollout (0 = auto: 4 when n_envs >= 128 divides, else 2; "
                    "1 = the reference's serial step: upload, forward, read-back one after the other)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if args.rehearse_on_one_gpu:
        local = 0
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    from mi355.numa import pin_to_gpu_node
    pin_to_gpu_node(local)                                  # host threads + pinned buffers on the GPU's NUMA node (MI355_NUMA_PIN=0: off)
    if world > 1:
        import torch.distributed as dist
        if args.rehearse_on_one_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=device)

    from agents.ppo import PPO
    from common.model import ImpalaModel
    from common.policy import CategoricalPolicy
    from common.storage import Storage

    hp = yaml.safe_load(open(os.path.join(PKG, "hyperparams", "procgen", "config.yml")))[args.param_name]
    T, E, A = hp["n_steps"], hp["n_envs"], args.n_actions
    if args.rank_share > 1:
        hp["mini_batch_size"] //= args.rank_share
    torch.manual_seed(6033)
    model = ImpalaModel(in_channels=3)
    policy = CategoricalPolicy(model, False, A)
    policy.device = device
    storage = Storage((3, 64, 64), model.output_dim, T, E, device)

    class _Log:
        episode_reward_buffer = [0.0]
        logdir = "/tmp"
    agent = PPO(None, policy, _Log(), storage, device, 1, seed=rank, precision=args.precision, **hp)
    if args.no_merge:
        agent.merge_accumulation = False
    if args.debug_flags:
        agent.engine.debug_flags(args.debug_flags)
    eng = agent.engine

    if args.no_h2d:
        args.engine_only = True
    rng = np.random.default_rng(rank)
    rew = eng.pinned((T, E), np.float32)
    done = eng.pinned((T, E), np.float32)
    rew[:] = rng.standard_normal((T, E))
    done[:] = rng.random((T, E)) < 0.01
    stage = eng.pinned((E, 64, 64, 3), np.uint8)
    host_frames = None
    for t in range(T + 1):
        stage[...] = rng.integers(0, 256, size=stage.shape, dtype=np.uint8)
        eng.put_obs(t, stage)
        eng.sync()
    G = args.rollout_groups if args.rollout_groups > 0 else (4 if E >= 128 and E % 4 == 0 else 2)
    if E % G:
        raise SystemExit(f"--rollout-groups {G} does not divide n_envs={E}")
    ng = E // G
    eng.rollout_groups(G)
    env = None
    if not args.engine_only:
        # the PRODUCT's rollout loop: PPO._collect on G synthetic sub-envs behind one VecEnv (common/env/vec_envs.py EnvGroups), each handing
        # out fresh uint8 NHWC frames in ordinary host memory every step, rewards / dones / infos as an env would (agents/ppo.py:225-236:
        # predict, env.step, Storage.store -- staging or in-place page-locking of the frames, note_stored and the info objects included)
        from common.env.vec_envs import EnvGroups, SyntheticTape
        env = EnvGroups([SyntheticTape(ng, A, seed=1000 * rank + g, length=T) for g in range(G)]) if G > 1 else SyntheticTape(E, A, seed=1000 * rank, length=T)
        roll = {"obs": env.reset(), "hidden": np.zeros((E, storage.hidden_state_size), np.float32), "done": np.zeros(E, np.float32)}
    if args.engine_only and not args.no_h2d:
        # what Procgen's rgb buffer would hand over: per env group a small ring of pinned (E/G,64,64,3) uint8 buffers
        host_frames = [[eng.pinned((ng, 64, 64, 3), np.uint8) for _ in range(4)] for _ in range(G)]
        for hg in host_frames:
            for h in hg:
                h[...] = rng.integers(0, 256, size=h.shape, dtype=np.uint8)

    phase = {"rollout_s": 0.0, "update_enqueue_s": 0.0, "estimates_host_s": 0.0, "optimize_head_s": 0.0, "optimize_tail_s": 0.0, "last_enq": 0.0, "first_mb": None, "t_opt0": 0.0}
    # host time inside the update's enqueueing calls (no sync in them): next to the update phase's wall time it says whether a box's
    # host kept the GPU fed (a loaded host shows up here, not in the per-kernel times)
    def _timed(fn):
        def w(*a, **k):
            t = time.perf_counter()
            if phase["first_mb"] is None:                      # optimize() entered -> its first enqueueing call (index permutation, plan)
                phase["first_mb"] = t; phase["optimize_head_s"] += t - phase["t_opt0"]
            try:
                return fn(*a, **k)
            finally:
                phase["last_enq"] = time.perf_counter()
                phase["update_enqueue_s"] += phase["last_enq"] - t
        return w
    eng.minibatch = _timed(eng.minibatch)
    agent.optimizer.step = _timed(agent.optimizer.step)

    def iteration(it):
        t_r = time.perf_counter()
        if env is not None:
            agent._iter = it + 1
            roll["obs"], roll["hidden"], roll["done"] = agent._collect(env, eng, storage, roll["obs"], roll["hidden"], roll["done"])
        # (--engine-only) T policy steps + the bootstrap-value step.  Per env group g the real loop's dependency chain is kept: the frames of step t
        # go up only AFTER the group's actions of step t-1 have reached the host (rollout_wait; env.step(act) would run right there),
        # and with them the reward / done that env.step returned.  Groups are independent chains: while the host waits for group g,
        # the other groups' uploads and forward passes are in flight on their own streams.
        for t in range(T + 1 if env is None else 0):
            for g in range(G):
                if t:
                    eng.rollout_wait(g)
                sl = slice(g * ng, (g + 1) * ng)
                eng.rollout_submit(t, g, None if host_frames is None else host_frames[g][t & 3],
                                   rew[t - 1, sl] if t else None, done[t - 1, sl] if t else None, seed=it)
        for g in range(G if env is None else 0):
            eng.rollout_wait(g)
        phase["rollout_s"] += time.perf_counter() - t_r         # every group's last step has been read back: no extra sync
        t_e = time.perf_counter()
        storage.compute_estimates(hp["gamma"], hp["lmbda"], hp["use_gae"], hp["normalize_adv"], agent.coll)
        phase["estimates_host_s"] += time.perf_counter() - t_e
        phase["t_opt0"] = time.perf_counter(); phase["first_mb"] = None
        out = agent.optimize()
        phase["optimize_tail_s"] += time.perf_counter() - phase["last_enq"]        # last enqueue returned -> optimize() returned (log readback = the sync)
        agent.draw_permutation_ahead()                         # as PPO.train does: the next update's first permutation is drawn behind the next rollout
        return out

    def fence():
        if world > 1:
            torch.distributed.barrier()
        eng.sync()
        torch.cuda.synchronize()

    for it in range(args.warmup):
        iteration(it)
    eng.profile_enable(0 if args.no_kernel_profile else ((2 if args.profile_rollout else 1) | (max(1, args.profile_period) << 8)))
    eng.profile_read(reset=True)
    fence()
    for k in ("rollout_s", "update_enqueue_s", "estimates_host_s", "optimize_head_s", "optimize_tail_s"):
        phase[k] = 0.0
    t0 = time.perf_counter()
    for it in range(args.steps):
        summary = iteration(args.warmup + it)
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=device)
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        dt = float(tt.item())
    prof = eng.profile_read(reset=True)
    eng.profile_enable(0)

    if rank == 0:
        steps_total = world * T * E * args.steps
        value = steps_total / dt
        upd = [r for r in prof if r["phase"] == "update" and r["kernel"].split("_")[0] in ("conv", "resblock")]
        dom = max(upd, key=lambda r: r["ms"]) if upd else None
        roof = None
        if dom is not None:
            sec = dom["ms"] / 1e3
            gbs, tfs = dom["bytes"] / sec / 1e9, dom["flops"] / sec / 1e12
            # matrix peak of the instruction this kernel issues: block1.conv (3 input channels) stays on the fp32 MFMA
            # in both modes; every other conv of the bf16 mode runs v_mfma_f32_16x16x32_bf16
            on_bf16_mfma = args.precision == "bf16"                   # every conv of the bf16 mode issues v_mfma_f32_16x16x32_bf16
            mpeak = MFMA_BF16_PEAK_TF if on_bf16_mfma else MFMA_F32_PEAK_TF
            f_h, f_m = gbs / HBM_PEAK_GBS, tfs / mpeak
            bound = "mfma" if f_m >= f_h else "hbm"
            roof = dict(bound=bound, achieved=(tfs if bound == "mfma" else gbs), peak=(mpeak if bound == "mfma" else HBM_PEAK_GBS),
                        unit=("TFLOP/s" if bound == "mfma" else "GB/s"), frac=(f_m if bound == "mfma" else f_h),
                        traffic=pmc_traffic(dom["kernel"], args.precision),
                        kernel=dom["kernel"], avg_launch_ms=dom["ms"] / dom["launches"], launches=dom["launches"],
                        samples_per_launch=dom["samples"] / dom["launches"], algorithmic_bytes_per_launch=dom["bytes"] / dom["launches"],
                        hbm_GBps=gbs, hbm_frac=f_h, mfma_TFps=tfs, mfma_frac=f_m, mfma_peak_TFps=mpeak,
                        model="algorithmic bytes = SURVEY 8(d) layer-boundary model (a fused kernel can exceed 100 % of it)",
                        own_min_bytes_per_launch=dom["bytes"] / dom["launches"] * own_traffic_ratio(dom["kernel"], args.precision),
                        own_hbm_frac=f_h * own_traffic_ratio(dom["kernel"], args.precision),
                        share_of_timed_region=dom["ms"] * max(1, args.profile_period) / 1e3 / dt,
                        sampled="HIP events bracket every %d-th minibatch update (same launch sizes in all of them)" % max(1, args.profile_period))
        out = {"metric": "env steps/sec (whole node), coinrun hard-500 IMPALA-CNN PPO", "value": value, "unit": "env steps/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": ("bf16" if args.precision == "bf16" else "f32"), "data": "synthetic" + (" -- REHEARSAL: every rank on GPU 0, gloo; not a result" if args.rehearse_on_one_gpu else ""),
               "config": {"workload": f"PPO iteration, {args.param_name}: IMPALA-CNN, T={T}, E={E} envs per GPU, "
                                      f"{hp['epoch']} epochs x {agent.n_minibatch} minibatches of {agent.mini_batch_size} (global), "
                                      f"A={A}" + (" (the reference's default action set)" if A == 9 else " (--no-reduce_duplicate_actions)") + ", "
                                      + (f"rollout = the product's PPO._collect on an EnvGroups of {G} synthetic sub-envs (fresh uint8 frames in ordinary host "
                                         "memory every step, uploaded inside the timed region; Storage bookkeeping and info objects included)" if env is not None
                                         else f"DIAGNOSTIC --engine-only: mi_rollout_submit / wait driven by bench.py, frames in pinned buffers ({G} env groups)" if host_frames is not None
                                         else "DIAGNOSTIC --no-h2d: frames resident in HBM, no per-step upload"),
                          "parallelism": f"dp{world} over n_envs"},
               "roofline": roof,
               # SURVEY 8(d), whole path: env steps/s per GPU x (B_step, F_step) of the layer-boundary model against the HBM / matrix peaks
               "whole_step_roofline": (lambda per_gpu, b_step, f_step, mpeak: {
                   "B_step_MB": b_step / 1e6, "F_step_MFLOP": f_step / 1e6, "hbm_frac": per_gpu * b_step / (HBM_PEAK_GBS * 1e9),
                   "mfma_frac": per_gpu * f_step / (mpeak * 1e12)})(value / world, 8.99e6 if args.precision == "bf16" else 17.90e6, 601.78e6,
                                                                   MFMA_BF16_PEAK_TF if args.precision == "bf16" else MFMA_F32_PEAK_TF),
               "phase_ms_per_step": {"rollout": phase["rollout_s"] / args.steps * 1e3, "update": (dt - phase["rollout_s"]) / args.steps * 1e3,
                                     "update_host_enqueue": phase["update_enqueue_s"] / args.steps * 1e3,
                                     "update_host_estimates": phase["estimates_host_s"] / args.steps * 1e3,
                                     "update_host_before_first_enqueue": phase["optimize_head_s"] / args.steps * 1e3,
                                     "update_after_last_enqueue": phase["optimize_tail_s"] / args.steps * 1e3},
               "kernel_profile_period": (0 if args.no_kernel_profile else max(1, args.profile_period)),    # kernels[]: the bracketed sample only
               "kernels": sorted(prof, key=lambda r: -r["ms"])[:24],
               "loss_total": summary["Loss/total"]}
        if world == 1 and args.precision == "bf16" and not args.no_fp32_record and not args.engine_only and args.rank_share == 1:
            eng.close()
            out["fp32"] = fp32_record(args)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(hp, E, A, args.cpu_sample, args.cpu_threads or host_cores())
        print(json.dumps(out))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
