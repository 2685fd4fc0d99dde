"""Rollout time of the training engine + the validation twin (two engines in one process): ms per T = 256 rollout pair at E = 256, bf16,
one after the other (agents/ppo.py:225-252's order) vs as lanes of one host loop (PPO._collect_lanes), with G env groups each."""
import os, sys, time, numpy as np
os.environ.setdefault("GPU_MAX_HW_QUEUES", sys.argv[1] if len(sys.argv) > 1 else "16")
G = int(sys.argv[2]) if len(sys.argv) > 2 else 4
sys.path[:0] = [".", "train-procgen-pytorch_amd"]
import torch, yaml
from agents.ppo import PPO
from common.env.vec_envs import EnvGroups, SyntheticTape
from common.model import ImpalaModel
from common.policy import CategoricalPolicy
from common.storage import Storage
hp = yaml.safe_load(open("train-procgen-pytorch_amd/hyperparams/procgen/config.yml"))["hard-500"]
T, E, A = hp["n_steps"], hp["n_envs"], 9
dev = torch.device("cuda", 0)
torch.manual_seed(1)
policy = CategoricalPolicy(ImpalaModel(3), False, A); policy.device = dev
st, stv = Storage((3, 64, 64), 256, T, E, dev), Storage((3, 64, 64), 256, T, E, dev)
class L: episode_reward_buffer = [0.0]; logdir = "/tmp"
agent = PPO(None, policy, L(), st, dev, 1, storage_valid=stv, precision="bf16", **hp)
mk = lambda s: EnvGroups([SyntheticTape(E // G, A, seed=s + g, length=T) for g in range(G)])
env, envv = mk(0), mk(100)
r = [env.reset(), np.zeros((E, 256), np.float32), np.zeros(E, np.float32)]
rv = [envv.reset(), np.zeros((E, 256), np.float32), np.zeros(E, np.float32)]
for it in range(3):
    agent.engine_valid.copy_params_from(agent.engine)
    t0 = time.perf_counter(); r = list(agent._collect(env, agent.engine, st, *r)); t1 = time.perf_counter()
    rv = list(agent._collect(envv, agent.engine_valid, stv, *rv)); t2 = time.perf_counter()
    (ra, rb) = agent._collect_lanes([(env, agent.engine, st, *r), (envv, agent.engine_valid, stv, *rv)]); t3 = time.perf_counter()
    r, rv = list(ra), list(rb)
    print(f"queues {os.environ['GPU_MAX_HW_QUEUES']} G {G} iteration {it}: train {1e3 * (t1 - t0):.1f} ms + validation {1e3 * (t2 - t1):.1f} ms one after the other; both as lanes of one loop {1e3 * (t3 - t2):.1f} ms")
