import os, sys, subprocess, numpy as np
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0]=[ROOT, os.path.join(ROOT,'train-procgen-pytorch_amd'), os.path.join(ROOT,'tests')]
W = r'''
import os, sys, numpy as np, torch, torch.distributed as dist
rank, world, port, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
sys.path[:0] = [sys.argv[5], sys.argv[6]]
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
if world > 1:
    dist.init_process_group("gloo", rank=rank, world_size=world)
from agents.ppo import PPO
from common.model import ImpalaModel
from common.policy import CategoricalPolicy
from common.storage import Storage
from mi355 import engine as M
from mi355.dist import shard_indices
T, EG, A = 4, 8, 15
E = EG // world
torch.manual_seed(6033)
policy = CategoricalPolicy(ImpalaModel(3), False, A)
storage = Storage((3, 64, 64), 256, T, E, torch.device("cuda", 0))
class L: episode_reward_buffer = [0.0]; logdir = "/tmp"
agent = PPO(None, policy, L(), storage, torch.device("cuda", 0), 1, n_steps=T, n_envs=E, epoch=2, n_minibatch=2,
            mini_batch_size=8, gamma=0.999, lmbda=0.95, learning_rate=5e-4, x_entropy_coef=0.02)
rng = np.random.default_rng(0)
frames = rng.integers(0, 256, size=(T + 1, EG, 64, 64, 3), dtype=np.uint8)
act = rng.integers(0, A, (T, EG)); logp = (np.log(1 / A) + 0.2 * rng.standard_normal((T, EG))).astype(np.float32)
val = rng.standard_normal((T + 1, EG)).astype(np.float32); rew = rng.standard_normal((T, EG)).astype(np.float32)
done = (rng.random((T, EG)) < 0.2).astype(np.float32)
sl = slice(rank * E, (rank + 1) * E)
eng = agent.engine
for t in range(T + 1):
    eng.put_obs(t, frames[t, sl]); eng.sync()
eng.write_field(M.F_ACT, act[:, sl].astype(np.float32)); eng.write_field(M.F_LOGP, logp[:, sl]); eng.write_field(M.F_VALUE, val[:, sl])
eng.write_field(M.F_REW, rew[:, sl]); eng.write_field(M.F_DONE, done[:, sl])
storage.compute_estimates(0.999, 0.95, True, True, agent.coll)
torch.manual_seed(5)
hp = agent._hparams(); coll = agent.coll
dump = {}
step = 0
SYNC = os.environ.get("DBG_SYNC", "0") == "1"
cnt = 1
for ep in range(2):
    for chunk in storage.minibatch_index_stream(8, False, EG):
        local = shard_indices(chunk, EG, coll.rank, coll.world)
        eng.minibatch(local, len(chunk), hp)
        if coll.active:
            with torch.cuda.stream(agent._tstream):
                coll.allreduce_sum_(agent._stats_t)
            eng.minibatch_finish()
        if SYNC: eng.sync()
        if cnt % 2 == 0:
            if coll.active:
                if SYNC: dump[f"gl{step}"] = eng.get_grads()
                with torch.cuda.stream(agent._tstream):
                    coll.allreduce_sum_(agent._grads_t)
            if SYNC: dump[f"g{step}"] = eng.get_grads()
            agent.optimizer.step(0.5)
            if SYNC: dump[f"p{step}"] = eng.get_params()
            step += 1
        cnt += 1
dump["pfinal"] = eng.get_params()
dump["log"] = eng.loss_log()
np.savez(out + f".{rank}", **dump)
if world > 1:
    dist.barrier(); dist.destroy_process_group()
'''
open('/tmp/w.py','w').write(W)
PKG=os.path.join(ROOT,'train-procgen-pytorch_amd')
env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
subprocess.run([sys.executable,'/tmp/w.py','0','1','29711','/tmp/one',ROOT,PKG],check=True,env=env)
ps=[subprocess.Popen([sys.executable,'/tmp/w.py',str(r),'2','29712','/tmp/two',ROOT,PKG],env=env) for r in range(2)]
for p in ps: assert p.wait()==0
import numpy as np
res={}
for sync in ("1","0"):
    env2=dict(env, DBG_SYNC=sync)
    subprocess.run([sys.executable,'/tmp/w.py','0','1','29711','/tmp/one'+sync,ROOT,PKG],check=True,env=env2)
    ps=[subprocess.Popen([sys.executable,'/tmp/w.py',str(r),'2','2971'+str(2+int(sync)),'/tmp/two'+sync,ROOT,PKG],env=env2) for r in range(2)]
    for p in ps: assert p.wait()==0
    res[sync]=(np.load('/tmp/one'+sync+'.0.npz'), np.load('/tmp/two'+sync+'.0.npz'), np.load('/tmp/two'+sync+'.1.npz'))
for sync,(a,b0,b1) in res.items():
    print('SYNC',sync,'final param diff 1-rank vs 2-rank', np.abs(a['pfinal']-b0['pfinal']).max(), ' rank0 vs rank1', np.abs(b0['pfinal']-b1['pfinal']).max())
print('1-rank sync vs nosync', np.abs(res['1'][0]['pfinal']-res['0'][0]['pfinal']).max())
print('2-rank sync vs nosync', np.abs(res['1'][1]['pfinal']-res['0'][1]['pfinal']).max())
a,b0,b1=res['1']
for s_ in range(4):
    g1=a[f'g{s_}']; g2=b0[f'g{s_}']; gs=b0[f'gl{s_}']+b1[f'gl{s_}']
    print(f"step {s_} |g1|={np.linalg.norm(g1):.4f} max|g1-g2|={np.abs(g1-g2).max():.3e} max|g2-(gl0+gl1)|={np.abs(g2-gs).max():.3e} param diff {np.abs(a[f'p{s_}']-b0[f'p{s_}']).max():.3e}")
