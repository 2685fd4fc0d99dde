import sys, os, numpy as np, torch
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0]=[ROOT, os.path.join(ROOT,'train-procgen-pytorch_amd'), os.path.join(ROOT,'tests')]
from conftest import load_npz, npz_json, npz_params
from oracle import ppo_oracle as O
from mi355 import engine as M, layout
from test_gpu_engine import make_engine, shapes_for, golden_params, load_rollout
from test_oracle_golden import _rollout_from
arch=sys.argv[1] if len(sys.argv)>1 else 'mlp'
z = load_npz(f"g56_{arch}_optimize.npz")
T,E=16,8; A=15 if arch=='impala' else 2
shapes=shapes_for(arch,A)
eng=make_engine(arch,T,E,A,16)
eng.set_params(layout.flatten(shapes, golden_params(arch)))
load_rollout(eng,z,T,E)
eng.compute_estimates(0.999,0.95,True,True)
ro=_rollout_from(z,arch,T,E)
print('adv diff', np.abs(eng.read_field(M.F_ADV)-ro['adv']).max(), 'ret diff', np.abs(eng.read_field(M.F_RET)-ro['ret']).max())
ag=O.OraclePPO(golden_params(arch),arch,T,E,epoch=3,n_minibatch=8,mini_batch_size=16,gamma=0.999,lmbda=0.95,learning_rate=5e-4,grad_clip_norm=0.5)
N=T*E
obs=torch.as_tensor(ro['obs'][:-1],dtype=torch.float32).reshape(N,*ro['obs'].shape[2:])
flat={k: torch.as_tensor(ro[k],dtype=torch.float32).reshape(-1) for k in ('act','logp','ret','adv')}
oldv=torch.as_tensor(ro['val'][:-1],dtype=torch.float32).reshape(-1)
torch.manual_seed(21)
hp=eng.hparams()
step=0
for e in range(3):
    for idx in O.minibatch_indices(N,16):
        ti=torch.as_tensor(idx)
        L,g=ag.loss_and_grads(obs[ti],flat['act'][ti],flat['logp'][ti],oldv[ti],flat['ret'][ti],flat['adv'][ti])
        eng.minibatch(idx,16,hp)
        rec=eng.loss_log()[0]
        mg=layout.unflatten(shapes, eng.get_grads())
        gerr=max(float(np.abs(mg[k]-g[k].numpy()).max()/(np.abs(g[k].numpy()).max()+1e-12)) for k in g)
        nrm,coef=O.clip_grad_norm(g,0.5)
        step+=1
        O.adam_step(ag.p,g,ag.m,ag.v,step,5e-4)
        gn=eng.optimizer_step(5e-4,0.5,step,want_norm=True)
        mp=layout.unflatten(shapes, eng.get_params())
        perr=max(float(np.abs(mp[k]-ag.p[k].numpy()).max()) for k in ag.p)
        print(f"step {step:2d} pi {rec[0]:+.6f}/{L['pi_loss']:+.6f} v {rec[1]:.5f}/{L['value_loss']:.5f} ent {rec[2]:.5f}/{L['entropy']:.5f} gerr {gerr:.2e} gnorm {gn:.5f}/{nrm:.5f} perr {perr:.2e}")
