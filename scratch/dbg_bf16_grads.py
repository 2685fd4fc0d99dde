import os, sys, numpy as np, torch
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0]=[ROOT, os.path.join(ROOT,'train-procgen-pytorch_amd'), os.path.join(ROOT,'tests')]
from conftest import load_npz, npz_params
from mi355 import engine as M, layout
from mi355.engine import Engine
T,E,A=16,64,15
rng=np.random.default_rng(1)
frames=rng.integers(0,256,size=(T+1,E,64,64,3),dtype=np.uint8)
shapes=layout.impala_param_shapes(A)
flat=layout.flatten(shapes, npz_params(load_npz("g3_impala_forward.npz")))
act=rng.integers(0,A,(T,E)); logp=(np.log(1/A)+0.3*rng.standard_normal((T,E))).astype(np.float32)
val=rng.standard_normal((T+1,E)).astype(np.float32)*0.5; rew=rng.standard_normal((T,E)).astype(np.float32); done=(rng.random((T,E))<0.05).astype(np.float32)
res={}
for B in (32, 512):
  idx=rng.permutation(T*E)[:B]
  for prec in ("fp32","bf16"):
    eng=Engine("impala",T,E,A,B,precision=prec); eng.set_params(flat)
    for t in range(T+1): eng.put_obs(t,frames[t]); eng.sync()
    eng.write_field(M.F_ACT,act.astype(np.float32)); eng.write_field(M.F_LOGP,logp); eng.write_field(M.F_VALUE,val); eng.write_field(M.F_REW,rew); eng.write_field(M.F_DONE,done)
    eng.compute_estimates(0.999,0.95,True,True)
    eng.minibatch(idx,B,eng.hparams())
    res[(B,prec)]=(eng.loss_log()[0], layout.unflatten(shapes, eng.get_grads()))
    eng.close()
  l32,g32=res[(B,"fp32")]; l16,g16=res[(B,"bf16")]
  print("B",B,"losses fp32",l32[:5],"bf16",l16[:5])
  for k in g32:
    a=g32[k].astype(np.float64).ravel(); b=g16[k].astype(np.float64).ravel()
    print(f"  {k:38s} rel {np.linalg.norm(a-b)/np.linalg.norm(a):.3f} cos {a@b/np.linalg.norm(a)/np.linalg.norm(b):.4f}")
