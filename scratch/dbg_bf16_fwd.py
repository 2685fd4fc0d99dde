import os, sys, numpy as np, torch, torch.nn.functional as F
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0]=[ROOT, os.path.join(ROOT,'train-procgen-pytorch_amd'), os.path.join(ROOT,'tests')]
from conftest import load_npz, npz_params
from mi355 import layout
from mi355.engine import Engine
from oracle import ppo_oracle as O
A=15
rng=np.random.default_rng(1)
frames=rng.integers(0,256,size=(64,64,64,3),dtype=np.uint8)
P=npz_params(load_npz("g3_impala_forward.npz"))
shapes=layout.impala_param_shapes(A)
flat=layout.flatten(shapes,P)
out={}
for prec in ("fp32","bf16"):
    eng=Engine("impala",2,64,A,64,precision=prec); eng.set_params(flat)
    out[prec]=eng.forward(frames,want_feat=True); 
    # layer-by-layer via op hooks (block1 only): conv -> pool -> res1
    x=eng.op_conv3x3(0,3,16,64,P['embedder.block1.conv.weight'],inp=frames,bias=P['embedder.block1.conv.bias'])
    p=eng.op_maxpool(0,x)
    a1=eng.op_conv3x3(0,16,16,32,P['embedder.block1.res1.conv1.weight'],inp=p,relu_in=True,bias=P['embedder.block1.res1.conv1.bias'])
    p1=eng.op_conv3x3(0,16,16,32,P['embedder.block1.res1.conv2.weight'],inp=a1,relu_in=True,bias=P['embedder.block1.res1.conv2.bias'],res=p)
    out[prec+'_l']=(x,p,a1,p1)
    eng.close()
rl=lambda a,b: float(np.linalg.norm(a-b)/np.linalg.norm(b))
print('feat relL2', rl(out['bf16'][2],out['fp32'][2]), 'logp', rl(out['bf16'][0],out['fp32'][0]), 'value', rl(out['bf16'][1],out['fp32'][1]))
for n,a,b in zip(('conv1','pool1','a1','p1'),out['bf16_l'],out['fp32_l']): print(n,'relL2',rl(a,b),'rms',float(np.sqrt((b**2).mean())))
p={k:torch.from_numpy(np.ascontiguousarray(v)) for k,v in P.items()}
with torch.no_grad():
    taps={}
    feat,_,_=O.impala_embed(p,O.frames_to_obs(frames),taps)
for k in ('embedder.block1','embedder.block2','embedder.block3'): print(k,'out rms',float(taps[k].pow(2).mean().sqrt()),'conv rms',float(taps[k+'_conv'].pow(2).mean().sqrt()))
print('oracle feat vs fp32 engine', rl(out['fp32'][2], feat.numpy()), 'feat rms', float(feat.pow(2).mean().sqrt()), 'frac zero', float((feat==0).float().mean()))
