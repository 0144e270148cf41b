import sys, torch, time, json
sys.path.insert(0,'/root/repo')
from mini_sglang_amd.runtime.qwen3 import Qwen3Config
from mini_sglang_amd.runtime import decode_loop as dl
torch.cuda.set_device(0)
import argparse
ap=argparse.ArgumentParser(); ap.add_argument('--model',default='8b'); ap.add_argument('--bs',type=int,default=256); ap.add_argument('--ctx',type=int,default=2048); ap.add_argument('--ps',type=int,default=16); ap.add_argument('--steps',type=int,default=20); ap.add_argument('--nograph',action='store_true'); ap.add_argument('--layers',type=int,default=0)
a=ap.parse_args()
cfg = {'8b':Qwen3Config.qwen3_8b,'0.6b':Qwen3Config.qwen3_0_6b,'32b':Qwen3Config.qwen3_32b}[a.model]()
if a.model=='0.6b': cfg.tie_word_embeddings=False
if a.layers: cfg.num_hidden_layers=a.layers
t0=time.time()
mr,batch=dl.build_decode_state(cfg,a.bs,a.ctx,a.ps,2*a.steps+32,use_graph=not a.nograph)
torch.cuda.synchronize(); print('setup s',time.time()-t0, 'mem GB', torch.cuda.memory_allocated()/1e9, flush=True)
dt=dl.time_decode(mr,batch,a.steps,5)
print(json.dumps(dict(model=a.model,bs=a.bs,ctx=a.ctx,ps=a.ps,ms_per_step=dt/a.steps*1e3,tok_s=a.bs*a.steps/dt,graph=not a.nograph)))
if not a.nograph:
    dt=dl.time_decode_chained(mr,batch,a.steps,2,chain=10)
    print(json.dumps(dict(model=a.model,bs=a.bs,ctx=a.ctx,ps=a.ps,ms_per_step_chained=dt/a.steps*1e3,tok_s_chained=a.bs*a.steps/dt)))
print([r.token_ids[-4:] for r in batch.reqs[:2]])
