import time, sys
t0=time.time()
sys.path.insert(0,'.')
import numpy as np, torch
t1=time.time(); print('import torch+numpy %.2f s'%(t1-t0))
from libbsc_amd import GpuContext, api
torch.cuda.init(); x=torch.zeros(1,device='cuda'); torch.cuda.synchronize()
t2=time.time(); print('hip init (torch) %.2f s'%(t2-t1))
n=64<<20
ctx=GpuContext(0,max_n=n+4096)
t3=time.time(); print('bscgpu_create (arena %.1f GiB) %.2f s'%(ctx.arena_bytes/2**30,t3-t2))
for d in (1,2,3):
    t=time.time(); p=ctx.pipe(d); dt=time.time()-t; p.close(); print('pipe depth %d (pinned slots) %.2f s'%(d,dt))
T=api.synth_text_v1(2,n); d_in=torch.from_numpy(T).cuda()
for i in range(3):
    t=time.time(); b=ctx.compress_device(d_in,n,1,1); print('compress_device #%d %.1f ms'%(i,(time.time()-t)*1e3))
