import sys, time, os
sys.path.insert(0,'.')
import numpy as np, torch
from libbsc_amd import GpuContext, api
n=64<<20
T=api.synth_text_v1(2,n); d=torch.from_numpy(T).cuda(); out=torch.empty_like(d)
ctx=GpuContext(0,max_n=n+4096)
r=1<<((n//8).bit_length()-1)
ctx.bwt_device(d,out,n,aux_rate=r)
c0=time.process_time(); w0=time.time()
for i in range(40): ctx.bwt_device(d,out,n,aux_rate=r)
c1=time.process_time(); w1=time.time()
print('bwt_device x40: wall %.3f s, process CPU %.3f s -> %.0f %% of one CPU'%(w1-w0,c1-c0,100*(c1-c0)/(w1-w0)))
c0=time.process_time(); time.sleep(1.0); print('idle 1 s: process CPU %.3f s'%(time.process_time()-c0))
os.environ['BSCGPU_HOST_THREADS']='1'
c0=time.process_time(); w0=time.time()
for i in range(10): ctx.compress_device(d,n,1,3)
c1=time.process_time(); w1=time.time()
print('compress_device(-e0) x10: wall %.3f s, process CPU %.3f s'%(w1-w0,c1-c0))
