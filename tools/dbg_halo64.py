import os, sys
sys.path.insert(0, "/root/repo" if os.path.isdir("/root/repo/michigan_amd") else os.getcwd())
import michigan_amd, torch
from michigan_amd import _cabi, ops
be=_cabi.backend(); g=torch.Generator().manual_seed(3)
def timed(fn, reps=20):
    for _ in range(4): fn()
    torch.cuda.synchronize(); s,e=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True); s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e)/reps*1e3
n=8
x=torch.randn(n,512,512,64,generator=g).to(torch.bfloat16).cuda()
w=(torch.randn(64,64,3,3,generator=g)*0.05).cuda(); wp=ops.pack_weight(w,None,torch.bfloat16,128,64,0); b=torch.randn(64,generator=g).cuda()
out=torch.empty(n,512,512,64,dtype=torch.bfloat16,device="cuda")
fn=lambda: ops._launch_conv(x,wp,out,b,ops.fwd_taps(3,3,1),act=ops.ACT_RELU,Hj=512,Wj=512,isy=1,isx=1,cout=64,cout_gemm=64)
for bits,label in ((0,"full"),(1,"no stores"),(2,"no DMA after tile 0"),(4,"no K loop"),(3,"no stores, no DMA (compute only)"),(6,"no DMA, no K loop (stores only)"),(5,"no stores no K loop (DMA only)"),(7,"nothing")):
    be.mg_set_option(23,bits); print("%-40s %7.1f us"%(label,timed(fn)))
be.mg_set_option(23,0)
t=torch.empty(n*512*512*64,dtype=torch.bfloat16,device="cuda")
print("copy 268 MB -> 268 MB: %.1f us"%timed(lambda: t.copy_(x.view(-1))))
