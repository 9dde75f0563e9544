import sys, os, torch
sys.path.insert(0, os.getcwd())
from dmvsnet_amd import ops
dev="cuda:0"
V,H,W=5,1184,1600
w3=torch.randn(16,32,3,3)*0.05; wl=torch.randn(32,8)*0.3; bl=torch.randn(32)*0.1
layer=ops.ConvLayer("o3",ops.CONV_S1,1,32,16,None,ops.pack_mfma(w3,32,16,ops.CONV_S1,1).to(dev),None,None,False)
layer.w_wino_fpn=ops.pack_wino_fpn(w3,wl,bl).to(dev)
lat=torch.randn(8,V,H,W,device=dev); td=torch.randn(32,V,H//2,W//2,device=dev)
def t(fn,reps=9):
    fn(); torch.cuda.synchronize(); ts=[]
    for _ in range(reps):
        a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    return sorted(ts)[len(ts)//2]
print("fpn2 q4 %.3f ms" % t(lambda: ops.conv3d_fpn(lat,td,wl.to(dev),bl.to(dev),layer,out_q4=True)))
