import sys; sys.path.insert(0,'.')
import torch
from bihome_amd import kernels as K
N,H,Ci,Co,k,s,p=128,32,64,64,3,1,1
d=K.conv_desc(N,H,H,Ci,Co,k,s,p)
x=torch.randn(N,H,H,Ci,device='cuda'); w=torch.randn(Co,k,k,Ci,device='cuda')*0.05
gy=torch.randn(N,d.Ho,d.Wo,Co,device='cuda'); gw=torch.zeros_like(w)
for _ in range(5):
    K.conv_fwd(x,w,None,d); K.conv_wgrad(x,gy,gw,None,d)
torch.cuda.synchronize()
