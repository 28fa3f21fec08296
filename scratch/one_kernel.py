import sys, torch, os
sys.path.insert(0,'/root/repo')
from preset_gen_vae_amd import ops
B=256
which = sys.argv[1] if len(sys.argv) > 1 else 'down2'
cfg = {'down2': (8,16,4,129,174), 'down3': (16,32,4,65,88), 'down4': (32,64,4,33,45)}[which]
Cb,Cs,k,Hb,Wb = cfg
g=ops.ConvGeom(Cb,Cs,k,2,2,Hb,Wb)
big=torch.randn(B,Cb,Hb,Wb,device='cuda'); w=torch.randn(Cs,Cb,k,k,device='cuda')*0.05
out=torch.empty(B,Cs,g.Hs,g.Ws,device='cuda'); bias=torch.zeros(Cs,device='cuda')
st=torch.empty(2*Cs,device='cuda',dtype=torch.float64)
sc=torch.ones(Cb,device='cuda'); sh=torch.zeros(Cb,device='cuda')
for _ in range(5): ops.conv_down(g,big,w,bias,1,0.1,in_scale=sc,in_shift=sh,stats=st,out=out)
torch.cuda.synchronize()
