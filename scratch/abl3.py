import sys, torch, os
sys.path.insert(0,'/root/repo')
from preset_gen_vae_amd import ops
def t(fn, it=20):
    fn(); torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); e1.synchronize(); return e0.elapsed_time(e1)/it
B=256
res=[]
for (Cb,Cs,k,Hb,Wb) in [(8,16,4,129,174),(16,32,4,65,88),(32,64,4,33,45)]:
    g=ops.ConvGeom(Cb,Cs,k,2,2,Hb,Wb)
    big=torch.randn(B,Cb,Hb,Wb,device='cuda'); small=torch.randn(B,Cs,g.Hs,g.Ws,device='cuda'); w=torch.randn(Cs,Cb,k,k,device='cuda')*0.05
    out=torch.empty(B,Cs,g.Hs,g.Ws,device='cuda'); outb=torch.empty_like(big); bias=torch.zeros(Cs,device='cuda'); biasb=torch.zeros(Cb,device='cuda')
    st=torch.empty(2*Cs,device='cuda',dtype=torch.float64); stb=torch.empty(2*Cb,device='cuda',dtype=torch.float64)
    sc=torch.ones(Cb,device='cuda'); sh=torch.zeros(Cb,device='cuda'); scs=torch.ones(Cs,device='cuda'); shs=torch.zeros(Cs,device='cuda')
    gw=torch.empty_like(w)
    res.append('%s down %.3f up %.3f wgrad %.3f' % ((Cb,Cs), t(lambda: ops.conv_down(g,big,w,bias,1,0.1,in_scale=sc,in_shift=sh,stats=st,out=out)),
        t(lambda: ops.conv_up(g,small,w,biasb,1,0.1,in_scale=scs,in_shift=shs,stats=stb,out=outb)), t(lambda: ops.conv_wgrad(g,big,small,gw,big_scale=sc,big_shift=sh))))
print(os.environ.get('PGV_LDS_TARGET','default'), ' | '.join(res))
