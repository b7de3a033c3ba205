"""Soak test: repeat the FP4 triangle on one panel; for a launch whose cells differ from the popcount kernel's, find
where the wrong rows' contents belong."""
import sys; sys.path.insert(0,'/root/repo')
import numpy as np, torch
from ld_tools_amd import PackedPanel, ld_triangle, synth
n,h,iters=int(sys.argv[1]),int(sys.argv[2]),int(sys.argv[3])
fmt=sys.argv[4] if len(sys.argv)>4 else "k16"
path=sys.argv[5] if len(sys.argv)>5 else "fp4"
n11=len(sys.argv)>6 and sys.argv[6]=="n11"
ignore_poison=len(sys.argv)>6 and sys.argv[6]=="ignorepoison"    # with LDX_ABLATE=2048 (no drain) parked cells stay poison
p=PackedPanel.from_codes(synth.synth_codes_device(n,h,seed=synth.BENCH_SEED))
cw=1 if fmt=="k16" else 2                                     # int32 words per cell
b=ld_triangle(p,fmt=fmt,path="popcount").cells.clone().view(torch.int32).view(-1)
w=torch.randint(1,2**31-1,(128*cw,),device=b.device,dtype=torch.int64)
sig_b=(b.view(-1,128*cw).to(torch.int64)*w).sum(dim=1)          # one signature per 128-cell row
order=torch.argsort(sig_b); sorted_sig=sig_b[order]
r=ld_triangle(p,fmt=fmt,path=path,want_n11=n11)
bad=0
T=(n+127)//128; G64=2*T
def unit_info(v):   # 64-row unit index -> (tile, group64)
    t=0
    base=lambda t: t*G64 - t*(t-1)
    lo,hi=0,T
    while hi-lo>1:
        mid=(lo+hi)//2
        if base(mid)<=v: lo=mid
        else: hi=mid
    return lo, v-base(lo)+2*lo
for it in range(iters):
    r.cells.view(torch.int32).fill_(-1)
    ld_triangle(p,fmt=fmt,path=path,out=r,want_n11=n11)
    a=r.cells.view(torch.int32).view(-1)
    if torch.equal(a,b): continue
    au,bu=a.view(-1,8192*cw),b.view(-1,8192*cw)                 # per 64-row unit (nonzero() on > 2^31 elements overflows)
    neq=(au!=bu)&(au!=-1) if ignore_poison else au!=bu
    ubad=neq.any(dim=1).nonzero().flatten()
    if ubad.numel():
        bad+=1
        units=ubad.cpu().tolist()
        d=neq[ubad].sum()
        print(it,"bad cells",int(d),"units",[(u,)+unit_info(u) for u in units])
        for u in units:
            rows=a[u*8192*cw:(u+1)*8192*cw].view(64,128*cw)
            if (rows==-1).all(): print("   unit",u,"poison"); continue
            sig=(rows.to(torch.int64)*w).sum(dim=1)
            pos=torch.searchsorted(sorted_sig,sig).clamp(max=sorted_sig.numel()-1)
            found=sorted_sig[pos]==sig
            src=order[pos]
            srcu=(src//64)
            print("   unit",u,"rows found elsewhere:",int(found.sum()),"source 64-row units",torch.unique(srcu[found]).cpu().tolist()[:6],
                  [unit_info(int(x)) for x in torch.unique(srcu[found]).cpu().tolist()[:3]])
        if bad>=3: break
print("iterations",it+1,"bad",bad)
