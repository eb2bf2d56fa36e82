import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as G
from oracle import vbmf_oracle as O
pkg = G.load_package()
np.set_printoptions(linewidth=200, precision=5)

def relF(a,b): return np.linalg.norm(a-b)/max(np.linalg.norm(b),1e-300)

# 1. bf16 roundtrip
L,M=203,97
rng=np.random.default_rng(7)
Y=rng.standard_normal((L,M))*np.exp(rng.uniform(-3,3,size=(L,1)))
with pkg.capi.Context(L,M,3,y_dtype=pkg.VBMF_Y_BF16) as c:
    c.set_Y(Y); back=c.get_Y()
u=Y.astype(np.float32).view(np.uint32).astype(np.uint64)
u=((u+0x7FFF+((u>>16)&1))>>16)<<16
want=u.astype(np.uint32).view(np.float32).astype(np.float64)
bad=np.argwhere(back!=want)
print("bf16 roundtrip mismatches:",len(bad)); 
for (i,j) in bad[:10]: print(i,j,Y[i,j],back[i,j],want[i,j])

# 2. update A f32
for (L,M,H) in [(10,20,2),(64,64,5),(300,200,5)]:
    rng=np.random.default_rng(1)
    Y,_,_=O.toy_matrix(L,M,H,0.05,rng); Y=Y.astype(np.float32).astype(np.float64)
    po=O.vbmf_init(Y,H,ca=0.1,cb=0.1,sigma2=0.1,rng=np.random.default_rng(2),materialize_yhat=False)
    with pkg.capi.Context(L,M,H,y_dtype=pkg.VBMF_Y_F32) as c:
        c.set_Y(Y)
        c.set_state(po.AHat,po.BHat,po.SigmaA,po.SigmaB,np.diag(po.CA),np.diag(po.CB),po.sigma2)
        s0=c.get_state()
        print(L,M,H,"state roundtrip A",relF(s0["AHat"],po.AHat),"B",relF(s0["BHat"],po.BHat))
        c.step(pkg.STEP_A); s=c.get_state()
        O.updateA(Y,po)
        print("  SigmaA err",relF(s["SigmaA"],po.SigmaA),"AHat err",relF(s["AHat"],po.AHat))
        if relF(s["AHat"],po.AHat)>1e-3:
            print("  gpu A[:4]\n",s["AHat"][:4],"\n  ref A[:4]\n",po.AHat[:4])
            r=s["AHat"]/po.AHat; print("  ratio stats",np.nanmedian(r),np.nanmin(r),np.nanmax(r))
        c.step(pkg.STEP_B); s=c.get_state(); O.updateB(Y,po)
        print("  SigmaB err",relF(s["SigmaB"],po.SigmaB),"BHat err",relF(s["BHat"],po.BHat))
