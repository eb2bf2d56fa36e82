import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as G
from oracle import vbmf_oracle as O
pkg = G.load_package(); capi=pkg.capi
np.set_printoptions(linewidth=220, precision=5)
def relF(a,b): return np.linalg.norm(a-b)/max(np.linalg.norm(b),1e-300)

# (a) standalone sigma2 on 10x20
rng=np.random.default_rng(102)
Y,_,_=O.toy_matrix(10,20,2,0.05,rng); Y=Y.astype(np.float32).astype(np.float64)
po=O.vbmf_init(Y,2,ca=0.1,cb=0.1,sigma2=0.1,rng=np.random.default_rng(103),materialize_yhat=False)
O.updateA(Y,po); O.updateB(Y,po); O.updateCA(po); O.updateCB(po)
with capi.Context(10,20,2,y_dtype=pkg.VBMF_Y_F32) as c:
    c.set_Y(Y)
    c.set_state(po.AHat,po.BHat,po.SigmaA,po.SigmaB,np.diag(po.CA),np.diag(po.CB),po.sigma2)
    c.step(pkg.STEP_SIGMA2); s=c.get_state()
    d=c.dims(); print(d)
    Q=c.peek(capi.PEEK_Q, d["nsplit2"]*d["Hp"]*d["XT2"]*32, dtype=np.float32).reshape(d["nsplit2"],d["Hp"],d["XT2"]*32).sum(0)
    Qref=(Y@po.AHat).T
    print("Q err", relF(Q[:2,:10],Qref), "Q pad max", np.abs(Q[2:]).max(), np.abs(Q[:2,10:]).max())
    O.updateSigma2(Y,po)
    print("sigma2 gpu",s["sigma2"],"ref",po.sigma2, "trYY", c.trYY(), (Y**2).sum())

# (b) cfg2 bf16x2
L,M,H=10000,1000,32
rng=np.random.default_rng(20170103)
A0,B0=rng.standard_normal((M,H)),rng.standard_normal((L,H))
for src in ("host","synth"):
  with capi.Context(L,M,H,y_dtype=pkg.VBMF_Y_BF16) as c:
    print(c.dims())
    if src=="synth": c.set_Y_synthetic(20170101,H,0.05)
    else:
        Yh,_,_=O.toy_matrix(L,M,H,0.05,np.random.default_rng(5)); c.set_Y(Yh)
    Ys=np.ascontiguousarray(c.get_Y())
    print(src,"Y finite",np.isfinite(Ys).all(),"std",Ys.std(),"trYY",c.trYY(),(Ys**2).sum())
    c.set_state(A0,B0,np.zeros((H,H)),np.zeros((H,H)),0.1*np.ones(H),0.1*np.ones(H),0.1)
    po=O.vbmf_parameters(); po.L,po.M,po.H=L,M,H; po.AHat,po.BHat=A0.copy(),B0.copy()
    po.SigmaA=np.zeros((H,H)); po.SigmaB=np.zeros((H,H)); po.CA=0.1*np.eye(H); po.CB=0.1*np.eye(H); po.invCA=10*np.eye(H); po.invCB=10*np.eye(H); po.sigma2=0.1
    c.step(pkg.STEP_A); s=c.get_state(); O.updateA(Ys,po)
    print("  A: SigmaA",relF(s["SigmaA"],po.SigmaA),"AHat",relF(s["AHat"],po.AHat), "finite", np.isfinite(s["AHat"]).all())
    c.step(pkg.STEP_B); s=c.get_state(); O.updateB(Ys,po)
    print("  B: SigmaB",relF(s["SigmaB"],po.SigmaB),"BHat",relF(s["BHat"],po.BHat), "finite", np.isfinite(s["BHat"]).all())
    c.step(pkg.STEP_CA|pkg.STEP_CB); s=c.get_state(); O.updateCA(po); O.updateCB(po)
    print("  C: ca",relF(s["CA_diag"],np.diag(po.CA)),"cb",relF(s["CB_diag"],np.diag(po.CB)))
    c.step(pkg.STEP_SIGMA2); s=c.get_state(); O.updateSigma2(Ys,po)
    print("  s2:",s["sigma2"],po.sigma2)
    it,dd,tr=c.run(2,eps=0.0,est_covs=True,est_var=True,want_trace=True)
    otr=[]; O.vbmf_(Ys,po,2,eps=0.0,est_covs=True,est_var=True,fused=True,trace=otr)
    print("  run2 trace gpu",tr,"\n  oracle",otr)
