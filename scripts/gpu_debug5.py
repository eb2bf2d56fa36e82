import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as G
from oracle import vbmf_oracle as O
pkg = G.load_package(); capi=pkg.capi
np.set_printoptions(linewidth=200, precision=6)
L,M,H=200,120,3
rng=np.random.default_rng(5)
Y,A,B=O.toy_matrix(L,M,H,0.05,rng); Y=(B*np.linspace(1.0,2.5,H))@A.T+0.05*rng.standard_normal((L,M))
Yf=Y.astype(np.float32).astype(np.float64)
po=O.vbmf_sparse_init(Yf,H,ca=0.1,cb=0.1,sigma=0.1,rng=np.random.default_rng(6),full_cov=False,materialize_yhat=False)
hyper=dict(alpha0=po.alpha0,beta0=po.beta0,gamma0=po.gamma0,delta0=po.delta0,eta0=po.eta0,zeta0=po.zeta0)
with capi.Context(L,M,H,y_dtype=pkg.VBMF_Y_F32,variant=capi.VBMF_VARIANT_SPARSE_DIAG) as c:
    c.set_Y(Yf)
    c.sparse_set_state(po.ATVecHat,po.diagSigmaATVec,po.CA,po.beta,po.BHat,po.SigmaB,po.CB,po.delta,po.sigmaHat,po.zeta,hyper)
    tot=0
    for blk in range(60):
        try:
            it,d,tr=c.sparse_run(5,eps=0.0,want_trace=True)
        except Exception as e:
            print("ERROR at sweeps",tot,"..",tot+5,e)
            s=c.sparse_get_state()
            print("CB",s["CB"],"delta",s["delta"],"sigmaHat",s["sigmaHat"],"zeta",s["zeta"])
            print("SigmaB\n",s["SigmaB"]); print("SigmaA",s["SigmaA_diag"])
            print("finite:",{k:bool(np.isfinite(np.asarray(v)).all()) for k,v in s.items()})
            print("CA range",s["CA"].min(),s["CA"].max(),"beta range",s["beta"].min(),s["beta"].max(),"dS range",s["diagSigmaATVec"].min(),s["diagSigmaATVec"].max())
            break
        tot+=it
        O.vbmf_sparse_(Yf,po,5,eps=0.0,full_cov=False)
        s=c.sparse_get_state()
        print(tot,"d",d,"sig gpu",s["sigmaHat"],"ref",po.sigmaHat,"CB gpu",s["CB"],"ref",po.CB)
