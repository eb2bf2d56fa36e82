import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as G
pkg = G.load_package()
np.set_printoptions(linewidth=200, precision=4, suppress=True)
for mode in (pkg.VBMF_Y_F32, pkg.VBMF_Y_BF16):
  for (L,M,H) in [(64,64,32),(70,40,5)]:
    with pkg.capi.Context(L,M,H,y_dtype=mode) as c:
        for (l0,m0,h0) in [(0,0,0),(1,0,0),(0,1,0),(0,0,1),(5,9,3),(37,33,4),(63,39,2)]:
            if h0>=H: continue
            Y=np.zeros((L,M)); Y[l0,m0]=1.0
            B=np.zeros((L,H)); B[l0,h0]=1.0
            A=np.zeros((M,H))
            c.set_Y(Y)
            c.set_state(A,B,np.zeros((H,H)),np.zeros((H,H)),np.ones(H),np.ones(H),1e6)
            c.step(pkg.STEP_A); s=c.get_state()
            P=s["AHat"]*1e6
            nz=np.argwhere(np.abs(P)>1e-3)
            print("mode",mode,(L,M,H),"Y[%d,%d] B[%d,%d] -> nonzero P at"%(l0,m0,l0,h0),[(int(i),int(j),round(float(P[i,j]),3)) for i,j in nz][:8])
