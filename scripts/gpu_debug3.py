import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as G
G.build()
pkg = G.load_package(); capi=pkg.capi
np.set_printoptions(linewidth=220, precision=4, suppress=True)
L,M,H=64,64,32
with capi.Context(L,M,H,y_dtype=pkg.VBMF_Y_F32) as c:
    d=c.dims(); print(d)
    for l0 in (0,1,2,4,5,8,33):
        Y=np.zeros((L,M)); Y[l0,0]=1.0
        B=np.zeros((L,H)); B[l0,0]=1.0
        c.set_Y(Y); c.set_state(np.zeros((M,H)),B,np.zeros((H,H)),np.zeros((H,H)),np.ones(H),np.ones(H),1e6)
        y1=c.peek(capi.PEEK_Y1, d["XT1"]*d["KS1"]*64*4, dtype=np.float32).reshape(d["XT1"],d["KS1"],64,4)
        fb=c.peek(capi.PEEK_FB, d["KS1"]*d["npart"]*d["NH"]*64*4, dtype=np.float32).reshape(d["KS1"],d["npart"]*d["NH"],64,4)
        print("l0",l0,"Y1 nz:",[tuple(int(v) for v in i) for i in np.argwhere(y1!=0)],"FB nz:",[tuple(int(v) for v in i) for i in np.argwhere(fb!=0)])
        c.step(pkg.STEP_A)
        P=c.peek(capi.PEEK_P, d["nsplit1"]*d["Hp"]*d["XT1"]*32, dtype=np.float32).reshape(d["nsplit1"],d["Hp"],d["XT1"]*32)
        print("    P nz:",[tuple(int(v) for v in i)+(float(P[tuple(i)]),) for i in np.argwhere(P!=0)][:6])
