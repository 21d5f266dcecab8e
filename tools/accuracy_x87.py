"""Accuracy of the single-sweep elimination vs the reference algorithm, both evaluated in x87 extended
precision on a fresh scene (run on the GPU box): shows which of GPU / oracle is closer to the
extended-precision result.  Development aid quoted in DESIGN.md appendix A.1."""
import sys, os
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'tests')); sys.path.insert(0,os.path.join(ROOT,'tools'))
import numpy as np
from helpers import PLANES
from picaso_amd import synthetic as syn, disco, fluxes
from oracle import oracle as orc
import single_sweep_numpy as ss
nlayer, nwno = 47, 1237
sc = syn.make_scene(nlayer, nwno, seed=41)
gang, gw, tang, tw = disco.get_angles_1d(7)
u0, u1, ct, _, _ = disco.compute_disco(7, 1, gang, tang, 0.0)
planes = [sc[k] for k in PLANES]
f0 = np.linspace(0.7, 1.4, nwno)
args = (nlayer + 1, sc["wno"], nwno, 7, 1, *planes, 0.15, u0, u1, 1.0, f0, 3, 0, 1.0, -1.0, 2.0, -0.5, 1.0)
xg, _ = fluxes.get_reflected_1d(*args)
xo, _ = orc.get_reflected_1d(*args)
L=np.longdouble
rs=np.full(nwno,0.15)
xn = ss.reflected_toa(nlayer+1,nwno,planes,rs,u0.ravel(),u1.ravel(),1.0,f0,3,0,1.,-1.,2.,-.5,1.,0,0.0)
xl = ss.reflected_toa(nlayer+1,nwno,[p.astype(L) for p in planes],rs.astype(L),u0.ravel().astype(L),u1.ravel().astype(L),L(1.0),f0.astype(L),3,0,1.,-1.,2.,-.5,1.,0,0.0).astype(float)
xg=xg.reshape(7,nwno); xo=xo.reshape(7,nwno)
def re(a,b): return np.abs(a-b)/np.abs(b)
print('gpu vs oracle', re(xg,xo).max(), ' numpy-sweep vs oracle', re(xn,xo).max())
print('vs longdouble:  gpu %.3e  oracle %.3e  numpy-sweep %.3e' % (re(xg,xl).max(), re(xo,xl).max(), re(xn,xl).max()))
e=re(xg,xo); k,w=np.unravel_index(np.argmax(e),e.shape); print('worst angle',k,'w',w, xg[k,w], xo[k,w], xl[k,w])
# closest approach to singularities in that column
sq3=np.sqrt(3.)
w0=sc['w0'][:,w]; fcg=sc['ftau_cld'][:,w]*sc['cosb'][:,w]
g1=(sq3*.5)*(2-w0*(1+fcg)); g2=(sq3*w0*.5)*(1-fcg); lam=np.sqrt(g1*g1-g2*g2)
print('min |lam^2-1/u0^2|', np.abs(lam*lam-1/u0.ravel()[k]**2).min(), ' min|lam*u1-1|', np.abs(lam*u1.ravel()[k]-1).min())
