import sys, os
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'tests'))
import numpy as np
from helpers import *
from picaso_amd import fluxes
from oracle import oracle as orc
for name in ['phase60','thick','cfg3like']:
    g=Golden(GOLDEN+'/scene1d_%s.npz'%name); nlevel,nwno=g.inp('tau').shape
    rs=np.zeros(nwno)+g.inp('surf_reflect')
    a=(nlevel,g.inp('wno'),nwno,g.geo('numg'),g.geo('numt'),g.inp('tlevel'),g.inp('dtau_og'),g.inp('w0_no_raman'),g.inp('cosb_og'),g.inp('plevel'),g.geo('ubar1'),rs,0,g['dwno'],0)
    flux,lv=fluxes.get_thermal_1d(*a)
    fo,lo=orc.get_thermal_1d(*a)
    ref4=[g['therm1d/hs0_ct0/%s'%nm] for nm in ('fm','fp','fmm','fpm')]
    scale=np.max(np.stack([np.max(np.abs(r),axis=(0,1,2)) for r in ref4]),axis=0)
    for nm,got,ref,o in zip(('fm','fp','fmm','fpm'),lv,ref4,lo):
        e=np.abs(got-ref)/scale; idx=np.unravel_index(np.argmax(e),e.shape)
        eo=np.abs(o-ref)/scale
        print(name,nm,'gpu err %.2e'%e.max(),idx,'got',got[idx],'ref',ref[idx],'scale',scale[idx[-1]],' oracle err %.2e'%eo.max(), 'dtau',g.inp('dtau_og')[min(idx[2],nlevel-2),idx[3]])
