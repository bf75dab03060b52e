import sys, importlib, os
sys.path[:0]=['/root/repo','/root/repo/tests']
import numpy as np, scenes, oracle_lib as O
import test_gpu_render as T
g=importlib.import_module('4dgaussiansplatrendering_amd')
n,W,H=60000,960,540
ctx=g.Context(W,H)
pos4,q,scale,life,fade,vel,rgba=scenes.cube_params_4d(n)
rec=g.build_records_4d(pos4,q,scale*4.0,life,fade,vel,rgba)
for t,mo in ((0.0, 0.0), (12.5, 0.0), (25.0, 0.05), (50.0, 0.0)):
    img,projd,st,(view,proj)=T.gpu_frame(ctx,g,rec,scenes.CAM_CUBE,t=t,min_opacity=mo)
    eimg,eperm,_=O.render_4d(rec,True,t,mo,scenes.CAM_CUBE[0],view,proj,W,H)
    d=np.abs(img.astype(np.float64)-eimg).max(axis=2)
    bad=np.argwhere(d>1e-4)
    print('t',t,'linf',d.max(),'nbad',len(bad), bad[:5], 'entries',st)
