import sys, importlib, os, time
sys.path[:0]=['/root/repo','/root/repo/tests', os.getcwd(), os.path.join(os.getcwd(),'tests')]
import numpy as np, scenes
g=importlib.import_module('4dgaussiansplatrendering_amd')
n=int(sys.argv[1]); sync=int(sys.argv[2]); steps=int(sys.argv[3])
W,H=1920,1080
pos,q,scale,rgba=scenes.cube_params(n); rec=g.build_records_3d(pos,q,scale,rgba)
ctx=g.Context(W,H); data=ctx.buffer(rec)
kb=[(ctx.buffer(nbytes=4*n),ctx.buffer(nbytes=4*n)) for _ in range(2)]
view=g.look_at(*scenes.CAM_CUBE); proj=g.perspective(60.0,W,H,0.1,5000.0)
ctx.set_clear_color(g.CLEAR_COLOR); ctx.set_mode(g.MODE_4D_SORTED); ctx.bind(2,data); ctx.set_uniforms(time=0.0,min_opacity=0.0,view=view,proj=proj)
t0=time.time()
try:
    for k in range(steps):
        keys,idx=kb[k&1]
        ctx.clear(); ctx.keygen(data,0.0,scenes.CAM_CUBE[0],keys,idx,n); ctx.sort_pairs(keys,idx,n); ctx.bind(1,idx); ctx.draw_instanced(n)
        if sync: ctx.finish()
    ctx.finish()
    print('ok n',n,'sync',sync,'steps',steps,'ms/frame',1e3*(time.time()-t0)/steps, ctx.stats())
except Exception as e:
    print('FAILED at step',k,'after',time.time()-t0,'s:',e)
