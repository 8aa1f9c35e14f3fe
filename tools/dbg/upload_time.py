import sys, time
sys.path.insert(0,'/root/repo')
import torch, ptx_import
m=ptx_import.load()
ctx=m.Context(0)
for n in ('pixar_room','teapot','bedroom2_v5','my_cornell_6'):
    s=m.load_scene(n); ctx.upload_scene(s)
    t0=time.time(); ctx.upload_scene(s); print(n, 'T',s.T, 'upload %.0f ms'%((time.time()-t0)*1e3))
