import sys, importlib, numpy as np
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
r8b=importlib.import_module('r8brain-free-src_amd')
src,dst,maxin=176400.,44100.,16384
a=r8b.BatchResampler(src,dst,maxin,2.0,180.15,nch=5,device=0)
b=r8b.BatchResampler(src,dst,maxin,2.0,180.15,nch=5,device=0)
for o in (a,b): o.set_option("fuse_hbd",0)
b.set_option("fuse_hbconv",0)
print(a.describe())
rng=np.random.default_rng(3)
for i in range(4):
    x=rng.uniform(-1,1,(5,maxin))
    ya,yb=a.process_host(x),b.process_host(x)
    d=np.abs(ya-yb)
    print(i, ya.shape, "max diff per ch", d.max(axis=1), "first bad idx", [int(np.argmax(d[c]>0)) if d[c].max()>0 else -1 for c in range(5)], "n bad", (d>0).sum(axis=1))
