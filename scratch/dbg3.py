import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import ctypes as C
import numpy as np, torch
import libgdf_amd as gdf
from libgdf_amd.columns import column_from_numpy
from util import gen_rand
def mix64(x):
    x = x.astype(np.uint64)
    x ^= x >> np.uint64(33); x *= np.uint64(0xff51afd7ed558ccd)
    x ^= x >> np.uint64(33); x *= np.uint64(0xc4ceb9fe1a85ec53)
    x ^= x >> np.uint64(33)
    return x
lib = gdf._binding._gdf_cdll
np.random.seed(0xabcdef)
n=10000; fb=2
l = gen_rand(np.int32, n, low=0, high=2000)
cl = column_from_numpy(l)
ok=torch.empty(n,dtype=torch.int64,device='cuda'); oi=torch.empty(n,dtype=torch.int32,device='cuda')
off=(C.c_uint32*((1<<fb)+1))(); nj=C.c_uint32(0)
lib.gdf_amd_debug_partition.argtypes=[C.c_void_p,C.c_int,C.c_void_p,C.c_void_p,C.c_void_p,C.c_void_p]
bad=0
for it in range(3000):
    ok.fill_(-7); oi.fill_(-7)
    rc=lib.gdf_amd_debug_partition(C.byref(cl.c), fb, ok.data_ptr(), oi.data_ptr(), off, C.byref(nj))
    assert rc==0
    k=ok.cpu().numpy(); i=oi.cpu().numpy()
    good = (i>=0)&(i<n)
    good2 = good.copy(); good2[good] = (k[good]==l[i[good]].astype(np.uint32).astype(np.int64))
    if not good2.all() or len(set(i.tolist()))!=n:
        bad+=1
        w=np.nonzero(~good2)[0]
        print("iter",it,"bad positions", len(w), "offs", list(off), "vals", k[w[:5]], i[w[:5]], "uniq idx", len(set(i.tolist())))
        runs=np.split(w, np.nonzero(np.diff(w)>1)[0]+1) if len(w) else []
        print(" runs", [(int(r[0]),int(r[-1])) for r in runs][:10])
        cnt=np.bincount(i, minlength=n)
        missing=np.nonzero(cnt==0)[0]; dup=np.nonzero(cnt>1)[0]
        print(" missing rows", len(missing), missing[:8], missing[-3:], " dup rows", len(dup), dup[:8], dup[-3:])
        fine=(mix64(l.astype(np.uint32))>>np.uint64(62)).astype(int)
        print(" bins of missing", np.bincount(fine[missing],minlength=4), "bins of dup", np.bincount(fine[dup],minlength=4))
        pos_of={}
        for p,v in enumerate(i.tolist()): pos_of.setdefault(v,[]).append(p)
        dp=sorted(sum([pos_of[d] for d in dup.tolist()],[]))
        print(" dup positions", dp[:6], dp[-6:], "count", len(dp))
        if bad>=3: break
print("bad",bad)
