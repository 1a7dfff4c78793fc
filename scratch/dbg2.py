import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np, torch
import libgdf_amd as gdf
from libgdf_amd.columns import column_from_numpy
from oracle import oracle
from util import gen_rand
def mix64(x):
    x = x.astype(np.uint64)
    x ^= x >> np.uint64(33); x *= np.uint64(0xff51afd7ed558ccd)
    x ^= x >> np.uint64(33); x *= np.uint64(0xc4ceb9fe1a85ec53)
    x ^= x >> np.uint64(33)
    return x
np.random.seed(0xabcdef)
dt=np.int32
l = gen_rand(dt, 10000, low=0, high=2000); r = gen_rand(dt, 10000, low=0, high=2000)
el, er = oracle.join([l],[r])
cl, cr = column_from_numpy(l), column_from_numpy(r)
lb = (mix64(l.astype(np.uint32)) >> np.uint64(62)).astype(int)
nfail=0
for it in range(1500):
    li, ri = gdf.api.join([cl],[cr])
    if li.numel()!=len(el):
        nfail+=1
        got=set(zip(li.cpu().numpy().tolist(),ri.cpu().numpy().tolist())); exp=set(zip(el.tolist(),er.tolist()))
        miss=sorted(exp-got)
        ml=np.array(sorted(set(m[0] for m in miss)))
        print("iter",it,"missing pairs", len(miss), "uniq probe rows", len(ml), "min/max row", ml.min(), ml.max())
        print(" bins of missing probe rows", np.bincount(lb[ml], minlength=4), " tile-0(rows<4096) rows per bin", np.bincount(lb[:4096],minlength=4))
        # rank of missing rows within their bin among rows of chunk 0 (tile 0)
        for b in range(4):
            rows_b = np.nonzero(lb[:4096]==b)[0]
            pos = np.searchsorted(rows_b, ml[lb[ml]==b])
            if len(pos): print("  bin",b,"n",len(pos),"positions-in-bin(sorted by row) min/max", pos.min(), pos.max(), "rows", ml[lb[ml]==b][:8])
        if nfail>=3: break
print("fails", nfail)
