import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np, torch
import libgdf_amd as gdf
from libgdf_amd.columns import column_from_numpy
from oracle import oracle
from util import gen_rand
np.random.seed(0xabcdef)
l = gen_rand(np.int32, 10000, low=0, high=2000); r = gen_rand(np.int32, 10000, low=0, high=2000)
for it in range(2):
    li, ri = gdf.api.join([column_from_numpy(l)],[column_from_numpy(r)])
    li=li.cpu().numpy(); ri=ri.cpu().numpy()
    el, er = oracle.join([l],[r])
    print(len(li), len(el))
    got=set(zip(li.tolist(),ri.tolist())); exp=set(zip(el.tolist(),er.tolist()))
    miss=sorted(exp-got); extra=sorted(got-exp)
    print("missing",len(miss),"extra",len(extra), miss[:10])
    if miss:
        mr=np.array([m[1] for m in miss]); ml=np.array([m[0] for m in miss])
        print("missing build rows uniq", len(set(mr)), "probe rows uniq", len(set(ml)))
        ks=sorted(set(r[mr].tolist()))
        print("n missing keys", len(ks), ks[:30])
        # are all matches of those keys missing?
        k0=ks[0]; print("key",k0,"build rows", np.nonzero(r==k0)[0], "missing build rows for it", sorted(set(mr[r[mr]==k0])))
