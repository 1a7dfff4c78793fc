import sys, os, ctypes as C
sys.path.insert(0,'.')
import torch
import libgdf_amd as gdf
from bench import make_probe_keys, read_profile
from libgdf_amd._binding import rmmOptions_t
from libgdf_amd.columns import Column
gdf.librmm.rmmInitialize(C.byref(rmmOptions_t(1,0,False)))
lib = gdf._binding._gdf_cdll
n = int(sys.argv[1]) if len(sys.argv)>1 else 1000_000_000
fb = 15
keys = make_probe_keys(n, 100_000_000, 0x5EED0002, torch.device('cuda',0))
col = Column(keys)
ok=torch.empty(n,dtype=torch.int64,device='cuda'); oi=torch.empty(n,dtype=torch.int32,device='cuda')
off=(C.c_uint32*((1<<fb)+1))(); nj=C.c_uint32(0); info=(C.c_uint64*2)()
lib.gdf_amd_debug_partition.argtypes=[C.c_void_p,C.c_int,C.c_void_p,C.c_void_p,C.c_void_p,C.c_void_p,C.c_void_p]
for it in range(3):
    if it==1:
        lib.gdf_amd_profile_reset(); lib.gdf_amd_profile_enable(1)
    assert lib.gdf_amd_debug_partition(C.byref(col.c), fb, ok.data_ptr(), oi.data_ptr(), off, C.byref(nj), info)==0
p=read_profile(gdf)
print(os.environ.get('GDF_JK_SDBG','0'), {k:round(v[0]/v[1],3) for k,v in p.items() if v[0]/v[1]>0.05})
