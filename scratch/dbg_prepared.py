import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import libgdf_amd as gdf
from libgdf_amd import Column
from libgdf_amd.columns import column_from_numpy
print("A", flush=True)
empty = gdf.api.JoinBuild([column_from_numpy(np.zeros(0, dtype=np.int64))])
print("B", flush=True)
li, ri = empty.probe([column_from_numpy(np.arange(5, dtype=np.int64))], how="inner")
print("C", li.numel(), flush=True)
li, ri = empty.probe([column_from_numpy(np.arange(5, dtype=np.int64))], how="left")
print("D", li.cpu().tolist(), ri.cpu().tolist(), flush=True)
jb = gdf.api.JoinBuild([column_from_numpy(np.arange(100, dtype=np.int64))])
try:
    jb.probe([column_from_numpy(np.arange(5, dtype=np.int32))])
except Exception as e:
    print("E", e, flush=True)
os.environ["GDF_JK_SPEC_MIN"] = "1000"
build = torch.randperm(3_000_000, dtype=torch.int64, device="cuda")[:2_000_000]
jb2 = gdf.api.JoinBuild([Column(build)])
print("F", flush=True)
probe = torch.randint(0, 3_000_000, (5_000_000,), dtype=torch.int64, device="cuda")
li, ri = jb2.probe([Column(probe)])
print("G", li.numel(), flush=True)
