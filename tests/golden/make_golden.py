"""Regenerates tests/golden/murmur3_32.json from the REFERENCE's own hash header.

Run in the build container only (needs /root/reference and `make -C oracle ref`):
    python tests/golden/make_golden.py
Inputs: a fixed list of edge values per dtype plus seeded random values.  Expected outputs come from
oracle/_ref/libref_hash.so, i.e. /root/reference/libgdf/src/hashmap/hash_functions.cuh compiled for the
host.  The JSON holds data only (values and their hashes).
"""
import ctypes as C
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.path.join(HERE, "..", "..", "oracle", "_ref", "libref_hash.so")


def main():
    ref = C.CDLL(REF)
    rng = np.random.RandomState(0xabcdef)
    out = {"source": "reference src/hashmap/hash_functions.cuh:30-164 via oracle/ref_hash_shim.cpp", "murmur3_32": {}}
    specs = {
        "i8": (np.int8, [0, 1, -1, 7, 127, -128]),
        "i16": (np.int16, [0, 1, -1, 300, 32767, -32768]),
        "i32": (np.int32, [0, 1, -1, 2147483647, -2147483648, 123456789]),
        "i64": (np.int64, [0, 1, -1, 123456789012345, 9223372036854775807, -9223372036854775808]),
        "f32": (np.float32, [0.0, -0.0, 1.5, -1.5, 3.4028235e38, 1e-45]),
        "f64": (np.float64, [0.0, -0.0, 1.5, -1.5, 1.7976931348623157e308, 5e-324]),
    }
    for name, (dt, fixed) in specs.items():
        if np.dtype(dt).kind == "i":
            info = np.iinfo(dt)
            rnd = rng.randint(info.min, info.max, size=26, dtype=np.int64).astype(dt)
        else:
            rnd = ((rng.random_sample(26) - 0.5) * 2e6).astype(dt)
        vals = np.concatenate([np.array(fixed, dtype=dt), rnd])
        fn = getattr(ref, "ref_murmur_" + name)
        fn.restype = C.c_uint32
        fn.argtypes = [C.c_void_p]
        hashes = [int(fn(vals[i:i + 1].ctypes.data)) for i in range(len(vals))]
        # store the raw little-endian bytes so float values survive JSON exactly
        out["murmur3_32"][name] = [{"bytes": vals[i:i + 1].tobytes().hex(), "hash": hashes[i]} for i in range(len(vals))]
    ref.ref_hash_combine.restype = C.c_uint32
    ref.ref_hash_combine.argtypes = [C.c_uint32, C.c_uint32]
    pairs = [(593689054, 4226891818), (0, 0), (4294967295, 1), (1669671676, 1392991556)]
    pairs += [(int(a), int(b)) for a, b in rng.randint(0, 2**32, size=(12, 2), dtype=np.int64)]
    out["hash_combine"] = [{"l": l, "r": r, "out": int(ref.ref_hash_combine(l, r))} for l, r in pairs]
    ref.ref_identity_i64.restype = C.c_uint32
    ref.ref_identity_i64.argtypes = [C.c_int64]
    ref.ref_identity_i32.restype = C.c_uint32
    ref.ref_identity_i32.argtypes = [C.c_int32]
    out["identity"] = {
        "i64": [{"v": v, "out": int(ref.ref_identity_i64(v))} for v in [0, 1, -1, 0x1FFFFFFFF, -5000000000]],
        "i32": [{"v": v, "out": int(ref.ref_identity_i32(v))} for v in [0, 1, -1, 2147483647, -2147483648]],
    }
    with open(os.path.join(HERE, "murmur3_32.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("wrote murmur3_32.json")


if __name__ == "__main__":
    main()
