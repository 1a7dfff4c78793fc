"""gdf_ipc_parser_* (csrc/ipc.cpp): the Arrow IPC hand-off of SURVEY.md 8f rank 4.

Follows the reference's python/tests/test_ipc.py:27-155: pyarrow serialises a schema and a record batch
(int32 / dictionary-encoded string / float64 columns), the schema message is parsed from HOST memory, the record
batch message from DEVICE memory, and the columns are read back through the offsets of the layout JSON.
The schema half needs no GPU; the record-batch half is marked gpu."""
import ctypes as C
import json

import numpy as np
import pytest

pa = pytest.importorskip("pyarrow")


def _expected_values():
    """python/tests/test_ipc.py:27-34"""
    rng = np.random.RandomState(1234)
    names = ["pear", "orange", "grape", "apple"]
    means = [0.26, 0.47, 0.36, 0.69]
    for i in range(30):
        j = rng.choice(range(4))
        yield i, names[j], float(rng.uniform(low=means[j] - 0.25, high=means[j] + 0.25))


def _make_batch():
    indices, names, weights = zip(*_expected_values())
    unique = sorted(set(names))
    d_name = pa.DictionaryArray.from_arrays(pa.array([unique.index(n) for n in names], type=pa.int32()), pa.array(unique))
    return pa.RecordBatch.from_arrays([pa.array(indices, type=pa.int32()), d_name, pa.array(weights)], ["idx", "name", "weight"]), unique


def _lib():
    from libgdf_amd._binding import _gdf_cdll as lib
    lib.gdf_ipc_parser_open.restype = C.c_void_p
    lib.gdf_ipc_parser_open.argtypes = [C.c_char_p, C.c_size_t]
    for name in ("get_schema_json", "get_layout_json", "get_error", "to_json"):
        f = getattr(lib, "gdf_ipc_parser_" + name)
        f.restype, f.argtypes = C.c_char_p, [C.c_void_p]
    lib.gdf_ipc_parser_failed.restype, lib.gdf_ipc_parser_failed.argtypes = C.c_int, [C.c_void_p]
    lib.gdf_ipc_parser_close.restype, lib.gdf_ipc_parser_close.argtypes = None, [C.c_void_p]
    lib.gdf_ipc_parser_open_recordbatches.restype = None
    lib.gdf_ipc_parser_open_recordbatches.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    lib.gdf_ipc_parser_get_data_offset.restype, lib.gdf_ipc_parser_get_data_offset.argtypes = C.c_int64, [C.c_void_p]
    lib.gdf_ipc_parser_get_data.restype, lib.gdf_ipc_parser_get_data.argtypes = C.c_void_p, [C.c_void_p]
    return lib


def _open(lib, schema_bytes):
    h = lib.gdf_ipc_parser_open(schema_bytes, len(schema_bytes))
    assert h
    return h


@pytest.mark.parametrize("legacy_framing", [False, True])
def test_schema_message_from_host_memory(legacy_framing):
    lib = _lib()
    batch, _ = _make_batch()
    sb = batch.schema.serialize().to_pybytes()
    assert sb[:4] == b"\xff\xff\xff\xff"                 # current framing: continuation marker first
    if legacy_framing:
        sb = sb[4:]                                        # the 0.x framing the reference was written against
    h = _open(lib, sb)
    assert not lib.gdf_ipc_parser_failed(h), lib.gdf_ipc_parser_get_error(h)
    js = json.loads(lib.gdf_ipc_parser_get_schema_json(h).decode())
    fields = js["schema"]["fields"]
    assert [f["name"] for f in fields] == ["idx", "name", "weight"]            # test_ipc.py:103-106
    assert fields[0]["type"] == {"name": "int", "bitWidth": 32, "isSigned": True}
    assert fields[2]["type"] == {"name": "floatingpoint", "precision": "DOUBLE"}
    assert fields[1]["type"] == {"name": "utf8"} and fields[1]["dictionary"]["indexType"]["bitWidth"] == 32
    assert [d["id"] for d in js["dictionaries"]] == [fields[1]["dictionary"]["id"]]   # test_ipc.py:108-109
    assert all(f["nullable"] for f in fields) and all(f["children"] == [] for f in fields)
    lib.gdf_ipc_parser_close(h)


def test_schema_type_names_cover_the_relational_dtypes():
    lib = _lib()
    types = [("a", pa.int8()), ("b", pa.int16()), ("c", pa.int64()), ("d", pa.uint32()), ("e", pa.float32()), ("f", pa.date32()),
             ("g", pa.date64()), ("h", pa.timestamp("ms")), ("i", pa.bool_()), ("j", pa.string()),
             ("k", pa.list_(pa.int32())), ("l", pa.struct([("x", pa.int32()), ("y", pa.float64())]))]
    sb = pa.schema(types).serialize().to_pybytes()
    h = _open(lib, sb)
    assert not lib.gdf_ipc_parser_failed(h), lib.gdf_ipc_parser_get_error(h)
    fields = json.loads(lib.gdf_ipc_parser_get_schema_json(h).decode())["schema"]["fields"]
    assert [f["type"]["name"] for f in fields] == ["int", "int", "int", "int", "floatingpoint", "date", "date", "timestamp", "bool",
                                                   "utf8", "list", "struct"]
    assert fields[3]["type"]["isSigned"] is False and fields[5]["type"]["unit"] == "DAY" and fields[6]["type"]["unit"] == "MILLISECOND"
    assert fields[7]["type"]["unit"] == "MILLISECOND"
    assert [c["name"] for c in fields[11]["children"]] == ["x", "y"] and len(fields[10]["children"]) == 1
    lib.gdf_ipc_parser_close(h)


def test_malformed_input_sets_the_failure_flag():
    lib = _lib()
    for junk in (b"", b"\x01\x02", b"\xff\xff\xff\xff\x10\x00\x00\x00" + b"\x00" * 4, b"\x08\x00\x00\x00" + b"\xff" * 8):
        h = _open(lib, junk)
        assert lib.gdf_ipc_parser_failed(h) == 1
        assert lib.gdf_ipc_parser_get_error(h).decode().startswith("ParseError")       # ipc.cu:127-133
        assert json.loads(lib.gdf_ipc_parser_get_layout_json(h).decode()) == []
        lib.gdf_ipc_parser_close(h)
    batch, _ = _make_batch()
    rb = batch.serialize().to_pybytes()
    h = _open(lib, rb)                                       # a record batch is not a schema
    assert lib.gdf_ipc_parser_failed(h) == 1 and b"expecting schema" in lib.gdf_ipc_parser_get_error(h)
    lib.gdf_ipc_parser_close(h)
    sb = batch.schema.serialize().to_pybytes()
    # a crafted vtable offset (soffset = position + 1 / + 2, i.e. a vtable "before" the buffer) must be a
    # ParseError, not a read in front of the heap block
    body = bytearray(sb[8:])                                 # continuation marker + length, then the flatbuffer
    root = int.from_bytes(body[0:4], "little")
    for delta in (1, 2, 1 << 20):
        evil = bytearray(body)
        evil[root:root + 4] = (root + delta).to_bytes(4, "little", signed=True)
        msg = sb[:8] + bytes(evil)
        h = _open(lib, msg)
        assert lib.gdf_ipc_parser_failed(h) == 1 and lib.gdf_ipc_parser_get_error(h).decode().startswith("ParseError")
        lib.gdf_ipc_parser_close(h)
    h = _open(lib, sb)
    lib.gdf_ipc_parser_open(sb, len(sb))                     # unrelated second parser: fine
    assert not lib.gdf_ipc_parser_failed(h)
    lib.gdf_ipc_parser_close(h)


@pytest.mark.gpu
def test_record_batch_in_device_memory(gdf):
    """python/tests/test_ipc.py:50-155 end to end."""
    import torch
    lib = _lib()
    batch, unique = _make_batch()
    sb = batch.schema.serialize().to_pybytes()
    rb = batch.serialize().to_pybytes()
    h = _open(lib, sb)
    assert not lib.gdf_ipc_parser_failed(h)
    dev = torch.frombuffer(bytearray(rb), dtype=torch.uint8).cuda()
    lib.gdf_ipc_parser_open_recordbatches(h, dev.data_ptr(), dev.numel())
    assert not lib.gdf_ipc_parser_failed(h), lib.gdf_ipc_parser_get_error(h)
    layout = json.loads(lib.gdf_ipc_parser_get_layout_json(h).decode())
    assert layout == json.loads(lib.gdf_ipc_parser_to_json(h).decode())
    offset = lib.gdf_ipc_parser_get_data_offset(h)
    assert lib.gdf_ipc_parser_get_data(h) == dev.data_ptr() + offset
    lib.gdf_ipc_parser_close(h)
    body = dev[offset:]
    assert [n["name"] for n in layout] == ["idx", "name", "weight"]
    assert [n["dtype"]["name"] for n in layout] == ["INT32", "DICTIONARY", "DOUBLE"]     # test_ipc.py:115,127,139
    assert [n["dtype"]["bitwidth"] for n in layout] == [32, 32, 64]
    assert all(n["length"] == 30 and n["null_count"] == 0 for n in layout)

    def column(i, dtype):
        b = layout[i]["data_buffer"]
        raw = body[b["offset"]:b["offset"] + b["length"]].cpu().numpy()
        return raw.view(dtype)[:layout[i]["length"]]
    idx, name, weight = column(0, np.int32), column(1, np.int32), column(2, np.float64)
    for (ei, en, ew), gi, gn, gw in zip(_expected_values(), idx, name, weight):
        assert ei == gi and en == unique[gn] and ew == gw
    # the columns can be handed straight to the relational entry points: sum of weight per name
    from libgdf_amd.columns import Column
    keys, agg = gdf.api.group_by("sum", [Column(torch.from_numpy(name.copy()).cuda())], Column(torch.from_numpy(weight.copy()).cuda()))
    exp = {}
    for _, en, ew in _expected_values():
        exp[unique.index(en)] = exp.get(unique.index(en), 0.0) + ew
    got = dict(zip(keys[0].cpu().tolist(), agg.cpu().tolist()))
    assert got.keys() == exp.keys() and all(abs(got[k] - exp[k]) < 1e-12 for k in exp)


@pytest.mark.gpu
def test_record_batch_with_nulls_and_wrong_message(gdf):
    import torch
    lib = _lib()
    arr = pa.array([1, None, 3, None, 5], type=pa.int64())
    batch = pa.RecordBatch.from_arrays([arr], ["v"])
    h = _open(lib, batch.schema.serialize().to_pybytes())
    rb = batch.serialize().to_pybytes()
    dev = torch.frombuffer(bytearray(rb), dtype=torch.uint8).cuda()
    lib.gdf_ipc_parser_open_recordbatches(h, dev.data_ptr(), dev.numel())
    assert not lib.gdf_ipc_parser_failed(h), lib.gdf_ipc_parser_get_error(h)
    node = json.loads(lib.gdf_ipc_parser_get_layout_json(h).decode())[0]
    assert node["null_count"] == 2 and node["dtype"] == {"name": "INT64", "bitwidth": 64}
    body = dev[lib.gdf_ipc_parser_get_data_offset(h):]
    mask = body[node["null_buffer"]["offset"]:node["null_buffer"]["offset"] + node["null_buffer"]["length"]].cpu().numpy()
    assert list(np.unpackbits(mask, bitorder="little")[:5]) == [1, 0, 1, 0, 1]            # LSB-first, 1 = valid: gdf_column's own layout
    lib.gdf_ipc_parser_close(h)
    h = _open(lib, batch.schema.serialize().to_pybytes())
    sb = torch.frombuffer(bytearray(batch.schema.serialize().to_pybytes()), dtype=torch.uint8).cuda()
    lib.gdf_ipc_parser_open_recordbatches(h, sb.data_ptr(), sb.numel())                   # a schema where a batch is expected
    assert lib.gdf_ipc_parser_failed(h) == 1
    lib.gdf_ipc_parser_close(h)
