"""No-GPU checks of the drop-in boundary: the libraries load, export every symbol include/gdf/gdf.h and
include/memory.h declare, and the struct layouts are the reference's."""
import ctypes as C
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, "libgdf_amd", "lib")


def _declared(header, extra_args=()):
    src = subprocess.check_output(["gcc", "-E", "-P", "-I", os.path.join(ROOT, "include"), *extra_args, header]).decode()
    src = re.sub(r"\s+", " ", src)
    names = set(re.findall(r"\b([A-Za-z_][A-Za-z0-9_]*) ?\([^()]*\) ?;", src))
    return {n for n in names if not n.startswith("__")}


@pytest.fixture(scope="module")
def libs():
    # (the test-hook library first: libgdf.so's weak reference to its lookup is bound when libgdf.so is loaded)
    C.CDLL(os.path.join(LIBDIR, "libgdf_testhook.so"), mode=C.RTLD_GLOBAL)
    rmm = C.CDLL(os.path.join(LIBDIR, "librmm.so"), mode=C.RTLD_GLOBAL)
    gdf = C.CDLL(os.path.join(LIBDIR, "libgdf.so"), mode=C.RTLD_GLOBAL)
    return gdf, rmm


@pytest.fixture(scope="module")
def hook():
    lib = C.CDLL(os.path.join(LIBDIR, "libgdf_testhook.so"), mode=C.RTLD_GLOBAL)
    lib.gdf_amd_debug_force.restype = C.c_int
    lib.gdf_amd_debug_force.argtypes = [C.c_char_p, C.c_char_p]
    lib.gdf_amd_testhook_forced.restype = C.c_char_p
    lib.gdf_amd_testhook_forced.argtypes = [C.c_char_p]
    return lib


def test_libgdf_exports_every_declared_symbol(libs):
    gdf, _ = libs
    names = _declared(os.path.join(ROOT, "include", "gdf", "gdf.h"))
    assert len(names) > 280, len(names)          # functions.h declares ~283 + 2 io entry points
    missing = [n for n in sorted(names) if not hasattr(gdf, n)]
    assert not missing, missing


def test_libgdf_exports_every_extension_symbol(libs):
    """include/gdf/gdf_amd_ext.h: the prepared-build / accumulated-probe handles, the shuffle partitions and the fused
    multi-GPU join's four entry points -- what libgdf_amd/multigpu.py and a host in another language bind."""
    gdf, _ = libs
    names = _declared(os.path.join(ROOT, "include", "gdf", "gdf_amd_ext.h")) - _declared(os.path.join(ROOT, "include", "gdf", "gdf.h"))
    assert {"gdf_amd_fj_plan", "gdf_amd_fj_send", "gdf_amd_fj_build_create", "gdf_amd_fj_probe_add", "gdf_amd_join_build_create",
            "gdf_amd_join_probe_begin", "gdf_amd_shuffle_partition_stable", "gdf_amd_narrow_keys"} <= names, sorted(names)
    missing = [n for n in sorted(names) if not hasattr(gdf, n)]
    assert not missing, missing
    assert hasattr(gdf, "gdf_amd_profile_read")           # (its array-pointer argument escapes the declaration regex)


def test_librmm_exports_every_declared_symbol(libs):
    _, rmm = libs
    names = _declared(os.path.join(ROOT, "include", "memory.h"))
    assert {"rmmInitialize", "rmmFinalize", "rmmAlloc", "rmmRealloc", "rmmFree", "rmmGetInfo", "rmmGetAllocationOffset",
            "rmmWriteLog", "rmmLogSize", "rmmGetLog", "rmmGetErrorString"} <= names
    missing = [n for n in sorted(names) if not hasattr(rmm, n)]
    assert not missing, missing


def test_reference_function_list_is_covered(libs):
    """Every function name the reference's cffi headers declare is exported (only when /root/reference is present)."""
    gdf, _ = libs
    ref = "/root/reference/libgdf/include/gdf/cffi"
    if not os.path.isdir(ref):
        pytest.skip("reference tree not present on this machine")
    text = open(os.path.join(ref, "functions.h")).read() + open(os.path.join(ref, "io_functions.h")).read()
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    text = re.sub(r"//[^\n]*", " ", text)
    text = re.sub(r"\s+", " ", text)
    names = set(re.findall(r"\b((?:gdf|gpu)_[A-Za-z0-9_]+|read_csv|get_column_byte_width) ?\(", text))
    assert len(names) > 280
    missing = [n for n in sorted(names) if not hasattr(gdf, n)]
    assert not missing, missing


_BASE_TYPES = {"void", "char", "short", "int", "long", "float", "double", "signed", "unsigned", "const", "struct", "enum", "_Bool", "bool",
               "size_t", "int8_t", "int16_t", "int32_t", "int64_t", "uint8_t", "uint16_t", "uint32_t", "uint64_t"}


def _prototypes(header, include_dirs):
    """{function name: (return type, (parameter types...))} of every prototype a header declares after `gcc -E` -- parameter NAMES
    dropped, array parameters decayed to pointers, `()` and `(void)` alike, qualifiers kept."""
    args = ["gcc", "-E", "-P", "-x", "c"]
    for d in include_dirs:
        args += ["-I", d]
    src = subprocess.check_output(args + [header]).decode()
    src = re.sub(r"__attribute__ ?\(\(.*?\)\)", " ", src)
    src = re.sub(r"#pragma[^\n]*", " ", src)
    src = re.sub(r"\s+", " ", src)
    types = set(_BASE_TYPES)
    types |= set(re.findall(r"\btypedef [^;{}]*?\b([A-Za-z_]\w*) ?;", src))                # typedef T name;
    types |= set(re.findall(r"\} ?([A-Za-z_]\w*) ?;", src))                               # typedef struct / enum { ... } name;
    types |= set(re.findall(r"\b(?:struct|enum) ([A-Za-z_]\w*)", src))

    def norm(decl, is_param):
        decl = decl.strip()
        ptr_extra = decl.count("[")
        decl = re.sub(r"\[[^\]]*\]", " ", decl)
        toks = re.findall(r"[A-Za-z_]\w*|\*", decl)
        if is_param and len(toks) >= 2 and toks[-1] != "*" and toks[-1] not in types:
            toks = toks[:-1]                                                               # the parameter's name
        return " ".join(toks + ["*"] * ptr_extra)
    out = {}
    for m in re.finditer(r"([A-Za-z_][\w \*]*?[ \*])([A-Za-z_]\w*) ?\(([^()]*)\) ?;", src):
        ret, name, params = m.group(1), m.group(2), m.group(3)
        if name.startswith("__") or "typedef" in ret:
            continue
        ps = [norm(q, True) for q in params.split(",")] if params.strip() not in ("", "void") else []
        out[name] = (norm(ret, False), tuple(ps))
    return out


def test_reference_prototypes_match_type_for_type():
    """not only the NAMES -- return type and every parameter type of every function the reference's cffi headers
    declare (include/gdf/cffi/functions.h:1-785, io_functions.h, with types.h / io_types.h / convert_types.h for the typedefs) must be
    what include/gdf/gdf.h declares.  Both sides go through `gcc -E`; skipped where the reference tree is absent (the GPU box)."""
    ref_inc = "/root/reference/libgdf/include"
    if not os.path.isdir(os.path.join(ref_inc, "gdf", "cffi")):
        pytest.skip("reference tree not present on this machine")
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        # the reference's gdf.h is C++; its five cffi headers are plain C, read in the order libgdf_cffi/libgdf_build.py:6 cdefs them
        tu = os.path.join(d, "ref.h")
        open(tu, "w").write("#include <stddef.h>\n#include <stdint.h>\n#include <stdbool.h>\n" + "".join(
            f'#include "gdf/cffi/{h}"\n' for h in ("types.h", "convert_types.h", "functions.h", "io_types.h", "io_functions.h")))
        ref = _prototypes(tu, [ref_inc])
    ours = _prototypes(os.path.join(ROOT, "include", "gdf", "gdf.h"), [os.path.join(ROOT, "include")])
    ref = {k: v for k, v in ref.items() if re.match(r"(gdf|gpu)_|read_csv$|get_column_byte_width$", k)}
    assert len(ref) > 280, len(ref)
    missing = sorted(set(ref) - set(ours))
    assert not missing, missing
    different = {k: (ref[k], ours[k]) for k in ref if ref[k] != ours[k]}
    assert not different, different


def test_reference_struct_and_enum_definitions_match():
    """gdf_column / gdf_context member types and order, and every enumerator's VALUE (types.h:15-195), against the reference's own
    headers: both compiled by gcc into a table of offsets / sizes / values and compared."""
    ref_inc = "/root/reference/libgdf/include"
    if not os.path.isdir(os.path.join(ref_inc, "gdf", "cffi")):
        pytest.skip("reference tree not present on this machine")
    import tempfile
    prog = r"""
#include <stdio.h>
#include <stddef.h>
#include <stdint.h>
#include <stdbool.h>
#ifdef REF
#include "gdf/cffi/types.h"
#include "gdf/cffi/convert_types.h"
#include "gdf/cffi/functions.h"
#include "gdf/cffi/io_types.h"
#include "gdf/cffi/io_functions.h"
#else
#include <gdf/gdf.h>
#endif
int main(void) {
  printf("column %zu %zu %zu %zu %zu %zu %zu %zu\n", sizeof(gdf_column), offsetof(gdf_column, data), offsetof(gdf_column, valid), offsetof(gdf_column, size),
         offsetof(gdf_column, dtype), offsetof(gdf_column, null_count), offsetof(gdf_column, dtype_info), offsetof(gdf_column, col_name));
  printf("context %zu %zu %zu %zu %zu %zu\n", sizeof(gdf_context), offsetof(gdf_context, flag_sorted), offsetof(gdf_context, flag_method),
         offsetof(gdf_context, flag_distinct), offsetof(gdf_context, flag_sort_result), offsetof(gdf_context, flag_sort_inplace));
  printf("dtype %d %d %d %d %d %d %d %d %d %d %d %d %d\n", GDF_invalid, GDF_INT8, GDF_INT16, GDF_INT32, GDF_INT64, GDF_FLOAT32, GDF_FLOAT64, GDF_DATE32,
         GDF_DATE64, GDF_TIMESTAMP, GDF_CATEGORY, GDF_STRING, N_GDF_TYPES);
  printf("error %d %d %d %d %d %d %d %d\n", GDF_SUCCESS, GDF_CUDA_ERROR, GDF_UNSUPPORTED_DTYPE, GDF_COLUMN_SIZE_MISMATCH, GDF_COLUMN_SIZE_TOO_BIG,
         GDF_DATASET_EMPTY, GDF_VALIDITY_UNSUPPORTED, GDF_UNSUPPORTED_METHOD);
  printf("misc %d %d %d %d %d %d %d %d %d %d\n", GDF_HASH_MURMUR3, GDF_HASH_IDENTITY, GDF_SORT, GDF_HASH, GDF_SUM, GDF_MIN, GDF_MAX, GDF_AVG, GDF_COUNT,
         GDF_COUNT_DISTINCT);
  printf("cmp %d %d %d %d %d %d %zu %zu\n", GDF_EQUALS, GDF_NOT_EQUALS, GDF_LESS_THAN, GDF_LESS_THAN_OR_EQUALS, GDF_GREATER_THAN, GDF_GREATER_THAN_OR_EQUALS,
         sizeof(gdf_size_type), sizeof(gdf_dtype_extra_info));
  return 0;
}
"""
    outs = []
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "t.c"), "w").write(prog)
        for inc in (ref_inc, os.path.join(ROOT, "include")):
            exe = os.path.join(d, "t")
            subprocess.check_call(["gcc", "-I", inc, *(["-DREF"] if inc == ref_inc else []), os.path.join(d, "t.c"), "-o", exe])
            outs.append(subprocess.check_output([exe]).decode())
    assert outs[0] == outs[1], outs


def test_struct_layout(libs):
    gdf, _ = libs
    from libgdf_amd._binding import gdf_column, gdf_context
    gdf.gdf_column_sizeof.restype = C.c_size_t
    assert gdf.gdf_column_sizeof() == 56 == C.sizeof(gdf_column)
    assert C.sizeof(gdf_context) == 20
    offs = {f: getattr(gdf_column, f).offset for f in ("data", "valid", "size", "dtype", "null_count", "dtype_info", "col_name")}
    assert offs == dict(data=0, valid=8, size=16, dtype=24, null_count=32, dtype_info=40, col_name=48)


def test_error_names_and_views(libs):
    gdf, _ = libs
    from libgdf_amd._binding import gdf_column, gdf_context
    gdf.gdf_error_get_name.restype = C.c_char_p
    assert gdf.gdf_error_get_name(0) == b"GDF_SUCCESS"
    assert gdf.gdf_error_get_name(4) == b"GDF_COLUMN_SIZE_TOO_BIG"
    assert gdf.gdf_error_get_name(7) == b"GDF_VALIDITY_UNSUPPORTED"
    assert gdf.gdf_error_get_name(16) == b"GDF_HASH_TABLE_INSERT_FAILURE"
    assert gdf.gdf_error_get_name(22) == b"GDF_NULL_NVTX_NAME"
    assert gdf.gdf_error_get_name(23) == b"Internal error. Unknown error code."
    col = gdf_column()
    col.null_count = 9
    gdf.gdf_column_view.argtypes = [C.POINTER(gdf_column), C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    assert gdf.gdf_column_view(C.byref(col), 0x1000, None, 5, 4) == 0
    assert (col.data, col.valid, col.size, col.dtype, col.null_count) == (0x1000, None, 5, 4, 0)
    ctx = gdf_context()
    assert gdf.gdf_context_view(C.byref(ctx), 0, 1, 0, 1, 0) == 0
    assert (ctx.flag_method, ctx.flag_sort_result) == (1, 1)
    w = C.c_int(0)
    assert gdf.get_column_byte_width(C.byref(col), C.byref(w)) == 0 and w.value == 8
    col.dtype = 11   # GDF_STRING
    assert gdf.get_column_byte_width(C.byref(col), C.byref(w)) == 2 and w.value == -1


def test_out_of_scope_entry_points_report_unsupported(libs):
    gdf, _ = libs
    assert gdf.gdf_sin_f32(None, None) == 12          # GDF_UNSUPPORTED_METHOD
    assert gdf.gdf_add_i32(None, None, None) == 12
    assert gdf.gdf_cast_i32_to_f64(None, None) == 12


def test_host_side_argument_errors_need_no_gpu(libs):
    """Validation paths that return before any device work (joining.cu:495-511, sqls_ops.cu:1095-1106,
    hashing.cu:573-585, scan.cu:55-58)."""
    gdf, _ = libs
    from libgdf_amd._binding import gdf_column, gdf_context
    assert gdf.gdf_inner_join(None, 0, None, None, 0, None, 1, 0, None, None, None, None) == 5      # GDF_DATASET_EMPTY
    assert gdf.gdf_group_by_sum(0, None, None, None, None, None, None) == 5
    assert gdf.gdf_hash_partition(0, None, None, 0, 0, None, None, 0) == 8                            # GDF_INVALID_API_CALL
    assert gdf.gdf_hash(0, None, 0, None) == 5
    assert gdf.gdf_order_by(0, None, 0, None, None, None) == 5
    a, b = gdf_column(), gdf_column()
    a.size, b.size, a.dtype, b.dtype = 4, 5, 3, 3
    assert gdf.gdf_prefixsum_i32(C.byref(a), C.byref(b), 1) == 3                                      # GDF_COLUMN_SIZE_MISMATCH
    b.size, b.dtype = 4, 4
    assert gdf.gdf_prefixsum_i32(C.byref(a), C.byref(b), 1) == 2                                      # GDF_UNSUPPORTED_DTYPE
    b.dtype, a.valid = 3, 0x10
    assert gdf.gdf_prefixsum_i32(C.byref(a), C.byref(b), 1) == 7                                      # GDF_VALIDITY_UNSUPPORTED
    # the SORT group-by rejects valid masks before touching the device (sqls_ops.cu:1103-1106); the HASH method
    # accepts them (BASELINE config C5), see tests/test_gpu_groupby.py
    key, agg, out = gdf_column(), gdf_column(), gdf_column()
    key.size = agg.size = 3
    key.valid = 0x10
    ctx = gdf_context()
    ctx.flag_method = 0
    keys = (C.POINTER(gdf_column) * 1)(C.pointer(key))
    assert gdf.gdf_group_by_sum(1, keys, C.byref(agg), None, keys, C.byref(out), C.byref(ctx)) == 7
    # joins: INT_MAX rows is refused (tests/join/join-tests.cu:750-760)
    big_l, big_r, ol, orr = gdf_column(), gdf_column(), gdf_column(), gdf_column()
    big_l.size = 2**31 - 1
    big_r.size = 10
    L = (C.POINTER(gdf_column) * 1)(C.pointer(big_l))
    R = (C.POINTER(gdf_column) * 1)(C.pointer(big_r))
    idx = (C.c_int * 1)(0)
    assert gdf.gdf_inner_join(L, 1, idx, R, 1, idx, 1, 0, None, C.byref(ol), C.byref(orr), C.byref(ctx)) == 4


def test_shipped_library_reads_no_environment(libs, hook):
    """csrc/lab.h: the shipped libgdf.so / librmm.so import no getenv at all -- a stray GDF_* variable in a caller's
    environment cannot change which algorithm runs -- and (round 6) libgdf.so exports no path switch either:
    gdf_amd_debug_force lives in libgdf_testhook.so (csrc/testhook.cpp, test infrastructure), libgdf.so only has a WEAK reference
    to that library's lookup, null in every process that did not load it first."""
    for name in ("libgdf.so", "librmm.so"):
        und = subprocess.check_output(["nm", "-D", "--undefined-only", os.path.join(LIBDIR, name)]).decode()
        assert "getenv" not in und, name
    exported = subprocess.check_output(["nm", "-D", "--defined-only", os.path.join(LIBDIR, "libgdf.so")]).decode()
    assert "debug_force" not in exported and "testhook" not in exported
    und = subprocess.check_output(["nm", "-D", "--undefined-only", os.path.join(LIBDIR, "libgdf.so")]).decode()
    assert any(l.split()[:1] == ["w"] and "gdf_amd_testhook_forced" in l for l in und.splitlines()), "the lookup must be a WEAK reference"
    assert hook.gdf_amd_debug_force(b"GDF_JK_NO_SPEC", b"1") == 0
    assert hook.gdf_amd_testhook_forced(b"GDF_JK_NO_SPEC") == b"1"
    assert hook.gdf_amd_debug_force(b"GDF_JK_NO_SPEC", None) == 0
    assert hook.gdf_amd_testhook_forced(b"GDF_JK_NO_SPEC") is None
    assert hook.gdf_amd_debug_force(None, None) != 0
    lab = os.path.join(LIBDIR, "lab", "libgdf.so")
    if os.path.exists(lab):        # the LAB build (experiment knobs) is the one that reads the environment
        assert "getenv" in subprocess.check_output(["nm", "-D", "--undefined-only", lab]).decode()


def test_no_exception_crosses_the_c_boundary(libs, hook):
    """the reference lets std::bad_alloc / thrust::system_error escape extern "C"
    (managed_allocator.cuh:34-45, thrust_rmm_allocator.h:44-49); here every relational entry point runs its body through
    gdf_amd::guarded (csrc/common.h) and answers GDF_MEMORYMANAGER_ERROR.  The test hook makes make_key_table -- the first thing
    each entry point does after its argument checks, in front of its std::vector allocations -- throw std::bad_alloc; no device
    is touched before that, so this runs without a GPU.  The process must survive and the calls must work again afterwards
    (argument errors come back as before)."""
    gdf, _ = libs
    from libgdf_amd._binding import gdf_column, gdf_context
    fake = 0x1000                                    # never dereferenced: the forced failure comes first
    def col(dtype=3, size=8):
        c = gdf_column()
        c.data, c.size, c.dtype = fake, size, dtype
        return c
    assert hook.gdf_amd_debug_force(b"GDF_FORCE_HOST_ALLOC_FAILURE", b"1") == 0
    try:
        key, agg, outk, outa = col(), col(), col(), col()
        ctx = gdf_context()
        ctx.flag_method = 1                          # GDF_HASH
        keys = (C.POINTER(gdf_column) * 1)(C.pointer(key))
        outs = (C.POINTER(gdf_column) * 1)(C.pointer(outk))
        for fn in (gdf.gdf_group_by_sum, gdf.gdf_group_by_avg, gdf.gdf_group_by_count):
            assert fn(1, keys, C.byref(agg), None, outs, C.byref(outa), C.byref(ctx)) == 20      # GDF_MEMORYMANAGER_ERROR
        ctx.flag_method = 0                          # GDF_SORT group-by
        assert gdf.gdf_group_by_max(1, keys, C.byref(agg), C.byref(outa), outs, C.byref(outa), C.byref(ctx)) == 20
        ctx.flag_method = 1
        li, ri = gdf_column(), gdf_column()
        idx = (C.c_int * 1)(0)
        for fn in (gdf.gdf_inner_join, gdf.gdf_left_join, gdf.gdf_full_join):
            assert fn(keys, 1, idx, outs, 1, idx, 1, 0, None, C.byref(li), C.byref(ri), C.byref(ctx)) == 20
        inp = (gdf_column * 1)(col())
        outp = (gdf_column * 1)(col())
        inp_p = (C.POINTER(gdf_column) * 1)(C.pointer(inp[0]))
        out_p = (C.POINTER(gdf_column) * 1)(C.pointer(outp[0]))
        offs = (C.c_int * 4)()
        assert gdf.gdf_hash_partition(1, inp_p, idx, 1, 4, out_p, offs, 0) == 20
        h = col()
        assert gdf.gdf_hash(1, keys, 0, C.byref(h)) == 20
        cols = (gdf_column * 1)(col())
        # (d_cols / d_types NULL: gdf_order_by uploads them before it looks at the keys, sqls_ops.cu:1373-1392)
        assert gdf.gdf_order_by(C.c_size_t(8), cols, C.c_size_t(1), None, None, C.c_void_p(fake)) == 20
    finally:
        assert hook.gdf_amd_debug_force(b"GDF_FORCE_HOST_ALLOC_FAILURE", None) == 0
    assert gdf.gdf_group_by_sum(0, None, None, None, None, None, None) == 5          # alive, and answering as before
