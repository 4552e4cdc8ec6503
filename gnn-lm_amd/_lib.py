"""ctypes binding of libgnnlm_hip.so (C ABI: include/gnnlm.h).

The descriptor structs are generated from the header itself, so the header stays the single source
of truth.  There is NO fallback: if the shared library is missing or is not the gfx950 build,
importing a kernel entry point raises (the product path must fail loudly without the HIP library).
"""
import ctypes
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(_HERE)
HEADER = os.path.join(ROOT, "include", "gnnlm.h")
LIB_PATH = os.environ.get("GNNLM_LIB") or os.path.join(_HERE, "lib", "libgnnlm_hip.so")   # GNNLM_LIB: A/B builds

_SCALARS = {
    "int32_t": ctypes.c_int32, "int64_t": ctypes.c_int64, "float": ctypes.c_float,
    "double": ctypes.c_double, "size_t": ctypes.c_size_t, "uint8_t": ctypes.c_uint8, "int": ctypes.c_int,
    "uint16_t": ctypes.c_uint16, "uint32_t": ctypes.c_uint32,
}


def _parse_structs(text):
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    structs = {}
    for m in re.finditer(r"typedef\s+struct\s+(\w+)\s*\{(.*?)\}\s*(\w+)\s*;", text, flags=re.S):
        fields = []
        for stmt in m.group(2).split(";"):
            stmt = " ".join(stmt.split())
            if not stmt:
                continue
            mm = re.match(r"(const\s+)?(\w+)\s*(.*)", stmt)
            base, rest = mm.group(2), mm.group(3)
            for decl in rest.split(","):
                decl = decl.strip()
                ptr = decl.startswith("*")
                decl = decl.lstrip("* ")
                arr = re.match(r"(\w+)\s*\[(\d+)\]", decl)
                name = arr.group(1) if arr else decl
                if ptr:
                    ctype = ctypes.c_void_p
                elif base in _SCALARS:
                    ctype = _SCALARS[base]
                elif base in structs:                     # a struct of the header embedded by value
                    ctype = structs[base]
                else:
                    raise ValueError(f"gnnlm.h: unsupported field type {base!r} in {m.group(3)}")
                if arr:
                    ctype = ctype * int(arr.group(2))
                fields.append((name, ctype))
        structs[m.group(3)] = type(m.group(3), (ctypes.Structure,), {"_fields_": fields})
    return structs


def _normalise(text):
    # "const float* A" -> "const float *A" so that the pointer star belongs to the declarator
    return re.sub(r"(\w)\*\s*(\w)", r"\1 *\2", text)


ABI_VERSION = int(re.search(r"#define\s+GNNLM_ABI_VERSION\s+(\d+)", open(HEADER).read()).group(1))
STRUCTS = _parse_structs(_normalise(open(HEADER).read()))
globals().update(STRUCTS)


def exported_symbols():
    """Every function the header declares (used by the CPU-side ABI test)."""
    text = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    return sorted(set(re.findall(r"\b(gnnlm_\w+)\s*\(", text)))


class GnnlmError(RuntimeError):
    pass


_lib = None


def lib():
    """Load the shared library (once).  Raises if it is absent -- there is no CPU fallback."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise GnnlmError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  gnnlm_amd has no CPU fallback.")
        # torch first: its wheel bundles a HIP runtime with the same SONAME as /opt/rocm's
        # (libamdhip64.so.7); whichever is loaded first serves the whole process, and device memory
        # handed over from torch must belong to the runtime the kernels are launched on.
        import torch  # noqa: F401
        L = ctypes.CDLL(LIB_PATH)
        L.gnnlm_last_error.restype = ctypes.c_char_p
        L.gnnlm_target_arch.restype = ctypes.c_char_p
        L.gnnlm_adaptive_workspace_bytes.restype = ctypes.c_size_t
        L.gnnlm_hgt_workspace_bytes.restype = ctypes.c_size_t
        L.gnnlm_sizeof.restype = ctypes.c_size_t
        L.gnnlm_sizeof.argtypes = [ctypes.c_char_p]
        L.gnnlm_kernel_name.restype = ctypes.c_char_p
        L.gnnlm_kernel_name.argtypes = [ctypes.c_int32]
        L.gnnlm_profile_begin.argtypes = [ctypes.c_uint32]
        L.gnnlm_profile_end.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p]
        L.gnnlm_store_codes.restype = ctypes.c_void_p
        L.gnnlm_store_vals.restype = ctypes.c_void_p
        vp, i32, i64, f32 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_float
        L.gnnlm_causal_softmax.argtypes = [vp, i64, i32, i64, i32, vp]
        L.gnnlm_causal_attn.argtypes = [vp, vp, vp, i64, vp, i64, i32, i32, i32, i32, i32, vp]
        L.gnnlm_layernorm.argtypes = [vp, i64, vp, vp, vp, i64, i64, i32, f32, vp, vp]
        L.gnnlm_half_to_float.argtypes = [vp, vp, i64, vp]
        L.gnnlm_filter_neighbors.argtypes = [vp, vp, i64, i32, i64, vp, vp]
        L.gnnlm_gelu.argtypes = [vp, i64, vp]
        L.gnnlm_row_lse_pick.argtypes = [vp, i64, i64, vp, i32, vp, vp, vp, vp]
        L.gnnlm_lse_reduce.argtypes = [vp, i32, i64, vp, vp, vp]
        L.gnnlm_pq_encode.argtypes = [vp, i64, vp, vp, i32, i32, i64, vp, vp]
        L.gnnlm_bucket_rows.argtypes = [vp, i64, i64, i64, i32, i32, vp, vp, vp, vp, vp]
        L.gnnlm_bucket_rows_padded.argtypes = [vp, i64, i64, i64, i32, i64, vp, vp, vp, vp, vp]
        L.gnnlm_enable_peer_access.argtypes = [i32]
        L.gnnlm_adaptive_workspace_bytes.argtypes = [vp, i64]
        L.gnnlm_adaptive_target_logp.argtypes = [vp, vp, i64, vp, i64, vp, vp, ctypes.c_size_t, vp]
        L.gnnlm_masked_sum_f64.argtypes = [vp, vp, i64, vp, vp]
        L.gnnlm_ivfpq_pack_codes.argtypes = [vp, i64, i32, vp, vp]
        L.gnnlm_ivfpq_pack_lut.argtypes = [vp, i64, i64, i32, vp, vp]
        L.gnnlm_ivfpq_pack_tiles.argtypes = [vp, i64, i32, vp, vp]
        L.gnnlm_ivfpq_quantize_lut.argtypes = [vp, i64, i64, i32, vp, vp, vp]
        L.gnnlm_ivfpq_build_groups.argtypes = [vp, i64, i64, i32, i32, i64, vp, vp, vp, vp, vp, vp]
        L.gnnlm_label_tags.argtypes = [vp, i32, i64, vp, vp]
        L.gnnlm_knn_interp_scratch_bytes.argtypes = [i64, i32, i64]
        L.gnnlm_knn_interp_scratch_bytes.restype = ctypes.c_size_t
        L.gnnlm_ivfpq_split_payload.argtypes = [vp, i64, i32, i32, vp, vp]
        L.gnnlm_hgt_workspace_bytes.argtypes = [vp, vp]
        L.gnnlm_hgt_forward.argtypes = [vp, vp, vp, ctypes.c_size_t, vp]
        for nm in ("gnnlm_gemm_nt", "gnnlm_pq_gather_decode", "gnnlm_star_attn", "gnnlm_chain_attn",
                   "gnnlm_knn_interp", "gnnlm_topk_merge", "gnnlm_ivfpq_scan", "gnnlm_gather_rows_peer",
                   "gnnlm_ivfpq_scan8", "gnnlm_ivfpq_rescore", "gnnlm_ivfpq_tau", "gnnlm_ivfpq_refine", "gnnlm_group_assign"):
            getattr(L, nm).argtypes = [vp, vp]
        if L.gnnlm_target_arch() != b"gfx950" or L.gnnlm_abi_version() != ABI_VERSION:
            raise GnnlmError(f"libgnnlm_hip.so is not the gfx950 / ABI-{ABI_VERSION} build")
        _lib = L
    return _lib


def check(rc, what=""):
    if rc != 0:
        raise GnnlmError(f"{what} failed ({rc}): {lib().gnnlm_last_error().decode()}")


def ptr(t):
    """Device pointer of a torch tensor (None -> NULL).  Tensors must be contiguous CUDA tensors."""
    if t is None:
        return None
    if not t.is_cuda:
        raise GnnlmError("gnnlm_amd kernels need device (HIP) tensors; there is no CPU fallback")
    if not t.is_contiguous():
        raise GnnlmError("gnnlm_amd kernels need contiguous tensors")
    return ctypes.c_void_p(t.data_ptr())


def raw_stream(device=None):
    """The current HIP stream of ``device`` (default: the current device) as an integer handle.  torch's
    ``current_stream().cuda_stream`` builds a Stream object per call (~9 us: a step asks for it a dozen times); the raw getter
    is the same handle without the object."""
    import torch
    try:
        idx = device.index if hasattr(device, "index") and device.index is not None else \
            (device if isinstance(device, int) else torch.cuda.current_device())
        return int(torch._C._cuda_getCurrentRawStream(idx))
    except (AttributeError, TypeError):                              # (a torch without the private getter)
        return int(torch.cuda.current_stream(device).cuda_stream)


def stream():
    return ctypes.c_void_p(raw_stream())


def call(name, *args):
    check(getattr(lib(), name)(*args), name)


def profile_begin(mask=0xFFFFFFFF):
    check(lib().gnnlm_profile_begin(mask), "gnnlm_profile_begin")


def profile_end():
    """-> {kernel name: dict(launches, total_ms, flops, bytes)} for the kernels that were launched."""
    L = lib()
    arr = (STRUCTS["gnnlm_profile_entry_t"] * 32)()
    n = ctypes.c_int32(0)
    check(L.gnnlm_profile_end(arr, 32, ctypes.byref(n)), "gnnlm_profile_end")
    out = {}
    for i in range(n.value):
        e = arr[i]
        if e.launches:
            out[L.gnnlm_kernel_name(e.kernel_id).decode()] = {"launches": e.launches, "total_ms": e.total_ms,
                                                              "flops": e.flops, "bytes": e.bytes}
    return out


def call_desc(name, desc):
    check(getattr(lib(), name)(ctypes.byref(desc), stream()), name)
