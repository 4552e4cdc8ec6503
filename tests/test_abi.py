"""CPU-side checks of the C-ABI boundary: the library loads, exports every symbol include/gnnlm.h
declares, and the generated ctypes mirrors have the sizes the C compiler sees.  No compute calls."""
import ctypes
import os

import pytest

from gnnlm_amd import _lib


def test_library_loads_and_is_gfx950():
    L = _lib.lib()
    assert L.gnnlm_target_arch() == b"gfx950"
    assert L.gnnlm_abi_version() == _lib.ABI_VERSION == 11


def test_every_declared_symbol_is_exported():
    L = _lib.lib()
    syms = _lib.exported_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(L, s), s


def test_struct_mirrors_match_c_sizes():
    L = _lib.lib()
    assert len(_lib.STRUCTS) >= 9
    for name, st in _lib.STRUCTS.items():
        assert L.gnnlm_sizeof(name.encode()) == ctypes.sizeof(st), name
    assert L.gnnlm_sizeof(b"nope") == 0


def test_code_object_targets_gfx950_only():
    data = open(_lib.LIB_PATH, "rb").read()
    assert b"gfx950" in data
    for other in (b"gfx942", b"gfx90a", b"sm_90"):
        assert other not in data


def test_host_tensors_are_refused():
    import torch
    from gnnlm_amd import ops
    with pytest.raises(_lib.GnnlmError):
        ops.gemm_nt(torch.zeros(4, 4), torch.zeros(4, 4))


def test_product_package_does_not_import_oracle():
    root = os.path.join(_lib.ROOT, "gnn-lm_amd")
    for dirpath, _, files in os.walk(root):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in src and "from oracle" not in src, f


def test_integration_c_snippet_compiles(tmp_path):
    """The C caller shown in INTEGRATION.md section 2.7 is real code against include/gnnlm.h: it must compile as C and as
    C++ (syntax + types only; HIP's allocator is declared by hand, the image's host compiler has no HIP headers on
    its default path)."""
    import re
    import shutil
    import subprocess
    text = open(os.path.join(_lib.ROOT, "INTEGRATION.md")).read()
    sec = text[text.index("### 2.7"):]
    snippet = re.search(r"```c\n(.*?)```", sec, flags=re.S).group(1)
    body = snippet.replace('#include "gnnlm.h"\n', "")
    src = ('#include <stdio.h>\n#include "gnnlm.h"\n'
           '#ifdef __cplusplus\nextern "C"\n#endif\nint hipMalloc(void** p, size_t n);\n'
           "int caller(int64_t n_store, const uint8_t* host_codes, const int32_t* host_vals, gnnlm_hgt_io_t io, void* stream) {\n"
           + body + "    gnnlm_store_destroy(st);\n    return 0;\n}\n")
    for cc, std, name in (("gcc", "-std=c11", "caller.c"), ("g++", "-std=c++17", "caller.cpp")):
        if shutil.which(cc) is None:
            pytest.skip(cc + " not found")
        f = tmp_path / name
        f.write_text(src)
        r = subprocess.run([cc, std, "-fsyntax-only", "-Wall", "-Werror", "-I", os.path.join(_lib.ROOT, "include"), str(f)],
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr


def test_generated_gemm_loop_is_current(tmp_path):
    """csrc/gemm_sched_loop*.inc / gemm_sched_clobbers.inc are written by tools/gen_gemm_sched.py: the committed files are
    what the generator emits (an edit of one without the other would ship a stale main loop)."""
    import subprocess, sys
    root = _lib.ROOT
    subprocess.run([sys.executable, os.path.join(root, "tools", "gen_gemm_sched.py"), str(tmp_path)], check=True, capture_output=True)
    for f in ("gemm_sched_loop.inc", "gemm_sched_loop_t.inc", "gemm_sched_clobbers.inc"):
        assert open(tmp_path / f).read() == open(os.path.join(root, "gnn-lm_amd", "csrc", f)).read(), f
