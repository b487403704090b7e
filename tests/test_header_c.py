"""include/msm_hip.h is the drop-in boundary: it must be a valid C99 header (a cgo / JNI / bindgen consumer compiles it as C,
not C++), and the three structs that cross it by value must have the layout the hand-written mirrors assume -- the ctypes
structures of mopro_msm_hip and the #[repr(C)] block of the Rust shim (rust/mopro-msm-hip/src/lib.rs).  tests/test_abi.py only
regex-parses the header; here gcc reads it.  Counterpart in the reference: the `metal` crate's typed pipeline arguments
(host/metal_wrapper.rs:55-217) -- a layout mismatch there is a compile error, here it has to be a test."""
import ctypes as C
import os
import re
import subprocess
import sys

import pytest

import mopro_msm_hip as mh
from conftest import ROOT

INC = os.path.join(ROOT, "include")

USE_ALL = r"""
#include "msm_hip.h"
#include "msm_hip_testhooks.h"
/* every declaration is used as a C99 object: taking the address type-checks the prototype */
typedef int32_t (*msm_fn)(msm_ctx *, const uint32_t *, uint32_t, const uint8_t *, const uint32_t *, size_t, uint32_t *, uint32_t *,
                          uint8_t *);
static msm_fn the_call = msm_bn254_g1;
int use(void) {
    msm_config_t cfg = {-1, 0u, MSM_FLAG_NO_GLV, 0u, 0u, MSM_BATCH_LAYOUT_AUTO, 0u};
    msm_plan_t pl;
    msm_timings_t tm;
    (void)cfg; (void)pl; (void)tm; (void)the_call;
    return MSM_OK + MSM_ERR_EMPTY + (int)MSM_FORM_MONT + (int)MSM_MULTI_EXCHANGE_RCCL + (int)MSM_HIP_ABI_VERSION;
}
"""

LAYOUT = r"""
#include <stddef.h>
#include <stdio.h>
#include "msm_hip.h"
#define F(T, f) printf(#T "." #f " %zu %zu\n", offsetof(T, f), sizeof(((T *)0)->f))
int main(void) {
    printf("msm_config_t %zu\n", sizeof(msm_config_t));
    F(msm_config_t, device); F(msm_config_t, window_bits); F(msm_config_t, flags); F(msm_config_t, stream_chunk_log2);
    F(msm_config_t, max_points); F(msm_config_t, batch_layout); F(msm_config_t, host_threads);
    printf("msm_plan_t %zu\n", sizeof(msm_plan_t));
    F(msm_plan_t, window_bits); F(msm_plan_t, num_windows); F(msm_plan_t, num_buckets); F(msm_plan_t, signed_digits);
    F(msm_plan_t, workspace_bytes); F(msm_plan_t, virtual_points); F(msm_plan_t, glv); F(msm_plan_t, scalar_bits);
    F(msm_plan_t, table_factor); F(msm_plan_t, bucket_arrays); F(msm_plan_t, table_bytes); F(msm_plan_t, top_digit_bits);
    F(msm_plan_t, reserved);
    printf("msm_timings_t %zu\n", sizeof(msm_timings_t));
    F(msm_timings_t, h2d_ms); F(msm_timings_t, convert_ms); F(msm_timings_t, decompose_ms); F(msm_timings_t, sort_ms);
    F(msm_timings_t, accumulate_ms); F(msm_timings_t, reduce_ms); F(msm_timings_t, finish_ms); F(msm_timings_t, total_ms);
    F(msm_timings_t, num_points); F(msm_timings_t, num_adds); F(msm_timings_t, stream_chunks); F(msm_timings_t, batch_layout);
    F(msm_timings_t, plan_ms); F(msm_timings_t, combine_ms);
    printf("abi %u\n", (unsigned)MSM_HIP_ABI_VERSION);
    return 0;
}
"""


def test_header_is_valid_c99(tmp_path):
    src = tmp_path / "use_all.c"
    src.write_text(USE_ALL)
    p = subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-fsyntax-only", "-I", INC, str(src)],
                       capture_output=True, text=True)
    assert p.returncode == 0, p.stderr


def _c_layout(tmp_path):
    src, exe = tmp_path / "layout.c", tmp_path / "layout"
    src.write_text(LAYOUT)
    subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-I", INC, "-o", str(exe), str(src)])
    sizes, fields, abi = {}, {}, None
    for line in subprocess.check_output([str(exe)], text=True).splitlines():
        w = line.split()
        if w[0] == "abi":
            abi = int(w[1])
        elif "." in w[0]:
            t, f = w[0].split(".")
            fields.setdefault(t, []).append((f, int(w[1]), int(w[2])))
        else:
            sizes[w[0]] = int(w[1])
    return sizes, fields, abi


def test_struct_layouts_match_the_ctypes_mirrors(tmp_path):
    sizes, fields, abi = _c_layout(tmp_path)
    for cname, mirror in (("msm_config_t", mh.Config), ("msm_plan_t", mh.Plan), ("msm_timings_t", mh.Timings)):
        assert C.sizeof(mirror) == sizes[cname], cname
        got = [(n, getattr(mirror, n).offset, getattr(mirror, n).size) for n, _ in mirror._fields_]
        assert got == fields[cname], (cname, got, fields[cname])
    assert abi == mh.load_library().msm_abi_version()


RUST_TYPES = {"i32": 4, "u32": 4, "u64": 8, "i64": 8, "f32": 4, "usize": 8, "u8": 1}


def _check_rust_binding(txt, where, sizes, fields, min_externs):
    """a Rust source text (the shim, or a ```rust fence of a document): its #[repr(C)] MsmConfig laid out by the C rules must be gcc's
    msm_config_t, and every `fn msm_*` of its extern "C" block must be declared by include/msm_hip.h with the same NUMBER of parameters"""
    m = re.search(r"#\[repr\(C\)\]\s*struct MsmConfig \{(.*?)\}", txt, re.S)
    assert m, f"MsmConfig mirror not found in {where}"
    off, got, align = 0, [], 1
    for name, ty in re.findall(r"(\w+):\s*(\w+)\s*[,}\n]", m.group(1) + "\n"):
        sz = RUST_TYPES[ty]
        off = (off + sz - 1) // sz * sz
        got.append((name, off, sz))
        off += sz
        align = max(align, sz)
    assert got == fields["msm_config_t"], (where, got, fields["msm_config_t"])
    assert (off + align - 1) // align * align == sizes["msm_config_t"], where
    block = txt[txt.index('extern "C"'):]
    block = block[:block.index("\n}\n")]  # the extern block only: methods of the shim's own types are not ABI symbols
    decl = re.findall(r"\bfn (msm_[a-z0-9_]+)\s*\((.*?)\)\s*(?:->\s*[^;]+)?;", block, re.S)
    assert len(decl) >= min_externs, (where, len(decl))
    hdr = re.sub(r"/\*.*?\*/", "", open(os.path.join(INC, "msm_hip.h")).read(), flags=re.S)
    for f, params in decl:
        h = re.search(r"\b%s\s*\((.*?)\)\s*;" % f, hdr, re.S)
        assert h, f"{f} declared by {where} but not by include/msm_hip.h"
        n_rust = len([a for a in params.split(",") if a.strip()])
        n_c = 0 if h.group(1).strip() in ("", "void") else len(h.group(1).split(","))
        assert n_rust == n_c, f"{f}: {n_rust} parameters in {where}, {n_c} in include/msm_hip.h"
    return {f for f, _ in decl}


def test_struct_layouts_match_the_rust_shim(tmp_path):
    """the shim cannot be compiled here (no cargo): its #[repr(C)] mirror of msm_config_t is parsed and laid out by the C rules"""
    sizes, fields, _ = _c_layout(tmp_path)
    txt = open(os.path.join(ROOT, "rust", "mopro-msm-hip", "src", "lib.rs")).read()
    _check_rust_binding(txt, "rust/mopro-msm-hip/src/lib.rs", sizes, fields, 10)


def test_integration_md_prints_the_real_binding(tmp_path):
    """INTEGRATION.md section 2 is what a maintainer copies: round 4 shipped it with the 24-byte ABI-4 MsmConfig while the header's had grown
    to 32.  Every ```rust fence of the document that declares MsmConfig or an extern "C" block is held to the same rules as
    lib.rs, and the extern block must be the shim's own (same functions)."""
    sizes, fields, _ = _c_layout(tmp_path)
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    fences = re.findall(r"```rust\n(.*?)```", doc, re.S)
    binding = [f for f in fences if "struct MsmConfig" in f or 'extern "C"' in f]
    assert binding, "INTEGRATION.md no longer shows the binding"
    shim = open(os.path.join(ROOT, "rust", "mopro-msm-hip", "src", "lib.rs")).read()
    shim_decl = _check_rust_binding(shim, "lib.rs", sizes, fields, 10)
    for k, f in enumerate(binding):
        assert "struct MsmConfig" in f and 'extern "C"' in f, "a binding fence must show the struct AND the externs it is passed to"
        assert _check_rust_binding(f, f"INTEGRATION.md rust fence {k}", sizes, fields, 10) == shim_decl
    # no other place of the document spells the struct's fields out (a second, stale copy)
    assert len(re.findall(r"struct MsmConfig\s*\{", doc)) == len(binding)
