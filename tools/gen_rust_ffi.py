#!/usr/bin/env python3
"""Emit the `extern "C"` block of crates/voidin_hip/src/ffi.rs from include/voidin_abi.h: one Rust declaration per
exported function, in header order.  INTEGRATION.md section 2 embeds the output (tests/test_abi_symbols.py checks that
every export is bound there).   python tools/gen_rust_ffi.py [--write]   (--write replaces the block in INTEGRATION.md)"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TYPES = {"int": "i32", "float": "f32", "uint32_t": "u32", "uint64_t": "u64", "int32_t": "i32", "void": "core::ffi::c_void", "char": "core::ffi::c_char",
         "VdInstance": "Instance", "VdMeshInfo": "MeshInfo", "VdDrawIndexedIndirect": "DrawIndexedIndirect", "VdCameraUniform": "CameraUniform",
         "VdBvhNode": "BvhNode", "VdTlasNode": "TlasNode", "VdHostFn": "VdHostFn"}


def rust_type(c):
    c = c.strip()
    const = c.startswith("const ")
    c = c[6:] if const else c
    stars = c.count("*")
    base = c.replace("*", "").strip()
    r = TYPES.get(base, base)
    for k in range(stars):
        r = ("*const " if (const and k == 0) else "*mut ") + r
    return r


def declarations():
    src = open(os.path.join(ROOT, "include", "voidin_abi.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    src = src[src.index("typedef struct VdCtx VdCtx;"):]
    out = []
    for m in re.finditer(r"^(const char\*|int|float)\s+(vd_[a-z_0-9]+)\s*\(([^;]*?)\)\s*;", src, flags=re.M | re.S):
        ret, name, args = m.group(1), m.group(2), " ".join(m.group(3).split())
        params = []
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                mm = re.match(r"(.*?)([A-Za-z_][A-Za-z_0-9]*)$", a)
                params.append((mm.group(2), rust_type(mm.group(1))))
        out.append((name, params, "*const core::ffi::c_char" if ret.startswith("const char") else TYPES[ret]))
    return out


def block():
    L = ['extern "C" {']
    for name, params, ret in declarations():
        head = f"    pub fn {name}("
        body = ", ".join(f"{'r#' + n if n in ('in', 'type', 'ref') else n}: {t}" for n, t in params)
        line = f"{head}{body}) -> {ret};"
        if len(line) > 118:      # wrap long declarations
            pad = " " * len(head)
            parts, cur = [], ""
            for piece in body.split(", "):
                if len(head) + len(cur) + len(piece) > 112 and cur:
                    parts.append(cur.rstrip()); cur = ""
                cur += piece + ", "
            parts.append(cur.rstrip(", "))
            line = head + ("\n" + pad).join(parts) + f") -> {ret};"
        L.append(line)
    L.append("}")
    return "\n".join(L)


if __name__ == "__main__":
    b = block()
    if "--write" in sys.argv:
        p = os.path.join(ROOT, "INTEGRATION.md")
        s = open(p).read()
        a, z = s.index("<!-- ffi:begin -->"), s.index("<!-- ffi:end -->")
        s = s[:a] + "<!-- ffi:begin -->\n```rust\n" + b + "\n```\n" + s[z:]
        open(p, "w").write(s)
        print(f"INTEGRATION.md: {len(declarations())} declarations written")
    else:
        print(b)
