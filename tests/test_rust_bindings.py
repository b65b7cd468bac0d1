"""The Rust -sys crate (rust/pbrt-mi355-sys/src/lib.rs) cannot be compiled here (no Rust toolchain), so this test PARSES it
and the C header (include/pbrt_mi355.h) and checks, struct by struct and function by function, that they declare the same
ABI: names, field order, field types, array lengths, sizes and alignments (computed with the C / repr(C) layout rules), and
the same function signatures.  The ctypes binding (pbrt-rs_amd/_capi.py) and the header's static_asserts are checked
against the same sizes."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

C_SCALARS = {"float": ("f32", 4), "double": ("f64", 8), "int32_t": ("i32", 4), "uint32_t": ("u32", 4), "int": ("i32", 4),
             "uint64_t": ("u64", 8), "int64_t": ("i64", 8), "uint8_t": ("u8", 1), "char": ("c_char", 1), "void": ("c_void", 1)}
RUST_SIZE = {"f32": 4, "f64": 8, "i32": 4, "u32": 4, "u64": 8, "i64": 8, "u8": 1}


def _strip_c(text):
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return re.sub(r"//[^\n]*", "", text)


def parse_c_header():
    text = _strip_c(open(os.path.join(ROOT, "include", "pbrt_mi355.h")).read())
    structs = {}
    for body, name in re.findall(r"typedef\s+struct\s+\w+\s*\{(.*?)\}\s*(\w+)\s*;", text, flags=re.S):
        fields = []
        for decl in body.split(";"):
            decl = " ".join(decl.split())
            if not decl:
                continue
            m = re.match(r"(const\s+)?(\w+)\s*(\*?)\s*(.*)$", decl)
            const, ty, star0, rest = m.groups()
            for item in rest.split(","):
                item = item.strip()
                star = star0 or ("*" if item.startswith("*") else "")
                item = item.lstrip("* ")
                am = re.match(r"(\w+)\s*(?:\[(\d+)\])?$", item)
                fname, alen = am.group(1), am.group(2)
                if star:
                    rty = ("*const " if const else "*mut ") + C_SCALARS[ty][0]
                else:
                    rty = C_SCALARS[ty][0] if ty in C_SCALARS else ty
                if alen:
                    rty = f"[{rty}; {alen}]"
                fields.append((fname, rty))
        structs[name] = fields
    funcs = {}
    for ret, name, args in re.findall(r"^\s*((?:const\s+)?\w+\s*\*?)\s*(pm_\w+)\s*\(([^;{]*?)\)\s*;", text, flags=re.M | re.S):
        params = []
        for a in " ".join(args.split()).split(","):
            a = a.strip()
            if a == "void":
                continue
            m = re.match(r"(const\s+)?(\w+)\s*(\*{0,2})\s*(\w+)$", a)
            const, ty, stars, pname = m.groups()
            base = C_SCALARS[ty][0] if ty in C_SCALARS else ty
            if stars == "":
                rty = base
            elif stars == "*":
                rty = ("*const " if const else "*mut ") + base
            else:
                rty = "*mut *mut " + base
            params.append(rty)
        ret = " ".join(ret.split())
        if ret == "void":
            r = None
        elif ret == "int":
            r = "i32"
        elif ret in C_SCALARS:
            r = C_SCALARS[ret][0]
        else:
            r = "*const c_char" if "char" in ret else ret
        funcs[name] = (params, r)
    return structs, funcs


def parse_rust():
    text = open(os.path.join(ROOT, "rust", "pbrt-mi355-sys", "src", "lib.rs")).read()
    text = re.sub(r"//[^\n]*", "", text)
    structs = {}
    for name, body in re.findall(r"pub struct (\w+)\s*\{(.*?)\n\}", text, flags=re.S):
        fields = []
        for fname, ty in re.findall(r"(?:pub\s+)?(\w+)\s*:\s*([^,\n]+)", body):
            fields.append((fname[:-1] if fname.endswith("_") and fname != "_" else fname, ty.strip()))
        structs[name] = fields
    funcs = {}
    for name, args, ret in re.findall(r"pub fn (\w+)\s*\((.*?)\)\s*(?:->\s*([^;]+))?;", text, flags=re.S):
        params = [a.split(":", 1)[1].strip().replace("std::ffi::c_void", "c_void") for a in args.split(",") if ":" in a]
        funcs[name] = (params, ret.strip() if ret else None)
    consts = dict(re.findall(r"pub const (\w+)\s*:\s*\w+\s*=\s*([^;]+);", text))
    return structs, funcs, consts


def layout(fields, structs):
    """(size, align) under the C / repr(C) rules on LP64"""
    off, align = 0, 1
    for _, ty in fields:
        n = 1
        m = re.match(r"\[(.+); (\d+)\]$", ty)
        if m:
            ty, n = m.group(1), int(m.group(2))
        if ty.startswith("*"):
            s = a = 8
        elif ty in RUST_SIZE:
            s = a = RUST_SIZE[ty]
        else:
            s, a = layout(structs[ty], structs)
        off = (off + a - 1) // a * a + s * n
        align = max(align, a)
    return (off + align - 1) // align * align, align


def test_structs_agree_field_by_field():
    cs, _ = parse_c_header()
    rs, _, _ = parse_rust()
    assert len(cs) >= 13
    for name, cf in cs.items():
        assert name in rs, f"lib.rs lacks struct {name}"
        assert rs[name] == cf, f"{name}: header {cf} vs lib.rs {rs[name]}"
        assert layout(cf, cs) == layout(rs[name], rs), name
    assert set(rs) - {"pm_ctx"} == set(cs)


def test_sizes_match_static_asserts_and_ctypes(pm):
    cs, _ = parse_c_header()
    text = open(os.path.join(ROOT, "include", "pbrt_mi355.h")).read()
    asserted = {n: int(v) for n, v in re.findall(r"PM_STATIC_ASSERT\(sizeof\((\w+)\) == (\d+)", text)}
    assert set(asserted) == set(cs)
    for name, fields in cs.items():
        size, _ = layout(fields, cs)
        assert size == asserted[name], (name, size, asserted[name])
        ct = getattr(pm._capi, name, None)
        if ct is not None:
            assert C.sizeof(ct) == size, (name, C.sizeof(ct), size)
            assert [f[0] for f in ct._fields_] == [f[0] for f in fields], name
    # every ctypes struct of the binding is in the header
    for name in dir(pm._capi):
        obj = getattr(pm._capi, name)
        if isinstance(obj, type) and issubclass(obj, C.Structure) and obj is not C.Structure:
            assert name in cs, name


def test_functions_agree():
    _, cf = parse_c_header()
    _, rf, _ = parse_rust()
    assert len(cf) >= 30
    assert set(cf) == set(rf), sorted(set(cf) ^ set(rf))
    for name, (params, ret) in cf.items():
        rparams, rret = rf[name]
        assert rparams == params, f"{name}: header {params} vs lib.rs {rparams}"
        assert rret == ret, f"{name}: return {ret} vs {rret}"


def test_constants_agree():
    text = _strip_c(open(os.path.join(ROOT, "include", "pbrt_mi355.h")).read())
    _, _, consts = parse_rust()
    cvals = {}
    for name, val in re.findall(r"#define\s+(PM_[A-Z_]+)\s+([^\n]+)", text):
        if name.startswith("PM_STATIC"):
            continue
        cvals[name] = val.strip()
    for body in re.findall(r"enum\s*\{(.*?)\}", text, flags=re.S):
        for name, val in re.findall(r"(PM_[A-Z_]+)\s*=\s*(\d+)", body):
            cvals[name] = val

    def norm(v):
        v = v.replace("_", "").replace("u", "").replace("(", "").replace(")", "").strip()
        if "<<" in v:
            a, b = v.split("<<")
            return int(a, 0) << int(b, 0)
        return int(v, 0)
    for name, val in cvals.items():
        assert name in consts, f"lib.rs lacks {name}"
        assert norm(consts[name]) == norm(val), name
