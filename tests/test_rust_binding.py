"""bindings/rust/zoe_sw_gpu.rs cannot be compiled in this image (no Rust toolchain), so its `extern "C"` block is checked
against include/zoe_sw.h textually: every entry point declared exactly once on both sides, same arity, same integer widths,
same pointer constness, same #[repr(C)] field lists; and every entry point is used by a wrapper."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

C_SCALARS = {
    "int": "i32", "int32_t": "i32", "uint32_t": "u32", "uint8_t": "u8", "uint64_t": "u64", "int8_t": "i8", "size_t": "usize", "float": "f32", "double": "f64",
    "zsw_error": "i32", "zsw_int_type": "i32", "zsw_mem": "i32", "zsw_encoding": "i32", "zsw_option": "i32", "int64_t": "i64", "char": "c_char", "void": "c_void",
    "zsw_context": "ZswContext", "zsw_group": "ZswGroup", "zsw_batch": "ZswBatch", "zsw_alignment": "ZswAlignment",
}


def c_type_to_rust(t: str) -> str:
    """`const uint32_t*` -> `*const u32`, `uint8_t* const*` -> `*const *mut u8`, `zsw_context**` -> `*mut *mut ZswContext`"""
    t = t.strip()
    toks = re.findall(r"const|\*|\w+", t)
    base = [x for x in toks if x not in ("const", "*")]
    assert len(base) == 1, t
    rust = C_SCALARS[base[0]]
    # walk the declarator left to right: a `const` applies to what precedes it (or to the base type if it comes first)
    const_base = toks[0] == "const"
    i = 1 if const_base else 0
    assert toks[i] == base[0], t
    i += 1
    pending_const = const_base
    while i < len(toks):
        if toks[i] == "*":
            rust = ("*const " if pending_const else "*mut ") + rust
            pending_const = False
        elif toks[i] == "const":
            pending_const = True
        i += 1
    # a trailing const on the last pointer (`T* const`) is irrelevant for a by-value parameter
    return rust


def header_prototypes():
    txt = open(os.path.join(ROOT, "include", "zoe_sw.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    out = {}
    for ret, name, args in re.findall(r"([\w\s\*]+?)\b(zsw_\w+)\s*\(([^)]*)\)\s*;", txt):
        ret = ret.strip()
        if ret.startswith("typedef"):
            continue
        params = [] if args.strip() in ("", "void") else [a.strip() for a in args.split(",")]
        ptypes = []
        for p in params:
            m = re.match(r"(.+?)(\w+)$", p)  # type then parameter name
            ptypes.append(c_type_to_rust(m.group(1)))
        assert name not in out, name
        out[name] = (None if ret == "void" else c_type_to_rust(ret), ptypes)
    return out


def rust_prototypes():
    txt = open(os.path.join(ROOT, "bindings", "rust", "zoe_sw_gpu.rs")).read()
    block = txt.split('unsafe extern "C" {')[1].split("\n}\n")[0]
    out = {}
    for name, args, ret in re.findall(r"fn (zsw_\w+)\(([^)]*)\)(?:\s*->\s*([^;]+))?;", block):
        ptypes = [a.split(":", 1)[1].strip() for a in args.split(",") if a.strip()]
        assert name not in out, name
        out[name] = (ret.strip() if ret else None, ptypes)
    return txt, out


def test_every_entry_point_is_declared_with_the_same_signature():
    c = header_prototypes()
    txt, r = rust_prototypes()
    assert len(c) >= 35
    assert set(c) == set(r), (sorted(set(c) - set(r)), sorted(set(r) - set(c)))
    for name, (ret, params) in c.items():
        rret, rparams = r[name]
        assert ret == rret, (name, ret, rret)
        assert params == rparams, (name, params, rparams)
    # and nothing is declared without being called by a wrapper
    body = txt.split("\n}\n", 1)[1] if False else txt
    for name in r:
        assert len(re.findall(r"\b" + name + r"\(", body)) >= 2, f"{name} is declared but never called"


def test_repr_c_structs_match_the_header():
    h = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "zoe_sw.h")).read(), flags=re.S)
    rs = open(os.path.join(ROOT, "bindings", "rust", "zoe_sw_gpu.rs")).read()
    for cname, rname in (("zsw_batch", "ZswBatch"), ("zsw_alignment", "ZswAlignment")):
        cbody = re.search(r"typedef struct " + cname + r"\s*\{(.*?)\}\s*" + cname + ";", h, flags=re.S).group(1)
        cfields = []
        for decl in cbody.split(";"):
            decl = decl.strip()
            if not decl:
                continue
            m = re.match(r"(.+?)((?:\w+\s*,\s*)*\w+)$", decl)
            for fname in m.group(2).split(","):
                cfields.append((fname.strip(), c_type_to_rust(m.group(1))))
        rbody = re.search(r"#\[repr\(C\)\][^{]*pub struct " + rname + r"\s*\{(.*?)\n\}", rs, flags=re.S).group(1)
        rfields = [(n, t.strip()) for n, t in re.findall(r"pub (\w+):\s*([^,\n]+),", rbody)]
        assert cfields == rfields, (cname, cfields, rfields)
    # enum values
    for cname, val in (("ZSW_STATUS_SOME", 0), ("ZSW_STATUS_OVERFLOWED", 1), ("ZSW_STATUS_UNMAPPED", 2), ("ZSW_STATUS_EMPTY", 3), ("ZSW_MEM_HOST", 0), ("ZSW_MEM_DEVICE", 1)):
        assert re.search(cname + r"\s*=\s*" + str(val) + r"\b", h) and re.search(r"pub const " + cname + r": \w+ = " + str(val) + ";", rs), cname
    for k, v in (("I8", 0), ("I16", 1), ("I32", 2), ("U8", 3), ("U16", 4), ("U32", 5)):
        assert re.search(r"ZSW_" + k + r"\s*=\s*" + str(v), h) and re.search(k + r"\s*=\s*" + str(v) + ",", rs)
