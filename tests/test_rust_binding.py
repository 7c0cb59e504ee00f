"""bindings/rust/pmx_sys.rs (the `extern "C"` block a pharmsol maintainer links against) is generated
from include/pmx.h by tools/gen_rust_binding.py.  Rust cannot be compiled in this image, so the file is checked instead:
it is in sync with the header, every `#[repr(C)]` struct has the C layout (field order, offsets and sizeof computed
with the repr(C) rules and compared with ctypes' view of the same header and with the library's own
pmx_sizeof_struct()), and every function the header declares is bound and exported."""
import ctypes as C
import os
import re
import subprocess
import sys

from pharmsol_amd import _abi, _ffi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RS = os.path.join(ROOT, "bindings", "rust", "pmx_sys.rs")

PRIM = {"i8": (1, 1), "u8": (1, 1), "i16": (2, 2), "u16": (2, 2), "i32": (4, 4), "u32": (4, 4), "i64": (8, 8), "u64": (8, 8),
        "f64": (8, 8), "f32": (4, 4), "usize": (8, 8), "c_char": (1, 1)}


def parse_rs():
    text = open(RS).read()
    structs = {}
    for m in re.finditer(r"#\[repr\(C\)\]\s*(?:#\[derive\([^)]*\)\]\s*)?pub struct (\w+) \{(.*?)\}", text, flags=re.S):
        fields = [(f.group(1).replace("r#", ""), f.group(2).strip()) for f in re.finditer(r"pub ([\w#]+): ([^,]+),", m.group(2))]
        structs[m.group(1)] = fields
    funcs = re.findall(r"pub fn (pmx_\w+)\(", text)
    consts = dict((a, int(b)) for a, b in re.findall(r"pub const (\w+): i32 = (-?\d+);", text))
    return structs, funcs, consts


def layout(structs, name, cache):
    """(size, align, [(field, offset)]) under the repr(C) rules."""
    if name in cache:
        return cache[name]

    def size_align(t):
        t = t.strip()
        if t.startswith("*"):
            return 8, 8
        am = re.match(r"\[(.+); (\d+)\]$", t)
        if am:
            s, a = size_align(am.group(1))
            return s * int(am.group(2)), a
        if t in PRIM:
            return PRIM[t]
        s, a, _ = layout(structs, t, cache)
        return s, a

    off, align, offs = 0, 1, []
    for fname, t in structs[name]:
        s, a = size_align(t)
        off = (off + a - 1) // a * a
        offs.append((fname, off))
        off += s
        align = max(align, a)
    size = (off + align - 1) // align * align
    cache[name] = (size, align, offs)
    return cache[name]


def test_generated_file_is_in_sync_with_the_header():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_rust_binding.py"), "--check"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr


def test_struct_layouts_match_the_library_and_the_ctypes_binding():
    structs, _, _ = parse_rs()
    L = _ffi.lib()
    cache = {}
    checked = 0
    for name, fields in structs.items():
        if not fields or fields[0][0] == "_private":  # opaque handles
            continue
        size, _, offs = layout(structs, name, cache)
        assert L.pmx_sizeof_struct(name.encode()) == size, name
        ct = getattr(_abi, name)
        assert C.sizeof(ct) == size, name
        assert [f for f, _ in offs] == [f[0] for f in ct._fields_], name  # same fields in the same order
        for fname, off in offs:
            assert getattr(ct, fname).offset == off, (name, fname)
        checked += 1
    assert checked >= 8
    # the two structs a binding fills by hand, field count pinned so that a header change cannot go unnoticed
    assert len(structs["pmx_population_desc"]) == 19 and structs["pmx_population_desc"][-2:] == [
        ("ev_errorpoly", "*const f64"), ("ev_censor", "*const i8")]
    assert len(structs["pmx_model_desc"]) == 24
    assert L.pmx_sizeof_struct(b"no_such_struct") == -1


def test_every_declared_function_is_bound_and_exported():
    _, funcs, consts = parse_rs()
    declared = {name for name, _, _ in _ffi.SYMBOLS}
    assert set(funcs) == declared
    L = C.CDLL(_ffi.LIB_PATH)
    for f in funcs:
        assert hasattr(L, f), f
    assert consts["PMX_ABI_VERSION"] == _abi.PMX_ABI_VERSION == _ffi.lib().pmx_abi_version()
    for k in ("PMX_ERR_PAIR_FAILED", "PMX_PAIR_BAD_LAG", "PMX_K_CUSTOM", "PMX_FN_EQ", "PMX_CENSOR_ALOQ", "PMX_MAX_PARAMS"):
        assert consts[k] == getattr(_abi, k), k


# --------------------------------------------------------------------------- the safe layer (bindings/rust/hip.rs)
HIP_RS = os.path.join(ROOT, "bindings", "rust", "hip.rs")


def _split_args(s):
    """Top-level comma split of a call's argument text."""
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur)
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur)
    return out


def _call_args(text, start):
    """Argument text of the call whose '(' is at `start`."""
    depth = 0
    for i in range(start, len(text)):
        depth += text[i] == "("
        depth -= text[i] == ")"
        if depth == 0:
            return text[start + 1:i]
    raise AssertionError("unbalanced call")


def test_safe_layer_calls_match_the_raw_binding():
    """hip.rs cannot be compiled here (no rustc): every pmx_* call names a bound function with the declared number of
    arguments, every PMX_* constant exists, and the pmx_population_desc literal names each field once, in order."""
    rs = open(RS).read()
    arity = {}
    for m in re.finditer(r"pub fn (pmx_\w+)\((.*?)\)(?: -> [^;]+)?;", rs):
        arity[m.group(1)] = len(_split_args(m.group(2)))
    structs, funcs, consts = parse_rs()
    code = "\n".join(ln.split("//")[0] for ln in open(HIP_RS).read().splitlines())  # comments dropped
    n_calls = 0
    for m in re.finditer(r"\b(pmx_[a-z_0-9]+)\(", code):
        name = m.group(1)
        assert name in arity, f"hip.rs calls {name}, which pmx_sys.rs does not declare"
        got = len(_split_args(_call_args(code, m.end() - 1)))
        assert got == arity[name], f"{name}: hip.rs passes {got} arguments, the binding declares {arity[name]}"
        n_calls += 1
    assert n_calls >= 20
    for name in set(re.findall(r"\b(PMX_[A-Z_0-9]+)\b", code)):
        assert name in consts, f"hip.rs uses {name}, which pmx_sys.rs does not define"
    lit = re.search(r"pmx_population_desc \{\n(.*?)\n        \}", code, flags=re.S)
    fields = re.findall(r"^\s+(\w+):", lit.group(1), flags=re.M)
    assert fields == [f for f, _ in structs["pmx_population_desc"]]
    for m in re.finditer(r"pmx_error_model \{([^}]*)\}", code):
        assert [f.split(":")[0].strip() for f in _split_args(m.group(1))] == [f for f, _ in structs["pmx_error_model"]]
    # every struct type the layer names is a struct (or opaque handle) of the raw binding
    for t in set(re.findall(r"\b(pmx_[a-z_]+)\b(?!\()", code)) - set(arity) - {"pmx_sys", "pmx_hip"}:
        assert re.search(r"pub struct %s\b" % t, rs), t
