"""The struct layouts of the C-ABI, pinned three ways (VERDICT r4 item 6 / ADVICE r4): what a C compiler makes of
include/spacer_hip.h == the ctypes structures of spacer_amd/_lib.py == the ctypes stubs quoted in INTEGRATION.md.

A struct that a caller declares shorter than the library's is over-read by the library (round 4: the documented `Plan` stub had 5
fields while `spacer_plan` had 6).  The header is compiled here by gcc into a probe that prints sizeof / offsetof of every field;
the Python declarations are compared field by field, and the fenced ```python blocks of INTEGRATION.md are executed to obtain the
structures a maintainer would copy.  Host-only: no GPU, no library load."""
import ctypes as C
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "spacer_hip.h")
sys.path.insert(0, ROOT)

STRUCTS = {"spacer_plan": "Plan", "spacer_gemm_epilogue": "GemmEpilogue", "spacer_attn_segment": "AttnSegment"}


def _header_fields():
    """{struct: [field, ...]} parsed from the header's typedef bodies (comments stripped; `int a, b;` declares two fields)."""
    src = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    out = {}
    for m in re.finditer(r"typedef struct (\w+) \{(.*?)\} \1;", src, flags=re.S):
        names = []
        for decl in m.group(2).split(";"):
            decl = decl.strip()
            if not decl:
                continue
            first, *rest = decl.split(",")
            names.append(re.findall(r"(\w+)\s*$", first)[0])
            names += [re.findall(r"(\w+)\s*$", r)[0] for r in rest]
        out[m.group(1)] = names
    return out


@pytest.fixture(scope="module")
def c_layout(tmp_path_factory):
    """{struct: (sizeof, {field: offset})} as gcc lays the header's structs out."""
    fields = _header_fields()
    assert set(STRUCTS) <= set(fields), f"header lost a struct: {set(STRUCTS) - set(fields)}"
    d = tmp_path_factory.mktemp("abi")
    lines = ['#include <stdio.h>', '#include <stddef.h>', f'#include "{HEADER}"', "int main(void) {"]
    for s in STRUCTS:
        lines.append(f'  printf("{s} size %zu\\n", sizeof({s}));')
        for f in fields[s]:
            lines.append(f'  printf("{s} {f} %zu\\n", offsetof({s}, {f}));')
    lines += ["  return 0;", "}"]
    (d / "probe.c").write_text("\n".join(lines))
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-o", str(d / "probe"), str(d / "probe.c")], check=True)
    out = subprocess.run([str(d / "probe")], check=True, capture_output=True, text=True).stdout
    lay = {s: [None, {}] for s in STRUCTS}
    for line in out.splitlines():
        s, f, v = line.split()
        if f == "size":
            lay[s][0] = int(v)
        else:
            lay[s][1][f] = int(v)
    return {s: (sz, offs) for s, (sz, offs) in lay.items()}


def _ctypes_layout(cls):
    return C.sizeof(cls), {name: getattr(cls, name).offset for name, *_ in cls._fields_}


def _assert_same(what, got, want):
    (gsz, goff), (wsz, woff) = got, want
    assert list(goff) == list(woff), f"{what}: fields {list(goff)} != header's {list(woff)}"
    assert goff == woff, f"{what}: offsets {goff} != header's {woff}"
    assert gsz == wsz, f"{what}: sizeof {gsz} != header's {wsz}"


def test_lib_py_structures_match_the_header(c_layout):
    from spacer_amd import _lib
    for cname, pyname in STRUCTS.items():
        _assert_same(f"_lib.{pyname}", _ctypes_layout(getattr(_lib, pyname)), c_layout[cname])


def _integration_structs(text):
    """Every C.Structure subclass declared in the fenced python blocks of INTEGRATION.md (the class statements are executed on
    their own: the rest of a block needs torch / the built library)."""
    found = {}
    for block in re.findall(r"```python\n(.*?)```", text, flags=re.S):
        for m in re.finditer(r"^class (\w+)\(C\.Structure\):.*?\n((?:[ \t]+.*\n|\n)+)", block, flags=re.M):
            ns = {"C": C}
            ns.update(found)                     # a later struct may point at an earlier one
            exec(m.group(0), ns)
            found[m.group(1)] = ns[m.group(1)]
    return found


def test_integration_md_stubs_match_the_header(c_layout):
    stubs = _integration_structs(open(os.path.join(ROOT, "INTEGRATION.md")).read())
    by_py = {v: k for k, v in STRUCTS.items()}
    assert "Plan" in stubs and "GemmEpilogue" in stubs, f"INTEGRATION.md no longer shows the struct stubs: {sorted(stubs)}"
    for pyname, cls in stubs.items():
        assert pyname in by_py, f"INTEGRATION.md declares {pyname}, which is not a struct of include/spacer_hip.h"
        _assert_same(f"INTEGRATION.md {pyname}", _ctypes_layout(cls), c_layout[by_py[pyname]])


def test_the_stub_check_catches_a_short_struct(c_layout):
    """The round-4 drift, replayed: a Plan stub without its last field must be rejected."""
    want = c_layout["spacer_plan"]
    names = list(want[1])
    short = "```python\nclass Plan(C.Structure):\n    _fields_ = [" + ", ".join(f'("{n}", C.c_int)' for n in names[:-1]) + "]\n```\n"
    cls = _integration_structs(short)["Plan"]
    with pytest.raises(AssertionError):
        _assert_same("short Plan", _ctypes_layout(cls), want)
