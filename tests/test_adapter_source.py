"""Source-level check of adapter/TransMI355X.cc (the TransImpl subclass a maintainer builds against an installed Atlas;
it cannot be compiled in this image): (1) every pure virtual of the reference's TransImpl
(src/atlas/trans/detail/TransImpl.h:38-191) has an override in the adapter with the same name and the same parameter
types, (2) every atlas_amd__ symbol the adapter calls is declared in include/atlas_amd.h and exported by the library."""
import os
import re

import pytest

from atlas_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/src/atlas/trans/detail/TransImpl.h"


def _norm_params(text):
    """parameter TYPES of a declaration, names / defaults / whitespace removed"""
    text = re.sub(r"=\s*util::NoConfig\(\)", "", text)
    out = []
    for p in [q.strip() for q in text.split(",") if q.strip()]:
        p = re.sub(r"\s+", " ", p)
        m = re.match(r"^(.*?)(\b[A-Za-z_][A-Za-z_0-9]*)?(\[\])?$", p)
        base, name, arr = m.group(1).strip(), m.group(2), m.group(3) or ""
        if name and base and name not in ("int", "double", "Field", "FieldSet", "Configuration", "size_t"):
            p = base + arr
        out.append(p.replace(" &", "&").replace("& ", "&").strip())
    return tuple(out)


def _methods(text, pattern):
    text = re.sub(r"//[^\n]*", "", text)
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    found = set()
    for m in re.finditer(pattern, text, re.S):
        found.add((m.group(1), _norm_params(m.group(2))))
    return found


def test_every_atlas_amd_symbol_of_the_adapter_exists():
    src = open(os.path.join(ROOT, "adapter", "TransMI355X.cc")).read()
    header = open(os.path.join(ROOT, "include", "atlas_amd.h")).read()
    used = set(re.findall(r"\b(atlas_amd__[A-Za-z_0-9]+)\s*\(", src))
    assert len(used) >= 10
    for sym in used:
        assert re.search(r"\b" + sym + r"\s*\(", header), f"{sym} not declared in include/atlas_amd.h"
        assert hasattr(_lib.lib, sym), f"{sym} not exported by libatlas_amd.so"


@pytest.mark.skipif(not os.path.exists(REF), reason="the reference checkout is only present in the build container")
def test_adapter_overrides_every_pure_virtual_of_TransImpl():
    ref = open(REF).read()
    ref = ref[ref.index("class TransImpl"):]
    pure = _methods(ref, r"virtual\s+[^;{}()]*?\b([A-Za-z_0-9]+)\s*\(([^;{}]*?)\)\s*const\s*=\s*0\s*;")
    assert len(pure) >= 28, len(pure)
    src = open(os.path.join(ROOT, "adapter", "TransMI355X.cc")).read()
    cls = src[src.index("class TransMI355X"):src.index("TransMI355X::TransMI355X(")]
    mine = _methods(cls, r"\b([A-Za-z_0-9]+)\s*\(([^;{}]*?)\)\s*const\s*override")
    missing = sorted(m for m in pure if m not in mine)
    assert not missing, missing
    assert re.search(r'TransBuilderGrid<TransMI355X>\s+builder\("mi355x",\s*"mi355x"\)', src)
