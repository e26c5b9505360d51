"""include/atlas_amd.h: "a null handle is an error, never a crash" -- for EVERY entry point that takes a handle.  The prototypes are
read from the header; each function is called with a null pointer for every pointer argument and 0 for the rest, in a child
process (a crash must fail the test, not end the test session).  No device is touched: the handle check comes first, which is the
point.  (atlas_amd__Trans__timings[_vordiv] and 60 other getters dereferenced the handle unchecked.)"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
BOOLEAN = ("atlas_amd__LegendreCacheCreator__supported",)     # a yes / no answer: "no" for no grid
import ctypes as C, re, sys
sys.path.insert(0, ROOT)
from atlas_amd import _lib
lib = _lib._load()
for name, rtype, args in PROTOS:
    fn = getattr(lib, name)
    argtypes = []
    for a in args:
        if "*" in a or "[" in a:
            argtypes.append(C.c_void_p)
        elif re.search(r"\b(double|float)\b", a):
            argtypes.append(C.c_double if "double" in a else C.c_float)
        elif re.search(r"\b(long long|int64_t|size_t|long)\b", a):
            argtypes.append(C.c_longlong)
        else:
            argtypes.append(C.c_int)
    fn.argtypes = argtypes
    fn.restype = None if rtype == "void" else (C.c_void_p if "*" in rtype else (C.c_double if rtype == "double" else C.c_longlong if "64" in rtype or "long" in rtype else C.c_int))
    print("CALL", name, flush=True)
    r = fn(*[None if t is C.c_void_p else 0 for t in argtypes])
    if rtype != "void" and not name.endswith("__delete") and name not in BOOLEAN:
        ok = (r in (None, 0)) if "*" in rtype else (r != 0)     # pointer: NULL; int status: nonzero / -1
        if not ok:
            print("BADRET", name, r, flush=True)
print("DONE", flush=True)
'''


def _prototypes():
    text = open(os.path.join(ROOT, "include", "atlas_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"//[^\n]*", "", text)
    protos = []
    for m in re.finditer(r"^\s*((?:const\s+)?[\w]+(?:\s+[\w]+)?\s*\**)\s*(atlas_amd__\w+)\s*\(([^;{]*?)\)\s*;", text, re.M):
        rtype, name, args = m.group(1).strip(), m.group(2), m.group(3)
        arglist = [a.strip() for a in re.sub(r"\s+", " ", args).split(",")] if args.strip() not in ("", "void") else []
        if any(re.search(r"atlas_amd_\w+\s*\*", a) for a in arglist):
            protos.append((name, rtype, arglist))
    return protos


def test_every_entry_point_with_a_handle_survives_a_null_handle():
    protos = _prototypes()
    assert len(protos) > 120, len(protos)          # the header declares ~150 functions that take a handle
    # constructors that take ANOTHER object's handle (Trans__new(grid, ..)) are included: a null grid is an error there too
    code = f"ROOT = {ROOT!r}\nPROTOS = {protos!r}\n" + CHILD
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    calls = [ln.split()[1] for ln in p.stdout.splitlines() if ln.startswith("CALL")]
    assert p.returncode == 0 and "DONE" in p.stdout, \
        f"crashed (rc {p.returncode}) in {calls[-1] if calls else '?'} after {len(calls)} calls\n{p.stderr[-2000:]}"
    bad = [ln for ln in p.stdout.splitlines() if ln.startswith("BADRET")]
    assert not bad, bad
