#!/usr/bin/env python3
"""Per-kernel resources of the SHIPPED library, read from its code objects (no GPU needed).

libatlas_amd.so carries one `__CLANG_OFFLOAD_BUNDLE__` block per translation unit in `.hip_fatbin`; every block holds a
gfx950 ELF whose `NT_AMDGPU_METADATA` note lists, per kernel, `.vgpr_count`, `.vgpr_spill_count`, `.sgpr_spill_count`,
`.private_segment_fixed_size` (scratch bytes per lane), `.group_segment_fixed_size` (static LDS).  This module splits the
blocks, runs `llvm-readelf --notes` on each ELF and returns one record per kernel; `tests/test_kernel_resources.py` asserts
that the kernels of the headline path have no spills, `python tools/kernel_resources.py > profiles/rNN_kernel_resources.txt`
writes the table for all of them.
"""
import os
import re
import struct
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "atlas_amd", "lib", "libatlas_amd.so")
LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def code_objects(lib=LIB):
    """the gfx950 ELF images inside `lib`, one per translation unit with device code"""
    d = open(lib, "rb").read()
    out = []
    for m in re.finditer(MAGIC, d):
        o = m.start()
        (n,) = struct.unpack_from("<Q", d, o + 24)
        q = o + 32
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", d, q)
            q += 24
            triple = d[q:q + tl].decode()
            q += tl
            if size and "gfx950" in triple:
                out.append(d[o + off:o + off + size])
    return out


def demangle(names):
    p = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True)
    if p.returncode != 0:
        return list(names)
    return p.stdout.splitlines()


def kernels(lib=LIB):
    """[{name, demangled, vgpr_count, agpr_count, sgpr_count, vgpr_spill_count, sgpr_spill_count, scratch, lds, wg}]"""
    import yaml
    out, seen = [], set()
    with tempfile.TemporaryDirectory() as tmp:
        for i, img in enumerate(code_objects(lib)):
            path = os.path.join(tmp, f"co{i}.elf")
            open(path, "wb").write(img)
            txt = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", path],
                                 capture_output=True, text=True, check=True).stdout
            for doc in re.findall(r"^\s*---\n(.*?)^\s*\.\.\.", txt, flags=re.S | re.M):
                meta = yaml.safe_load(doc)
                for r in (meta or {}).get("amdhsa.kernels", []):
                    name = r[".name"]
                    if name in seen:
                        continue
                    seen.add(name)
                    out.append({
                        "name": name,
                        "vgpr_count": int(r.get(".vgpr_count", 0)),
                        "agpr_count": int(r.get(".agpr_count", 0)),
                        "sgpr_count": int(r.get(".sgpr_count", 0)),
                        "vgpr_spill_count": int(r.get(".vgpr_spill_count", 0)),
                        "sgpr_spill_count": int(r.get(".sgpr_spill_count", 0)),
                        "scratch": int(r.get(".private_segment_fixed_size", 0)),
                        "lds": int(r.get(".group_segment_fixed_size", 0)),
                        "wg": int(r.get(".max_flat_workgroup_size", 0)),
                        "dynamic_stack": bool(r.get(".uses_dynamic_stack", False)),
                    })
    for r, dm in zip(out, demangle([r["name"] for r in out])):
        r["demangled"] = re.sub(r"\s+", " ", dm)
    return out


def waves_per_simd(r):
    """wavefronts per SIMD the register file allows (512 VGPRs per lane and SIMD on gfx950, unified with AGPRs,
    allocation granule 8), capped at 8"""
    v = max(8, (r["vgpr_count"] + r["agpr_count"] + 7) // 8 * 8)
    return min(8, 512 // v)


def short(dm):
    dm = re.sub(r"^void ", "", dm)
    dm = dm.replace("atlas_amd::", "").replace("(anonymous namespace)::", "")   # before the argument list is cut at its "("
    dm = re.sub(r"\(.*$", "", dm)
    return dm


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    lib = argv[0] if argv else LIB
    ks = sorted(kernels(lib), key=lambda r: short(r["demangled"]))
    print(f"# {len(ks)} kernels in {os.path.relpath(lib, ROOT)} (code-object metadata; tools/kernel_resources.py)")
    print("# vgpr agpr sgpr vspill sspill scratchB staticLDS wg waves/SIMD(regs)  kernel")
    for r in ks:
        print(f"{r['vgpr_count']:4d} {r['agpr_count']:4d} {r['sgpr_count']:4d} {r['vgpr_spill_count']:6d} "
              f"{r['sgpr_spill_count']:6d} {r['scratch']:8d} {r['lds']:9d} {r['wg']:4d} {waves_per_simd(r):3d}  "
              f"{short(r['demangled'])}")
    nsp = [r for r in ks if r["vgpr_spill_count"] or r["scratch"]]
    print(f"# kernels with VGPR spills or scratch: {len(nsp)}")


if __name__ == "__main__":
    main()
