"""Code-object check of the SHIPPED library (CPU, no GPU): the kernels of the headline path are compiled without
register spills or scratch.  Two kinds of kernel REQUIRE that: the lean Legendre kernels count
their `s_waitcnt`s by hand around inline-assembly loads, and the row_ct3 Fourier kernels park an L2 prefetch in a
register the compiler must never move (csrc/fft_ct_rows.h); both step aside at run time if a toolchain spilled them,
which costs speed silently -- this test makes it loud at build time instead."""
import os
import re
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import kernel_resources as kr  # noqa: E402

pytestmark = pytest.mark.skipif(not os.path.exists(os.path.join(kr.LLVM, "llvm-readelf")),
                                reason="llvm-readelf of the ROCm toolchain not found")


@pytest.fixture(scope="module")
def kernels(lib_built):
    ks = kr.kernels()
    assert len(ks) > 150, "the code objects of libatlas_amd.so were not found / parsed"
    return {kr.short(k["demangled"]): k for k in ks}


# TL1279 -> O1280 / 137 levels launches exactly these (profiles/r05_*_kernel_stats.txt); FAST = <S, false, true>
HEADLINE = [
    r"trans::legendre_kernel_lean$",
    r"trans::legendre_kernel_lean_f32$",
    r"trans::legendre_kernel_lean_n<[12], (double|float)>$",
    r"trans::legendre_kernel_lean_f32_w2$",                      # [r6] fp32 stage on pairs of latitude tiles
    r"trans::legendre_kernel_lean_n_f32_w2<[12]>$",
    r"trans::fft_rows_ct_kernel<fft::CtShape<(1, 12|15, 8|9, 9|5, 10|3, 11)>, (false|true), true>$",     # row_ct3
    r"trans::fft_rows_ct_pair_kernel<fft::CtShape<(1, 12|15, 8|9, 9|5, 10|3, 11)>, true>$",              # row_ct3, fp32 pairs
    r"trans::fft_rows_ct_kernel<fft::CtShape<(3, 10|1, 11|5, 9|9, 8|5, 8|3, 9|1, 10|3, 8|1, 9|5, 7|1, 8|3, 7|5, 6)>, false, false>$",
    r"trans::fft_rows_dct_kernel<fft::CtShape<(5, 9|1, 11|3, 9|1, 9|9, 8|3, 8|1, 10|1, 8|5, 7|5, 8)>, false, false>$",
    r"trans::fft_rows_dct_pair_kernel<.*>$",
    r"trans::fft_rows_coarse(_multi)?_kernel$",
    r"trans::fft_rows_pair_kernel$",                  # [r6] run-time shaped rows of the fp32 variant, two fields per job
    r"trans::dft_gemm_kernel<8, (false|true)>$",      # [r6] no_nest targets / rows beyond the LDS: the Fourier sum as an fp64-MFMA matrix product
    r"trans::pack_rows_kernel.*", r"trans::spectra_prepare_kernel.*", r"trans::vd2uv_kernel.*",
    r"halo::.*",
]


def test_headline_kernels_have_no_spills_and_no_scratch(kernels):
    checked = 0
    bad = []
    for pat in HEADLINE:
        hits = [n for n in kernels if re.search(pat, n)]
        if pat.startswith(r"trans::legendre_kernel_lean") or "ct_kernel" in pat or "dct_kernel" in pat:
            assert hits, f"no kernel matches {pat}: the list in this test is stale"
        for n in hits:
            k = kernels[n]
            checked += 1
            if k["vgpr_spill_count"] or k["scratch"] or k["dynamic_stack"]:
                bad.append((n, k["vgpr_count"], k["vgpr_spill_count"], k["scratch"]))
    assert checked >= 40
    assert not bad, "kernels of the headline path compiled with spills / scratch: " + repr(bad)


def test_fp32_tile_pair_legendre_kernels_keep_two_workgroups_per_cu(kernels):
    # [r6] 105 VGPRs: four wavefronts per SIMD = two 8-wavefront workgroups per CU (the form lost its third workgroup against the 64-latitude
    # kernel's 73 registers and still wins; a fifth register class -- > 128 -- would halve its occupancy)
    k = kernels["trans::legendre_kernel_lean_f32_w2"]
    assert k["vgpr_count"] + k["agpr_count"] <= 128 and k["wg"] == 512


def test_round6_kernels_keep_their_occupancy(kernels):
    # the regional matrix-product kernel: 8 wavefronts per workgroup at <= 128 registers = two workgroups per CU (4 per SIMD; measured 8 %
    # faster than 4 wavefronts at 200 registers); the two-field run-time shaped rows: <= 128 registers as well (the one-field fp64 form: 194)
    for name in ("trans::dft_gemm_kernel<8, false>", "trans::dft_gemm_kernel<8, true>"):
        k = kernels[name]
        assert k["vgpr_count"] + k["agpr_count"] <= 128 and k["wg"] == 512 and not k["vgpr_spill_count"]
    k = kernels["trans::fft_rows_pair_kernel"]
    assert k["vgpr_count"] + k["agpr_count"] <= 128 and not k["vgpr_spill_count"]


def test_row_ct3_instances_fit_two_wavefronts_per_simd(kernels):
    # 256 registers = two wavefronts per SIMD = the two 4-wavefront workgroups per CU that LDS allows
    for n, k in kernels.items():
        if re.search(r"fft_rows_ct_kernel<.*, true>$", n) or re.search(r"fft_rows_ct_pair_kernel<.*, true>$", n):
            assert k["vgpr_count"] + k["agpr_count"] <= 256, n
            assert k["wg"] == 256, n


def test_table_lists_every_kernel_with_its_spills(kernels, capsys):
    kr.main([])
    out = capsys.readouterr().out
    ks = kr.kernels()
    assert f"# {len(ks)} kernels" in out
    spilled = [k for k in ks if k["vgpr_spill_count"] or k["scratch"]]
    assert f"scratch: {len(spilled)}" in out
