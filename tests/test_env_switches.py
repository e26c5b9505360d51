"""Environment hygiene of the shipped library: ONE table of every ATLAS_AMD_* switch (csrc/env.cpp), read only
through env_get(); sources, table and INTEGRATION.md stay in step; atlas_amd__set_ignore_env / ATLAS_AMD_IGNORE_ENV give the default
configuration in a hostile environment (the GPU test at the end runs the transform under one)."""
import glob
import os
import re

import numpy as np
import pytest

import atlas_amd
from atlas_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "atlas_amd", "csrc")


def _names_read_by_the_sources():
    names = set()
    exp = os.path.join(ROOT, "tools", "experiments")   # the kernels of the experiments build read (dev) switches too
    for p in glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.cpp")) + glob.glob(os.path.join(CSRC, "*.h")) + \
            glob.glob(os.path.join(exp, "*.inc")) + glob.glob(os.path.join(exp, "*.hip")) + glob.glob(os.path.join(exp, "*.h")):
        text = open(p).read()
        if os.path.basename(p) != "env.cpp":
            # nothing but env.cpp may call getenv for one of our names
            assert not re.search(r'(?<!env_)getenv\("ATLAS_AMD_', text), p
        names |= set(re.findall(r'env_get\("(ATLAS_AMD_\w+)"\)', text))
    return names


def test_every_switch_the_sources_read_is_in_the_table_and_in_the_document():
    cfg = _lib.effective_config()
    read = _names_read_by_the_sources()
    assert read and read <= set(cfg), sorted(read - set(cfg))
    assert set(cfg) - read <= {"ATLAS_AMD_IGNORE_ENV"}, sorted(set(cfg) - read)      # no dead rows either
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    missing = [n for n in cfg if f"`{n}`" not in doc]
    assert not missing, f"INTEGRATION.md section 8 lacks {missing}: regenerate with tools/gen_env_table.py"
    assert all(v["class"] in ("tuning", "behaviour", "test hook", "dev") and v["what"] for v in cfg.values())


def test_ignore_env_makes_every_switch_read_as_unset(monkeypatch):
    monkeypatch.setenv("ATLAS_AMD_FFT_STREAMS", "7")
    monkeypatch.setenv("ATLAS_AMD_LEG_KERNEL", "classic")
    cfg = _lib.effective_config()
    assert cfg["ATLAS_AMD_FFT_STREAMS"]["value"] == "7" and cfg["ATLAS_AMD_FFT_STREAMS"]["source"] == "env"
    try:
        _lib.set_ignore_env(1)
        cfg = _lib.effective_config()
        assert cfg["ATLAS_AMD_FFT_STREAMS"]["source"] == "ignored" and cfg["ATLAS_AMD_FFT_STREAMS"]["value"] == cfg["ATLAS_AMD_FFT_STREAMS"]["default"]
        assert cfg["ATLAS_AMD_LEG_KERNEL"]["value"] == "lean"
    finally:
        _lib.set_ignore_env(0)
    monkeypatch.setenv("ATLAS_AMD_IGNORE_ENV", "1")
    assert _lib.effective_config()["ATLAS_AMD_LEG_KERNEL"]["source"] == "ignored"
    monkeypatch.setenv("ATLAS_AMD_IGNORE_ENV", "0")
    assert _lib.effective_config()["ATLAS_AMD_LEG_KERNEL"]["source"] == "env"


HOSTILE = {"ATLAS_AMD_LEG_KERNEL": "classic", "ATLAS_AMD_LEG_CFG": "1,1", "ATLAS_AMD_FFT_GENERIC": "1", "ATLAS_AMD_FFT_STREAMS": "1",
           "ATLAS_AMD_FFT_PREFETCH": "0", "ATLAS_AMD_FFT_FAST_M": "1", "ATLAS_AMD_FFT_FINER_M": "0", "ATLAS_AMD_FFT_COARSE": "0",
           "ATLAS_AMD_FFT_COARSE_FUSED": "0", "ATLAS_AMD_FFT_COARSE_MULTI": "0", "ATLAS_AMD_FFT_NT_DIV": "4", "ATLAS_AMD_FFT_ROW_AFFINITY": "0",
           "ATLAS_AMD_FFT_SMOOTH_DIRECT": "1", "ATLAS_AMD_PREPARE": "rows", "ATLAS_AMD_PIPELINE": "3", "ATLAS_AMD_TABLES": "host",
           "ATLAS_AMD_HOST_PIPELINE": "0", "ATLAS_AMD_FFT_NATIVE": "1", "ATLAS_AMD_FFT_HYBRID": "1", "ATLAS_AMD_FFT_ONLY_M": "4096",
           "ATLAS_AMD_FFT_ABLATE": "64", "ATLAS_AMD_FFT_DEBUG": "1", "ATLAS_AMD_DIST_POISON": "1", "ATLAS_AMD_HOST_PIPELINE_FAIL_ALLOC": "1",
           "ATLAS_AMD_FFT_LDS_ELEMS": "600"}


@pytest.mark.gpu
def test_a_hostile_environment_does_not_change_the_plan_or_the_bits_once_ignored(monkeypatch):
    """the adapter's situation: the process environment sets every switch that selects a kernel to a non-default value; with
    atlas_amd__set_ignore_env(1) (what adapter/Library.cc calls) the launch plan and the results are those of a clean environment"""
    torch = pytest.importorskip("torch")
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import red_spectra
    g, T, nf = atlas_amd.Grid("O160"), 159, 9
    sp = torch.from_numpy(red_spectra(T, nf, seed=77)).cuda()

    def run():
        tr = atlas_amd.Trans(g, T)
        gp = torch.zeros(nf * g.size(), dtype=torch.float64, device="cuda")
        tr.invtrans(nf, sp, gp)
        tr.synchronize()
        return tr.fourier_launch_plan(), gp.cpu().numpy()

    plan0, gp0 = run()
    for k2, v in HOSTILE.items():
        monkeypatch.setenv(k2, v)
    try:
        _lib.set_ignore_env(1)
        cfg = _lib.effective_config()
        assert all(v["source"] in ("default", "ignored", "compiled out") for k2, v in cfg.items() if k2 != "ATLAS_AMD_IGNORE_ENV")
        plan1, gp1 = run()
    finally:
        _lib.set_ignore_env(0)
    assert plan1 == plan0 and np.array_equal(gp1, gp0)
    # and without the guard the same environment does select other kernels (the switches are live): another launch plan
    monkeypatch.delenv("ATLAS_AMD_HOST_PIPELINE_FAIL_ALLOC")
    plan2, gp2 = run()
    assert plan2 != plan0
    assert np.abs(gp2 - gp0).max() < 1e-11 * np.abs(gp0).max()
