"""Generates tests/golden/halo_adjoint_fixture.json from the reference's own test of the adjoint halo exchange
(src/tests/parallel/test_haloexchange_adjoint.cc): for every test function with explicit expected arrays, the three
per-rank input arrays and the three per-rank expected arrays (data only).  Run in the build container, where
/root/reference is mounted; the JSON is what travels.

    python tests/golden/make_halo_adjoint_fixture.py"""
import json
import os
import re

REF = "/root/reference/src/tests/parallel/test_haloexchange_adjoint.cc"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "halo_adjoint_fixture.json")

COMMENTS = {
    "rank0_arrview": "field(N)",
    "rank1": "field(N,2) = (10*b, 100*b)",
    "rank1_strided_v1": "shape (N,1) strides (2,1) on the (N,2) array: only component 0 takes part",
    "rank1_strided_v2": "same, starting at component 1",
    "rank2": "field(N,3,2) = (-b*10^i, +b*10^i)",
    "rank2_l1": "shape (N,1,2) strides (6,2,1) on the (N,3,2) array",
    "rank2_l2_v2": "shape (N,1,1) strides (6,2,1) starting at element (0,1,1)",
    "rank2_v2": "shape (N,3,1) strides (6,2,2) starting at element (0,0,1)",
    "rank0_wrap": "field(N) wrapped around existing memory; the inputs are zero on the halo, so nothing changes",
    "rank1_paralleldim1": "field(2,N), parallel dim = last",
    "rank2_paralleldim2": "field(3,N,2), parallel dim = 1",
    "rank1_cinterface": "field(N,2) through atlas__HaloExchange__execute_adjoint_strided_double",
}


def main():
    src = open(REF).read()
    lines = src.splitlines()
    starts = [(i, m.group(1)) for i, ln in enumerate(lines)
              for m in [re.match(r"void test_(\w+)\(Fixture& f\) \{", ln)] if m]
    cases = {}
    for k, (i0, name) in enumerate(starts):
        if name.endswith("_adj_test"):
            continue
        i1 = starts[k + 1][0] if k + 1 < len(starts) else len(lines)
        body = "\n".join(lines[i0:i1])
        body = re.sub(r"//[^\n]*", "", body)
        # entries are C constant expressions; one array of the reference (test_rank2_l2_v2, rank 1) lacks a comma
        # ("700 - 8"), which C evaluates to a single entry 692 and leaves the array one entry short: kept as it is, the
        # test skips that one array
        arrays = [[float(eval(" ".join(tok.split()), {"__builtins__": {}})) for tok in m.split(",") if tok.strip()]
                  for m in re.findall(r"POD arr_c\[\]\s*=\s*\{([^}]*)\}", body, flags=re.S)
                  if re.fullmatch(r"[-+\d\s.,]*", m)]
        assert len(arrays) == 6, (name, len(arrays))
        cases[name] = {"comment": f"test_{name} :{i0 + 1}-{i1}: {COMMENTS[name]}", "input": arrays[:3],
                       "expected": arrays[3:]}
    fix = {"source": "ecmwf/atlas 0.44.1 src/tests/parallel/test_haloexchange_adjoint.cc (input and expected arrays of "
                     "the execute_adjoint tests on the 3-rank fixture of tests/golden/halo_fixture.json; data only)",
           "cases": cases}
    with open(OUT, "w") as f:
        json.dump(fix, f, indent=1)
    print("wrote", OUT, "with", len(cases), "cases:", ", ".join(cases))


if __name__ == "__main__":
    main()
