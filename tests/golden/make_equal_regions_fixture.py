"""Writes tests/golden/equal_regions_O8_5.json from the expected array the reference's own test holds
(src/tests/functionspace/test_structuredcolumns.cc:87-106: partition of every point of O8 over 5 MPI tasks with the
default equal_regions partitioner, gathered on the root).  Run in the build container only (/root/reference)."""
import json
import os
import re

SRC = "/root/reference/src/tests/functionspace/test_structuredcolumns.cc"
text = open(SRC).read()
m = re.search(r"std::vector<double> check\{(.*?)\};", text, re.S)
vals = [int(v) for v in re.findall(r"\d+", m.group(1))]
out = {"source": "src/tests/functionspace/test_structuredcolumns.cc:87-106", "grid": "O8", "nparts": 5, "partition": vals}
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "equal_regions_O8_5.json")
with open(path, "w") as f:
    json.dump(out, f)
print(len(vals), "values ->", path)
