"""Prints the table of INTEGRATION.md section 8 from the library's own table of environment switches (csrc/env.cpp through
atlas_amd__effective_config): python tools/gen_env_table.py > /tmp/env.md.  tests/test_env_switches.py checks that every name
is in the document."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from atlas_amd import _lib  # noqa: E402

cfg = _lib.effective_config()
order = {"behaviour": 0, "tuning": 1, "test hook": 2, "dev": 3}
print("| switch | class | default | what it selects |")
print("|---|---|---|---|")
for name, v in sorted(cfg.items(), key=lambda kv: (order[kv[1]["class"]], kv[0])):
    print(f"| `{name}` | {v['class']} | {v['default']} | {v['what']} |")
