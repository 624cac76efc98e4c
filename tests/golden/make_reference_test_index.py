#!/usr/bin/env python3
"""Index of the reference's own tests: file, first line, last line and name of every `test "..."` block under
/root/reference/src (140 of them).  Data for tests/test_reference_test_index.py, which holds every block to at least one
known-answer vector of oracle/kat_main.cpp or tests/cpp/host_kat_main.cpp that cites a line inside it.

    python tests/golden/make_reference_test_index.py      (here, where /root/reference exists) -> tests/golden/reference_tests.json"""
import json, os, re
REF = "/root/reference/src"
rows = []
for root, _, files in os.walk(REF):
    for f in sorted(files):
        if not f.endswith(".zig"):
            continue
        lines = open(os.path.join(root, f)).read().splitlines()
        for i, line in enumerate(lines):
            m = re.match(r'test "(.*)" \{', line)
            if not m:
                continue
            j = i + 1
            while j < len(lines) and not lines[j].startswith("}"):
                j += 1
            rows.append({"file": f, "first_line": i + 1, "last_line": j + 1, "name": m.group(1)})
rows.sort(key=lambda r: (r["file"], r["first_line"]))
out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_tests.json")
json.dump(rows, open(out, "w"), indent=0)
print(len(rows), "tests ->", out)
