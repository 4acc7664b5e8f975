"""Extracts the known-answer vectors of the reference's gaunt_test (inputs and expected values only) into
tests/golden/gaunt_known_answers.json.  Run in the build container where /root/reference exists:
    python tests/golden/make_gaunt_golden.py
Source of the numbers: /root/reference/src/general/gaunt_test.cpp:6-1655 (275 gaunt_coefficient and 275
modified_gaunt_coefficient values, 17 significant digits, tolerance DBL_EPSILON*(1+|ref|) in the reference)."""
import json
import os
import re

SRC = "/root/reference/src/general/gaunt_test.cpp"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "gaunt_known_answers.json")

call = re.compile(r"val=helfem::gaunt::(gaunt_coefficient|modified_gaunt_coefficient)\(([-0-9,]+)\);")
ref = re.compile(r"ref=([-+0-9.eE]+);")
entries = []
pending = None
for line in open(SRC):
    m = call.search(line)
    if m:
        pending = (m.group(1), [int(x) for x in m.group(2).split(",")])
        continue
    m = ref.search(line)
    if m and pending:
        entries.append({"fn": pending[0], "args": pending[1], "ref": float(m.group(1)), "ref_text": m.group(1)})
        pending = None
json.dump({"source": "src/general/gaunt_test.cpp:6-1655", "entries": entries}, open(OUT, "w"), indent=0)
print(len(entries), "entries ->", OUT)
