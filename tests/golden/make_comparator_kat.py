"""Extracts the known-answer table of the reference's cseq_comparator unit test
(src/unit_tests/cseq_comparator_test.cpp) into tests/golden/cseq_comparator_kat.json:
the test sequences, and per check the comparator settings, the two sequences and the expected
value as the float the C++ expression evaluates to.  Data only; run in the build container."""
import json
import re
import sys

import numpy as np

src = open(sys.argv[1] if len(sys.argv) > 1 else "/root/reference/src/unit_tests/cseq_comparator_test.cpp").read()
seqs = {m.group(1): m.group(2) for m in re.finditer(r'cseq (c\d+)\s*\("",\s*"([^"]*)"\);', src)}
checks = []
for case in re.finditer(r'CASE\((\w+)\)\s*\{(.*?)\n\}', src, re.S):
    body = case.group(2)
    comps = {}
    for m in re.finditer(r'cseq_comparator (\w+)\(\s*CMP_IUPAC_(\w+),\s*CMP_DIST_(\w+),\s*CMP_COVER_(\w+),\s*(true|false)\)',
                         body):
        comps[m.group(1)] = dict(iupac=m.group(2).lower(), dist=m.group(3).lower(), cover=m.group(4).lower(),
                                 filter_lc=m.group(5) == "true")
    for m in re.finditer(r'^\s*EQUAL\((\w+)\((c\d+),\s*(c\d+)\),\s*(.+?)\);', body, re.M):
        name, a, b, expr = m.groups()
        expr = expr.strip()
        item = dict(case=case.group(1), a=a, b=b, **comps[name])
        m2 = re.match(r'(\w+)\((c\d+),\s*(c\d+)\)$', expr)
        if m2:  # equality with another comparator call
            item["same_as"] = dict(a=m2.group(2), b=m2.group(3), **comps[m2.group(1)])
        else:   # C++ float arithmetic: int/float literal division
            num, _, den = expr.partition("/")
            f = lambda t: np.float32(float(t.rstrip("f")))
            val = f(num) / f(den) if den else f(num)
            item["expect_bits"] = int(np.float32(val).view(np.uint32))
            item["expect"] = float(val)
        checks.append(item)
json.dump(dict(source="src/unit_tests/cseq_comparator_test.cpp", sequences=seqs, checks=checks),
          open("tests/golden/cseq_comparator_kat.json", "w"), indent=1)
print(len(seqs), "sequences", len(checks), "checks")
