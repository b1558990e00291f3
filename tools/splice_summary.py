#!/usr/bin/env python3
"""tools/splice_summary.py <prefix> : replaces the configuration table of DESIGN.md section 5 by the one of
profiles/<prefix>_summary.md (tools/profile_summary.py)."""
import os, sys
prefix = sys.argv[1] if len(sys.argv) > 1 else "r06"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tab = [l for l in open(os.path.join(root, "profiles", prefix + "_summary.md")).read().split("\n") if l.startswith("|")]
p = os.path.join(root, "DESIGN.md")
lines = open(p).read().split("\n")
i0 = next(i for i, l in enumerate(lines) if l.startswith("| configuration (1024 ch"))
i1 = i0
while i1 < len(lines) and lines[i1].startswith("|"):
    i1 += 1
lines[i0:i1] = tab
open(p, "w").write("\n".join(lines))
print("DESIGN.md: table of %d rows replaced by %d" % (i1 - i0 - 2, len(tab) - 2))
