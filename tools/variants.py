#!/usr/bin/env python3
"""Build experiment variants of libhipfeat.so (different flags / macros) into lhotse_amd/_lib/var_<name>.so
and print the shell loop that benches them on the GPU box.

    python tools/variants.py name1:"-DX=1 -fno-slp-vectorize" name2:"..."
"""
import os
import shlex
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lhotse_amd import build as B

names = []
for spec in sys.argv[1:]:
    name, _, flags = spec.partition(":")
    out = B.LIB_DIR / f"var_{name}.so"
    B.build(force=True, extra_flags=shlex.split(flags), output=out, verbose=False)
    names.append(name)
    print("built", out, flags, file=sys.stderr)
print(" ".join(names))
