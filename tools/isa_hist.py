#!/usr/bin/env python3
"""Instruction histogram of one kernel's steady-state loop from the -S output of hipfeat.hip (no GPU needed).

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -DHIPFEAT_BUILD -Iinclude --cuda-device-only -S lhotse_amd/csrc/hipfeat.hip -o /tmp/hf.s
    python tools/isa_hist.py /tmp/hf.s _ZN7hipfeat14fft512c_kernelILi13ELi12ELi0EEEvNS_13Fft512cParamsE [frames per loop trip per wave]
"""
import collections, re, sys

path, name = sys.argv[1], sys.argv[2]
frames = float(sys.argv[3]) if len(sys.argv) > 3 else 4.0
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith(name + ":"))
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
body = lines[start:end]
labels = {}
for i, l in enumerate(body):
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m:
        labels[m.group(1)] = i
loops = []
for i, l in enumerate(body):
    m = re.search(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
    if m and m.group(1) in labels and labels[m.group(1)] < i:
        loops.append((i - labels[m.group(1)], labels[m.group(1)], i))
loops.sort(reverse=True)
n, a, b = loops[0]
cnt = collections.Counter()
for l in body[a:b]:
    l = l.strip()
    if not l or l[0] in ".;" or l.endswith(":"):
        continue
    cnt[l.split()[0]] += 1
cls = lambda p: sum(v for k, v in cnt.items() if p(k))
valu = cls(lambda k: k.startswith("v_") and not k.startswith("v_mfma"))
print(f"loop of {n} lines: total {sum(cnt.values())}  VALU {valu} ({valu / frames:.1f}/frame)  MFMA {cls(lambda k: k.startswith('v_mfma'))}  "
      f"DS {cls(lambda k: k.startswith('ds_'))}  VMEM {cls(lambda k: k.startswith(('global_', 'buffer_')))}  SALU {cls(lambda k: k.startswith('s_'))}")
for k, v in cnt.most_common(40):
    print(f"  {k:30s}{v}")
