#!/bin/bash
# Round 6 (VERDICT r5 task 4): ONE same-call A/B on fft512c<13,12,2,FLAT> (MFCC 40 x 40, LibriSpeech-like lengths):
# product = DCT operands resident (40 VGPRs), split-step twiddles read from LDS every round;
# variant (-DHIPFEAT_ABL_MFCC_TWS) = split-step twiddles resident (16 VGPRs), last 4 of 10 DCT operand chunks re-read from memory every round.
# Keep iff >= +3 % and bit-identical.  Build the variant first:  python tools/variants.py mfcc_tws:"-DHIPFEAT_ABL_MFCC_TWS"
set -u
OUT=${1:-gpurun_out/r6_mfcc_ab}
mkdir -p "$OUT"
VAR=$PWD/lhotse_amd/_lib/var_mfcc_tws.so
for rep in 1 2 3; do
  for which in product variant; do
    if [ $which = variant ]; then export HIPFEAT_LIB=$VAR; else unset HIPFEAT_LIB; fi
    python bench.py --config mfcc40_libri --steps 100 --no-cpu-baseline --no-extra > "$OUT/${which}_$rep.json" 2> "$OUT/${which}_$rep.err"
    python - "$OUT/${which}_$rep.json" $which $rep <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], sys.argv[3], d["value"], "cuts/s", d["roofline"]["frac"], "launch_ms", d["roofline"].get("launch_ms"), "parity", d["parity"]["pass"], d["config"]["kernel"].split(" ")[0])
PY
  done
done
unset HIPFEAT_LIB
# bit-identity: the same ragged batch through both libraries
for which in product variant; do
  if [ $which = variant ]; then export HIPFEAT_LIB=$VAR; else unset HIPFEAT_LIB; fi
  python - <<'PY'
import hashlib, numpy as np, torch, lhotse_amd
rs = np.random.RandomState(3)
ws = [torch.from_numpy((rs.rand(int(n)).astype(np.float32) - 0.5)) for n in rs.randint(16000, 400000, size=200)]
ex = lhotse_amd.HipMfcc(lhotse_amd.HipMfccConfig(num_filters=40, num_ceps=40))
outs = ex.extract_batch(ws, 16000)
h = hashlib.sha256()
for o in outs:
    h.update(np.ascontiguousarray(o.cpu().numpy() if hasattr(o, "cpu") else o).tobytes())
print("sha256 of 200 MFCC matrices:", h.hexdigest()[:32], ex.kernel_name.split(" ")[0])
PY
done
