#!/bin/bash
# librosa default: fixed-schedule PLAIN instance vs generic (same call); 48 kHz phase split
set -u
OUT=gpurun_out/${1:-r4_run6}
mkdir -p "$OUT"
timeout 600 python -m pytest tests/test_gpu_fixed_schedule.py tests/test_gpu_librosa.py -x -q 2>&1 | tail -3 | tee "$OUT/pytest.txt"
for i in 1 2; do
python tools/bench_librosa.py --cuts 4000 --steps 20 2>/dev/null | tee -a "$OUT/librosa.txt"
HIPFEAT_NO_FIXED_SCHEDULE=1 python tools/bench_librosa.py --cuts 4000 --steps 20 2>/dev/null | tee -a "$OUT/librosa.txt"
done
python tools/bench_rates.py --cuts 4000 --rates 22050,24000,44100,48000 2>/dev/null | tee "$OUT/rates.txt"
HIPFEAT_LIB=lhotse_amd/_lib/var_timers.so python tools/phase_timers_w.py 2000 48000 2>/dev/null | tee "$OUT/timers_48k.txt"
HIPFEAT_LIB=lhotse_amd/_lib/var_timers.so python tools/phase_timers_w.py 2000 24000 2>/dev/null | tee "$OUT/timers_24k.txt"
