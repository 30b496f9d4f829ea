mkdir -p gpurun_out/r6_ring
python - <<'PY'
import sys; sys.path.insert(0,'tools'); import plumbing as P
P.write_corpus('/dev/shm/ring_wav', 64)
PY
for rep in 1 2; do
for leg in C D; do
  for w in 8 16 32; do
    for extra in "" "--pcm16 --half"; do
      HIPFEAT_NO_FORK_WARNING=1 python tools/plumbing.py --leg $leg --wav-dir /dev/shm/ring_wav --repeat 200 --workers $w --passes 1 $extra 2>/dev/null | python -c "
import sys,json
r=json.loads(sys.stdin.readline()); print('leg $leg workers $w', '$extra', r['cuts_per_s'], r['seconds_to_first_batch'], {k:v for k,v in r.items() if k.endswith('share')})"
    done
  done
done
done | tee gpurun_out/r6_ring/ab.txt
rm -rf /dev/shm/ring_wav
