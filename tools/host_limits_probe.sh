#!/bin/bash
# What the GPU box's host gives this container: CPUs, the cgroup's CPU quota and how often it throttled a plumbing leg, memory, /dev/shm.
# Usage (GPU box): bash tools/host_limits_probe.sh > gpurun_out/<dir>/host_limits.txt
echo "nproc: $(nproc)   affinity: $(python -c 'import os; print(len(os.sched_getaffinity(0)))')"
echo "cpu.max: $(cat /sys/fs/cgroup/cpu.max 2>/dev/null || echo n/a)   cfs_quota_us: $(cat /sys/fs/cgroup/cpu/cpu.cfs_quota_us 2>/dev/null || echo n/a) period $(cat /sys/fs/cgroup/cpu/cpu.cfs_period_us 2>/dev/null || echo n/a)"
echo "memory.max: $(cat /sys/fs/cgroup/memory.max 2>/dev/null || cat /sys/fs/cgroup/memory/memory.limit_in_bytes 2>/dev/null || echo n/a)"
echo "cpuset: $(cat /sys/fs/cgroup/cpuset.cpus.effective 2>/dev/null || echo n/a)"
df -h /dev/shm | tail -1
lscpu | grep -E "Model name|Socket|NUMA node|Thread|Core" 
echo "loadavg: $(cat /proc/loadavg)"
stat() { cat /sys/fs/cgroup/cpu.stat 2>/dev/null | tr '\n' ' '; echo; }
python - <<'PY'
import sys; sys.path.insert(0,'tools'); import plumbing as P
P.write_corpus('/dev/shm/probe_wav', 64)
PY
for w in 8 16 32 64; do
  echo "--- leg D, $w workers"; echo "before: $(stat)"
  HIPFEAT_NO_FORK_WARNING=1 python tools/plumbing.py --leg D --wav-dir /dev/shm/probe_wav --repeat 200 --workers $w --passes 1 2>/dev/null | python -c "
import sys,json
r=json.loads(sys.stdin.readline()); print(r['cuts_per_s'], {k:v for k,v in r.items() if k.endswith('share')})"
  echo "after:  $(stat)   loadavg: $(cat /proc/loadavg)"
done
rm -rf /dev/shm/probe_wav
