#!/bin/bash
O=gpurun_out/r05z8; mkdir -p $O
for S in 8 64 128; do for T in 2 4; do
  python bench.py --mode replay --steps 50 --warmup 8 --streams $S --host-threads $T --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); h=d['host_cpu']; print('$S streams, host threads $T: %6.0f windows/s  ms/frame %.2f; cores busy %.2f, cpu us/window %.0f, quota %s' % (d['value'], d['ms_per_step'], h['cores_busy_mean'], h['cpu_us_per_window'], h['cgroup_cpu_quota_cores']))"
done; done > $O/cpu.txt 2>&1
cat $O/cpu.txt
nproc; cat /sys/fs/cgroup/cpu.max; cat /sys/fs/cgroup/cpu.stat | head -8
lscpu | grep -i "numa\|L3\|Thread\|Socket\|Core" | head
python3 -c "import os; print(sorted(os.sched_getaffinity(0))[:40], len(os.sched_getaffinity(0)))"
