#!/bin/bash
O=gpurun_out/r05z34; mkdir -p $O
RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.tsan-x86_64.so | tail -1)
LD_PRELOAD=$RT TCV_LIB=tc-viml_amd/libtcv_hip_tsan.so TSAN_OPTIONS="halt_on_error=0:report_signal_unsafe=0:exitcode=0" timeout 900 python tests/dev/tsan_stream_drive.py 256 > $O/tsan_stream.txt 2>&1; echo "rc $?" >> $O/tsan_stream.txt
grep -E "stream mode driven|^rc" $O/tsan_stream.txt; grep -c "WARNING: ThreadSanitizer" $O/tsan_stream.txt
python3 - <<'PY'
import re
txt=open('gpurun_out/r05z34/tsan_stream.txt').read()
reps=txt.split('WARNING: ThreadSanitizer')[1:]
ours=0
for r in reps:
    tops=[]
    for b in re.split(r'\n\n', r):
        m=re.match(r'\s*(Previous )?(atomic )?(read|write|Read|Write|Atomic read|Atomic write)[^\n]*\n\s*#0 ([^\n]*)', b.strip('\n'))
        if m: tops.append(m.group(4))
    if tops and not any(('libamdhip64' in t or 'libhsa' in t or 'libclang_rt' in t) for t in tops):
        ours+=1; print('OURS:', r.split('\n')[0].strip(), tops[:2])
print('reports', len(reps), 'with both accesses outside the HIP runtime:', ours)
PY
