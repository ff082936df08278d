#!/bin/bash
O=gpurun_out/r05z33; mkdir -p $O
RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.tsan-x86_64.so | tail -1)
for cfg in "32 2 12" "8 2 12" "24 3 10"; do
  set -- $cfg
  LD_PRELOAD=$RT TCV_LIB=tc-viml_amd/libtcv_hip_tsan.so TSAN_OPTIONS="halt_on_error=0:report_signal_unsafe=0:exitcode=0" timeout 600 python tests/dev/tsan_replay_drive.py $1 $2 $3 > $O/tsan_$1_$2.txt 2>&1; echo "rc $?" >> $O/tsan_$1_$2.txt
  echo "== $cfg"; grep -E "windows optimised|^rc|SUMMARY" $O/tsan_$1_$2.txt | sort | uniq -c | head -20
done
