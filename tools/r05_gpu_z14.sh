#!/bin/bash
O=gpurun_out/r05z14; mkdir -p $O
python tc-viml_amd/build.py --profile > $O/build.log 2>&1; tail -2 $O/build.log
for B in 1 8; do TCV_LIB=tc-viml_amd/libtcv_hip_prof.so python tools/dev_phase_profile.py $B 256 --prior > $O/phase_B$B.txt 2>&1; cat $O/phase_B$B.txt | head -40; done
