#!/bin/bash
O=gpurun_out/r05v; mkdir -p $O
TCV_DEBUG_PIPE=1 python bench.py --mode replay --steps 12 --warmup 6 --streams 32 --host-threads 1 --no-cpu-baseline > /dev/null 2> $O/pipe.err
grep "\[pipe\]" $O/pipe.err | tail -40 > $O/pipe_1thread.txt
TCV_DEBUG_PIPE=1 python bench.py --mode replay --steps 12 --warmup 6 --streams 8 --host-threads 1 --no-cpu-baseline > /dev/null 2> $O/pipe8.err
grep "\[pipe\]" $O/pipe8.err | tail -24 > $O/pipe8_1thread.txt; grep -v "\[pipe\]" $O/pipe8.err | tail -12 >> $O/pipe8_1thread.txt
cat $O/pipe_1thread.txt; echo; cat $O/pipe8_1thread.txt
