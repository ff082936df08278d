#!/bin/bash
O=gpurun_out/r05y; mkdir -p $O
for rep in 1 2 3; do
python -X faulthandler bench.py --mode replay --steps 60 --warmup 10 --streams 8 --no-cpu-baseline > $O/out_$rep.json 2> $O/err_$rep.txt; echo "rep $rep rc $?" >> $O/summary.txt
grep -v "^Extension modules" $O/err_$rep.txt | head -60 | cut -c1-220 >> $O/summary.txt
done
cat $O/summary.txt
