#!/bin/bash
# round 5: where a lock-step frame of 128 streams goes on the host (TCV_DEBUG_EST laps), host threads 2 / 4, packer throughput on the box
O=gpurun_out/r05c; mkdir -p $O
python tools/dev_pack_bench.py 1 8 16 > $O/pack_bench.txt 2>&1
for T in 2 4; do
  python bench.py --mode replay --steps 40 --warmup 10 --streams 128 --host-threads $T > $O/replay_128_t$T.json 2> /dev/null
done
TCV_DEBUG_EST=1 TCV_DEBUG_PACK=1 python bench.py --mode replay --steps 20 --warmup 10 --streams 128 > /dev/null 2> $O/est.err
grep "^\[est\]\|batch_create" $O/est.err | tail -24 > $O/est_laps.txt; rm -f $O/est.err
python bench.py --mode replay --steps 60 --warmup 10 --streams 8 > $O/replay_8.json 2> /dev/null
python bench.py --mode replay --steps 60 --warmup 10 --streams 16 > $O/replay_16.json 2> /dev/null
cat $O/pack_bench.txt
