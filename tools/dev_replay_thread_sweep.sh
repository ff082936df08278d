# replay throughput against host threads / hardware queues / device-resident frame-to-frame state of the native estimator (developer tool, GPU box)
for D in ${DEVSTATE:-0 1}; do for Q in ${QUEUES:-4}; do for T in ${THREADS:-1 2 4 8}; do
if [ $D = 0 ]; then export TCV_EST_HOST_STATE=1; else unset TCV_EST_HOST_STATE; fi
GPU_MAX_HW_QUEUES=$Q python bench.py --mode replay --steps 120 --warmup 10 --host-threads $T --no-cpu-baseline 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('device state $D queues $Q threads $T', round(d['value']), d['native_profile_ms_per_call'])"
done; done; done
