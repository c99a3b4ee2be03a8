#!/bin/bash
# Local helper: gpurun with retries while the pod has no free GPU slot (exit code 3: nothing charged).
# Usage: scripts/gpurun_retry.sh <timeout-seconds> '<command>' <logfile>
T=$1; CMD=$2; LOG=$3
for i in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout $T -- "$CMD" > $LOG 2>&1; rc=$?
  [ $rc -ne 3 ] && exit $rc
  sleep 90
done
exit 3
