#!/bin/bash
# gpurun with a patient retry when the pool has no free slot (exit code 3 = nothing ran, nothing charged).
#   tools/gpurun_retry.sh <log> <timeout> '<command>'
LOG=$1; TMO=$2; shift 2
for i in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout $TMO -- "$@" > $LOG 2>&1
  rc=$?
  [ $rc -ne 3 ] && exit $rc
  sleep 90
done
exit 3
