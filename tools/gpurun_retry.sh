#!/bin/bash
# usage: tools/gpurun_retry.sh <timeout> '<command>'  -- retries while the pool reports no free slot (rc 3)
T=$1; shift
for i in 1 2 3 4 5 6 7 8; do
  /usr/local/graft/bin/gpurun --timeout $T -- "$@" > /tmp/gpurun_last.txt 2>&1
  rc=$?
  if grep -q "status=transient" /tmp/gpurun_last.txt; then sleep 45; continue; fi
  break
done
cat /tmp/gpurun_last.txt
exit $rc
