#!/bin/bash
# usage: tools/gpurun_retry.sh TIMEOUT 'command'   (build container only)
# gpurun exits with 3 when no GPU slot is free (nothing charged): try again
# every minute for up to 40 minutes.
for i in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout "$1" -- "$2"
  rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  sleep 60
done
exit 3
