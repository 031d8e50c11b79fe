#!/bin/bash
# Retry the gpurun CLIENT while it reports "no slot free" (exit 3: nothing ran, nothing charged).  Never retries a command that ran.
# usage: tools/gpurun_retry.sh TIMEOUT 'command'
for i in $(seq 1 40); do
    /usr/local/graft/bin/gpurun --timeout "$1" -- "$2"
    rc=$?
    if [ $rc -ne 3 ]; then exit $rc; fi
    sleep 45
done
exit 3
