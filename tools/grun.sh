#!/bin/bash
# gpurun wrapper for the build container: retries ONLY when no GPU slot / box was free (exit code 3: nothing ran, nothing
# was charged); any other outcome -- success, a failing command, a refusal -- is returned as is.   tools/grun.sh <timeout> '<command>'
t=$1; shift
for i in $(seq 1 30); do
  /usr/local/graft/bin/gpurun --timeout "$t" -- "$@"
  rc=$?
  [ $rc -ne 3 ] && exit $rc
  sleep 90
done
exit 3
