#!/bin/bash
# usage: gpurun_retry.sh <outfile> <timeout> <cmd...>  — retries while the pod's GPU slots are busy.
# Exit status: gpurun's own on the run that was not transient; 75 (EX_TEMPFAIL) when every retry was transient.
out=$1; shift; to=$1; shift
for i in $(seq 1 30); do
  /usr/local/graft/bin/gpurun --timeout $to -- "$@" > $out 2>&1
  rc=$?
  if ! grep -q "status=transient" $out; then exit $rc; fi
  sleep 90
done
echo "gpurun_retry.sh: still transient after 30 attempts" >> $out
exit 75
