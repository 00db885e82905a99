#!/bin/bash
# usage: gpurun_retry.sh <outfile> <timeout> <cmd...>  — retries while the pod's GPU slots are busy
out=$1; shift; to=$1; shift
for i in $(seq 1 30); do
  /usr/local/graft/bin/gpurun --timeout $to -- "$@" > $out 2>&1
  if ! grep -q "status=transient" $out; then exit 0; fi
  sleep 90
done
