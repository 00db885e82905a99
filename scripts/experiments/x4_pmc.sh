#!/bin/bash
# dev: instruction / wait counters of the four-streams-per-wave front-end (4096 and 8192 streams x 30 frames) and, for
# comparison, of the one-wave-per-stream one at 1024 streams. Per stream and symbol: counter / (streams x symbols).
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:?run on the GPU box through gpurun}; O=$R/gpurun_out/x4pmc; rm -rf $O; mkdir -p $O
P=$R/scripts/experiments/x4_probe.py
for CFG in "4096 4" "8192 4" "1024 1"; do
set -- $CFG
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY --kernel-include-regex "k_msk_frontend" --output-format csv -d $O/a$1 -- python3 $P $1 30 $2 > $O/a$1.log 2>&1
done
for f in $(find $O -name "*counter_collection.csv" | sort); do echo == $f; python3 - "$f" <<'PY'
import csv,sys,collections
agg=collections.defaultdict(float); n=collections.defaultdict(int)
for r in csv.DictReader(open(sys.argv[1])):
    k=(r.get('Kernel_Name','')[:24], r['Counter_Name']); agg[k]+=float(r['Counter_Value']); n[k]+=1
for k,v in sorted(agg.items()): print("%-26s %-22s per_dispatch=%.6g"%(k[0],k[1],v/n[k]))
PY
done
