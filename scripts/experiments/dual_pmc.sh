#!/bin/bash
# dev: instruction / wait counters of the two-waves-per-stream front-end (64 streams x 30 frames)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:?run on the GPU box through gpurun}; O=$R/gpurun_out/dualpmc; rm -rf $O; mkdir -p $O
P=$R/scripts/experiments/dual_probe.py
for M in -2 1; do
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY --kernel-include-regex "k_msk_frontend" --output-format csv -d $O/a$M -- python3 $P 64 30 $M > $O/a$M.log 2>&1
timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_WAVE_CYCLES --kernel-include-regex "k_msk_frontend" --output-format csv -d $O/b$M -- python3 $P 64 30 $M > $O/b$M.log 2>&1
done
for f in $(find $O -name "*counter_collection.csv" | sort); do echo == $f; python3 - "$f" <<'PY'
import csv,sys,collections
agg=collections.defaultdict(float); n=collections.defaultdict(int)
for r in csv.DictReader(open(sys.argv[1])):
    k=(r.get('Kernel_Name','')[:24], r['Counter_Name']); agg[k]+=float(r['Counter_Value']); n[k]+=1
for k,v in sorted(agg.items()): print("%-26s %-22s per_dispatch=%.6g"%(k[0],k[1],v/n[k]))
PY
done
