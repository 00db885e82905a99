#!/bin/bash
# dev: the 1024-stream question - placement, in-kernel clock and wave cycles of the one-wave front-end
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:?run on the GPU box through gpurun}; O=$R/gpurun_out/anomaly; rm -rf $O; mkdir -p $O
P=$R/scripts/experiments/placement_probe.py
for S in 64 256 512 1024 2048 4096; do
  timeout 200 python3 $P $S 30 3 2>&1 | grep -v amdgpu.ids | tee -a $O/probe.txt
done
for S in 256 1024 2048; do
  timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-include-regex k_msk_frontend --output-format csv -d $O/g$S -- python3 $P $S 30 2 > $O/g$S.log 2>&1
  timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU --kernel-include-regex k_msk_frontend --output-format csv -d $O/s$S -- python3 $P $S 30 2 > $O/s$S.log 2>&1
  timeout 300 rocprofv3 --pmc SQ_IFETCH SQ_WAIT_IFETCH SQ_INSTS_VALU SQ_INST_LEVEL_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_SALU --kernel-include-regex k_msk_frontend --output-format csv -d $O/i$S -- python3 $P $S 30 2 > $O/i$S.log 2>&1
done
for f in $(find $O -name "*counter_collection.csv" | sort); do echo == $f; python3 - "$f" <<'PY'
import csv,sys,collections
agg=collections.defaultdict(float); n=collections.defaultdict(int)
for r in csv.DictReader(open(sys.argv[1])):
    k=(r.get('Kernel_Name','')[:24], r['Counter_Name']); agg[k]+=float(r['Counter_Value']); n[k]+=1
for k,v in sorted(agg.items()): print("%-26s %-22s per_dispatch=%.6g dispatches=%d"%(k[0],k[1],v/n[k],n[k]))
PY
done | tee $O/pmc.txt
rocprofv3 -L 2>/dev/null | grep -E "SQ_.*(IFETCH|INST_LEVEL|WAIT)" | head -40 > $O/counters.txt
