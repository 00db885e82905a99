#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/x4pmc; rm -rf $O; mkdir -p $O
S=${1:-512}; F=${2:-30}
for spw in 1 4; do
  timeout 200 python3 $R/scripts/experiments/x4_probe.py $S $F $spw
  timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM --kernel-include-regex k_msk_frontend --output-format csv -d $O/a$spw -- python3 $R/scripts/experiments/x4_probe.py $S $F $spw > $O/a$spw.log 2>&1
  timeout 300 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA --kernel-include-regex k_msk_frontend --output-format csv -d $O/b$spw -- python3 $R/scripts/experiments/x4_probe.py $S $F $spw > $O/b$spw.log 2>&1
  timeout 300 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_LDS_BANK_CONFLICT SQ_INSTS_BRANCH --kernel-include-regex k_msk_frontend --output-format csv -d $O/c$spw -- python3 $R/scripts/experiments/x4_probe.py $S $F $spw > $O/c$spw.log 2>&1
done
for f in $(find $O -name "*counter_collection.csv" | sort); do echo == $f; python3 - "$f" <<'PY'
import csv,sys,collections
agg=collections.defaultdict(float); n=collections.defaultdict(int)
for r in csv.DictReader(open(sys.argv[1])):
    k=(r.get('Kernel_Name','')[:24], r['Counter_Name']); agg[k]+=float(r['Counter_Value']); n[k]+=1
for k,v in sorted(agg.items()): print("%-26s %-22s per_dispatch=%.6g"%(k[0],k[1],v/n[k]))
PY
done
