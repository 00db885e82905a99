#!/bin/bash
# dev helper: PMC passes on the front-end kernel (run through scripts/gpurun_retry.sh)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:?run on the GPU box through gpurun}
O=$R/gpurun_out/pmc
rm -rf $O; mkdir -p $O
ARGS="$R/bench.py --streams 64 --frames 20 --steps 1 --warmup 0 --no-extras"
rocprofv3 -L > $O/counters.txt 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM --kernel-include-regex k_msk_frontend --output-format csv -d $O/p1 -- python3 $ARGS > $O/p1.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU --kernel-include-regex k_msk_frontend --output-format csv -d $O/p2 -- python3 $ARGS > $O/p2.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SMEM SQ_LDS_BANK_CONFLICT --kernel-include-regex k_msk_frontend --output-format csv -d $O/p3 -- python3 $ARGS > $O/p3.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 $ARGS > $O/kt.log 2>&1
find $O -name "*.csv" | head -20
for f in $(find $O -name "*counter_collection.csv"); do echo == $f; python3 - "$f" <<'PY'
import csv,sys,collections
agg=collections.defaultdict(float)
for r in csv.DictReader(open(sys.argv[1])):
    agg[(r.get('Kernel_Name','')[:30], r['Counter_Name'])]+=float(r['Counter_Value'])
for k,v in sorted(agg.items()): print(k, v)
PY
done
for f in $(find $O/kt -name "*kernel_stats.csv"); do echo == $f; head -12 $f; done
tail -3 $O/p1.log
