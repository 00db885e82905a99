#!/bin/bash
# dev: HBM bytes of the sixteen-streams-per-wave front-end on 32 768 independent 8-frame captures (91.5 GB of IQ): FETCH_SIZE and
# WRITE_SIZE in separate passes (KiB; FETCH_SIZE x2 on gfx950 for 16 B/lane reads, MI355X_MICROARCH.md §HBM)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:?run on the GPU box through gpurun}; O=$R/gpurun_out/x16traffic; rm -rf $O; mkdir -p $O
for C in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  T=$(echo $C | cut -d' ' -f1)
  timeout 500 rocprofv3 --pmc $C --kernel-include-regex "k_msk_frontend_x16" --output-format csv -d $O/$T -- python3 $R/scripts/experiments/many_unique.py 32768 8 > $O/$T.log 2>&1
  grep "^S=" $O/$T.log
  f=$(find $O/$T -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 - "$f" <<'PY'
import csv,sys,collections
agg=collections.defaultdict(float); n=collections.defaultdict(int)
for r in csv.DictReader(open(sys.argv[1])):
    k=(r.get('Kernel_Name','')[:28], r['Counter_Name']); agg[k]+=float(r['Counter_Value']); n[k]+=1
for k,v in sorted(agg.items()): print("%-30s %-16s per_dispatch=%.6g dispatches=%d"%(k[0],k[1],v/n[k],n[k]))
PY
done
