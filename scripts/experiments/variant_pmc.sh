# dev: wait/active counters of the front-end kernel for library variants
R=${GRAFT_REPO_ROOT:?run on the GPU box through gpurun}
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  export OPV_LIB=$R/opv-cxx-demod_amd/libopv_$v.so
  i=0
  for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_WR" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_IFETCH SQ_INSTS_BRANCH SQ_INSTS_CBRANCH_TAKEN"; do
    i=$((i+1))
    O=$R/gpurun_out/abpmc_${v}_$i
    rm -rf $O
    timeout -k 10 300 rocprofv3 --pmc $C --kernel-include-regex "k_msk_frontend" --output-format csv -d $O -- python3 $R/bench.py --no-extras --steps 1 --warmup 0 > $O.log 2>&1 || { tail -5 $O.log; exit 1; }
    f=$(find $O -name "*counter_collection.csv" | head -1)
    python3 - $f $v <<'PY'
import csv,sys,collections
agg=collections.defaultdict(float); n=collections.defaultdict(int)
for r in csv.DictReader(open(sys.argv[1])):
    k=r['Counter_Name']; agg[k]+=float(r['Counter_Value']); n[k]+=1
nsym=64*2168099.0
for k,v in sorted(agg.items()): print("%-8s %-24s per_symbol=%.3f"%(sys.argv[2],k,v/n[k]/nsym))
PY
    rm -rf $O
  done
done
