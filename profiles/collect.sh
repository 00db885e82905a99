#!/bin/bash
# profiles/collect.sh <round-tag> — run on the GPU box (through gpurun). Produces, for the
# default `python3 bench.py` command (the configuration the driver records):
#   profiles/<tag>_kernel_stats.csv     rocprofv3 --kernel-trace --stats summary
#   profiles/<tag>_pmc_*.txt            per-kernel PMC sums (separate passes; HBM bytes per
#                                       MI355X_MICROARCH.md §HBM: FETCH_SIZE x2 for wide reads)
#   profiles/<tag>_traffic.json         HBM bytes per launch + issued instructions per symbol of k_msk_frontend,
#                                       derived from those passes (read by bench.py for roofline.traffic / .issue)
#   profiles/<tag>_bench.json           the bench line of the un-profiled run (made last, so that it carries this collection's counters)
# Everything is written under gpurun_out/profiles_<tag>/ and merged back by gpurun; copy the
# summaries into profiles/ afterwards (see profiles/README.md).
TAG=${1:-r06}
R=${GRAFT_REPO_ROOT:?run on the GPU box through gpurun}
O=$R/gpurun_out/profiles_$TAG
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 $R/bench.py --no-extras > $O/kt.log 2>&1
cp $(find $O/kt -name "*kernel_stats.csv" | head -1) $O/${TAG}_kernel_stats.csv
agg() { python3 - "$1" <<'PY'
import csv,sys,collections
agg=collections.defaultdict(float); n=collections.defaultdict(int)
for r in csv.DictReader(open(sys.argv[1])):
    k=(r.get('Kernel_Name','')[:40], r['Counter_Name']); agg[k]+=float(r['Counter_Value']); n[k]+=1
for k,v in sorted(agg.items()): print("%-42s %-24s sum=%.6g dispatches=%d per_dispatch=%.6g"%(k[0],k[1],v,n[k],v/n[k]))
PY
}
i=0
for C in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 900 rocprofv3 --pmc $C --kernel-include-regex "k_msk_frontend|k_frame_decode|k_offset_search|k_sync_track" --output-format csv -d $O/p$i -- python3 $R/bench.py --no-extras --steps 1 --warmup 0 > $O/p$i.log 2>&1
  f=$(find $O/p$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && agg $f > $O/${TAG}_pmc_$i.txt
done
python3 - $O $TAG <<'PY'
import json, re, sys
O, TAG = sys.argv[1], sys.argv[2]
def val(i, kernel, counter):
    for ln in open(f"{O}/{TAG}_pmc_{i}.txt"):
        f = ln.split()
        if f and f[0] == kernel and f[1] == counter:
            return float(re.search(r"per_dispatch=(\S+)", ln).group(1))
    return None
S, F = 64, 1000
K = "k_msk_frontend_rb"            # the front-end kernel the shim launches for this workload (one wave per stream, row-broadcast reduction)
n_sym = S * 2168099.0                     # symbols one launch demodulates (86 724 000 samples per stream, ~40 per symbol)
fetch, write = val(1, K, "FETCH_SIZE"), val(2, K, "WRITE_SIZE")
ips = {k: round(val(3, K, c) / n_sym, 2) for k, c in (("valu", "SQ_INSTS_VALU"), ("salu", "SQ_INSTS_SALU"), ("lds", "SQ_INSTS_LDS"))}
out = {"kernel": K, "workload": {"streams_per_gpu": S, "frames_per_stream": F, "ebn0": 16.0, "f0_edge_hz": 2000.0},   # SURVEY.md 8(d) C4 as written (workload.py)
       "FETCH_SIZE_KB_per_launch": fetch, "WRITE_SIZE_KB_per_launch": write,
       "hbm_read_bytes_per_launch": fetch * 1024 * 2, "hbm_write_bytes_per_launch": write * 1024,
       "hbm_bytes_per_launch": fetch * 1024 * 2 + write * 1024,
       "instr_per_symbol": ips, "wave_cycles_per_symbol": round(val(3, K, "SQ_WAVE_CYCLES") * 4 / n_sym, 1),
       "correction": "gfx950: FETCH_SIZE counts wide (16 B/lane) coalesced reads at 1/2 -> x2 (MI355X_MICROARCH.md §HBM); WRITE_SIZE exact; units are KiB",
       "source": f"profiles/{TAG}_pmc_1.txt (FETCH_SIZE pass), profiles/{TAG}_pmc_2.txt (WRITE_SIZE pass), profiles/{TAG}_pmc_3.txt (SQ_INSTS_*, SQ_WAVE_CYCLES x 4), rocprofv3 --pmc, separate passes, command: python3 bench.py --no-extras --steps 1 --warmup 0"}
json.dump(out, open(f"{O}/{TAG}_traffic.json", "w"), indent=1)
print(json.dumps(out))
PY
# the un-profiled bench line LAST, with this collection's traffic / instruction counts in place (bench.py reads the newest
# profiles/rNN_traffic.json and attaches it when kernel and workload match)
cp $O/${TAG}_traffic.json $R/profiles/${TAG}_traffic.json
# (--extras-budget 600: every extra runs, extras.skipped_for_budget stays empty; the driver's own run keeps the 75 s default.
# The default run is recorded beside it: what the driver's record will look like, and how long it takes end to end.)
timeout 900 python3 $R/bench.py --extras-budget 600 2>$O/bench.err | tail -1 > $O/${TAG}_bench.json
( time timeout 600 python3 $R/bench.py --steps 20 --warmup 5 2>$O/bench_default.err | tail -1 > $O/${TAG}_bench_default_budget.json ) 2> $O/${TAG}_bench_default_budget.time
cat $O/${TAG}_bench.json | head -c 600; echo; head -8 $O/${TAG}_kernel_stats.csv; cat $O/${TAG}_pmc_*.txt
