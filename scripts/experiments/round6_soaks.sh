# round 6 soaks as run on the GPU box (through gpurun): regression evidence for what CHANGED this round - the offset search's
# power-scaled near-tie bands and the stream-ordered host decision behind it, and opv_process's round bookkeeping (the host's
# view of a round is committed only once the round will be launched). The front-end, tracker and decoder kernels are
# instruction for instruction those of round 5.   -> profiles/r06_soaks.txt
mkdir -p gpurun_out
{
echo "round 6 soaks (MI355X; final tree of the round: stream-ordered host decision of offset-search ties, power-scaled bands)"
echo '$ python scripts/ber_curve.py'
timeout -k 10 300 python scripts/ber_curve.py > gpurun_out/r06_ber_curve.json 2> gpurun_out/r06_ber.err && python - <<'PY'
import json
a=json.load(open('gpurun_out/r06_ber_curve.json'))['rows']; b=json.load(open('profiles/r05_ber_curve.json'))['rows']
print("rows identical to profiles/r05_ber_curve.json:", a==b, "(%d rows, %d frames)"%(len(a), sum(r['frames_sent'] for r in a)))
PY
echo '$ python scripts/experiments/offset_soak.py 8 61      (4096 openings, tie decision on the host, stream-ordered)'
timeout -k 10 400 python scripts/experiments/offset_soak.py 8 61 2>&1 | tail -3
echo '$ OPV_OFFSET_DISTRUST_LIBM=1 python scripts/experiments/offset_soak.py 4 62      (2048 openings, the device-sincos fallback)'
OPV_OFFSET_DISTRUST_LIBM=1 timeout -k 10 300 python scripts/experiments/offset_soak.py 4 62 2>&1 | tail -3
echo '$ OPV_FUZZ_SEEDS=24 OPV_FUZZ_BASE=66600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "fuzz or random_call_sequences or channel_accidents or push"'
OPV_SKIP_RCCL_SELFTEST=1 OPV_FUZZ_SEEDS=24 OPV_FUZZ_BASE=66600 timeout -k 10 700 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "fuzz or random_call_sequences or channel_accidents or push" 2>&1 | grep -v "^ \|wall clock\|^session" | tail -2
echo '$ OPV_FRONTEND=16 OPV_FUZZ_SEEDS=12 OPV_FUZZ_BASE=67700 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "fuzz or random_call_sequences"'
OPV_SKIP_RCCL_SELFTEST=1 OPV_FRONTEND=16 OPV_FUZZ_SEEDS=12 OPV_FUZZ_BASE=67700 timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "fuzz or random_call_sequences" 2>&1 | grep -v "^ \|wall clock\|^session" | tail -2
echo '$ python scripts/experiments/many_streams.py 12000 3 65536 2'
timeout -k 10 300 python scripts/experiments/many_streams.py 12000 3 65536 2 2>&1 | grep "^S="
echo '$ python scripts/experiments/leak_probe.py'
timeout -k 10 200 python scripts/experiments/leak_probe.py 2>&1 | tail -2
} > gpurun_out/r06_soaks.txt 2>&1
tail -30 gpurun_out/r06_soaks.txt
