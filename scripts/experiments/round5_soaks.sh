mkdir -p gpurun_out
{
echo "round 5 soaks on the build with host-decided offset ties, the gather push and the batched compaction (MI355X)"
echo '$ python scripts/ber_curve.py'
timeout -k 10 300 python scripts/ber_curve.py > gpurun_out/r05_ber_curve.json 2> gpurun_out/r05_ber.err && python - <<'PY'
import json
a=json.load(open('gpurun_out/r05_ber_curve.json'))['rows']; b=json.load(open('profiles/r04_ber_curve.json'))['rows']
print("rows identical to profiles/r04_ber_curve.json:", a==b, "(%d rows, %d frames)"%(len(a), sum(r['frames_sent'] for r in a)))
PY
echo '$ python scripts/experiments/offset_soak.py 8 11'
timeout -k 10 400 python scripts/experiments/offset_soak.py 8 11 2>&1 | tail -3
echo '$ OPV_FUZZ_SEEDS=24 OPV_FUZZ_BASE=55500 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "fuzz or random_call_sequences or channel_accidents"'
OPV_SKIP_RCCL_SELFTEST=1 OPV_FUZZ_SEEDS=24 OPV_FUZZ_BASE=55500 timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "fuzz or random_call_sequences or channel_accidents" 2>&1 | tail -2
echo '$ python scripts/experiments/leak_probe.py'
timeout -k 10 200 python scripts/experiments/leak_probe.py 2>&1 | tail -2
} > gpurun_out/r05_soaks.txt 2>&1
tail -20 gpurun_out/r05_soaks.txt
