mkdir -p gpurun_out
{
echo "round 5 soaks, part 3 (HEAD: two-level tie guard, priority copy stream; MI355X)"
echo '$ OPV_FRONTEND=16 OPV_FUZZ_SEEDS=24 OPV_FUZZ_BASE=91000 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "fuzz or random_call_sequences or channel_accidents or push"'
OPV_SKIP_RCCL_SELFTEST=1 OPV_FRONTEND=16 OPV_FUZZ_SEEDS=24 OPV_FUZZ_BASE=91000 timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "fuzz or random_call_sequences or channel_accidents or push" 2>&1 | tail -2
echo '$ OPV_FUZZ_SEEDS=40 OPV_FUZZ_BASE=424200 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "fuzz or random_call_sequences"'
OPV_SKIP_RCCL_SELFTEST=1 OPV_FUZZ_SEEDS=40 OPV_FUZZ_BASE=424200 timeout -k 10 700 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "fuzz or random_call_sequences" 2>&1 | tail -2
echo '$ python scripts/experiments/offset_soak.py 8 77'
timeout -k 10 400 python scripts/experiments/offset_soak.py 8 77 2>&1 | tail -2
} > gpurun_out/r05_soaks3.txt 2>&1
tail -12 gpurun_out/r05_soaks3.txt
