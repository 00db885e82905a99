mkdir -p gpurun_out
{
echo "round 5 soaks, part 2 (final build: host-decided offset ties, gather push, batched compaction, asynchronous batches; MI355X)"
echo '$ python scripts/experiments/decoder_soak.py 40000 5'
timeout -k 10 300 python scripts/experiments/decoder_soak.py 40000 5 2>&1 | tail -1
echo '$ python scripts/experiments/many_streams.py 12000 3 20000 3 65536 2 100000 2'
timeout -k 10 400 python scripts/experiments/many_streams.py 12000 3 20000 3 65536 2 100000 2 2>&1 | grep "^S="
echo '$ python scripts/experiments/stream_soak.py'
timeout -k 10 500 python scripts/experiments/stream_soak.py 2>&1 | tail -1
echo '$ OPV_FRONTEND=4 OPV_FUZZ_SEEDS=12 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "fuzz or random_call_sequences or push"'
OPV_SKIP_RCCL_SELFTEST=1 OPV_FRONTEND=4 OPV_FUZZ_SEEDS=12 timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "fuzz or random_call_sequences or push" 2>&1 | tail -2
} > gpurun_out/r05_soaks2.txt 2>&1
tail -20 gpurun_out/r05_soaks2.txt
