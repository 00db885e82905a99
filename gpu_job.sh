#!/bin/bash
# scratch GPU job script (invoked through gpurun)
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
echo "== cli"; B=opv-cxx-demod_amd/bin
timeout 120 $B/opv-mod -S W5NYV -B 10 > /tmp/c1.iq
timeout 120 $B/opv-demod -s -r < /tmp/c1.iq > /tmp/c1_s.bin 2> gpurun_out/cli_stream_stderr.txt; echo rc=$? $(sha256sum < /tmp/c1_s.bin)
timeout 120 $B/opv-demod -r -q < /tmp/c1.iq > /tmp/c1_b.bin 2> gpurun_out/cli_batch_stderr.txt; echo rc=$? $(sha256sum < /tmp/c1_b.bin)
diff <(cat gpurun_out/cli_stream_stderr.txt) tests/golden/c1_stream_stderr.txt > gpurun_out/cli_stderr.diff; echo "stderr diff lines: $(wc -l < gpurun_out/cli_stderr.diff)"
echo "== bench small"; timeout 600 python bench.py --streams 8 --frames 50 --steps 2 --warmup 1 --no-extras 2>&1 | grep -v amdgpu.ids | tail -5
echo "== bench full"; timeout 1500 python bench.py 2>&1 | tail -5 | tee gpurun_out/bench_full.log
