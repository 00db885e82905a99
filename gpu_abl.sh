#!/bin/bash
# scratch: ablation timing of the front-end (results are WRONG by construction; timing only)
cat > /tmp/abl.py <<'PY'
import sys, time, numpy as np, torch
sys.path.insert(0,'tests')
from amd_lib import load
amd=load()
F=40; S=64
tx=amd.bert_frames(F); base=amd.modulate(tx); n=base.size//2
dev=torch.device('cuda',0)
d=torch.from_numpy(base).to(dev)
dm=amd.Demod(S,max_samples=n+64,streaming=True)
dm.enable_timing(True)
for rep in range(3):
    dm.reset()
    for k in range(S): dm.attach(k,d.data_ptr(),n,eof=True)
    dm.process(); dm.sync()
    t=dm.kernel_times()
dm.close(); print('frontend ms %.2f  (%.1f ns/symbol)'%(t['msk_frontend'], t['msk_frontend']*1e6/(F*2168+100)))
PY
for a in 0 1 2 3 4 5; do
  if [ $a = 0 ]; then unset OPV_AMD_LIB; else export OPV_AMD_LIB=$PWD/gpurun_abl/libabl$a.so; fi
  echo -n "ablate $a: "; timeout 120 python /tmp/abl.py 2>&1 | tail -1
done
