#!/bin/bash
# Round 5, GPU visit D: bench line with the certified mode in exact_modes, certified soak, rescore stats of the certified pass
cd "$(dirname "$0")/.."
OUT=gpurun_out/r05d
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench exit $?"
python - <<'PY'
import json
r=json.loads(open('gpurun_out/r05d/bench.json').read().strip().splitlines()[-1])
print(round(r['ms_per_step'],3), {a: round(b,3) for a,b in r['stage_ms'].items()}, r.get('rows_rescored_per_token'))
for k in ('exact_modes','dither_off','zipf'):
    print(k, json.dumps(r.get(k))[:700])
PY
timeout 600 python - <<'PY' 2>&1 | tee gpurun_out/r05d/certified_stage.txt
import sys, torch, json
sys.path[:0]=['.','multimodal-sae_amd','tests']
import bench
from msae import ops
dev=torch.device('cuda:0')
T,d,N,k=8192,4096,131072,32
W_enc,b_enc,W_dec,b_dec,x=bench.make_inputs(dev,T,d,N)
prep=ops.prepare_encoder(W_enc)
for mode in ('default','certified'):
    ops.set_certified(mode=='certified')
    buf=torch.zeros(T,dtype=torch.int32,device=dev)
    prof=ops.StageProfile(6)
    for _ in range(2): ops.encode_topk(x,W_enc,b_enc,b_dec,prep,k)
    with ops.profiling(prof), ops.rescore_rows(buf):
        for _ in range(5): v,i,st=ops.encode_topk(x,W_enc,b_enc,b_dec,prep,k)
    torch.cuda.synchronize()
    s=prof.read().mean(0)
    ok=buf>0
    print(mode, 'stages', [round(float(a),3) for a in s], 'sum', round(float(s.sum()),3), 'rows/token', float((buf[ok]&0xFFF).float().mean()), 'verified', float((st==0).float().mean()))
ops.set_certified(False)
PY
timeout 900 python tools/soak_fused.py --tokens 262144 --N 131072 --d 4096 --coarse certified --out $OUT/soak_certified_c2.json > $OUT/soak.log 2>&1; echo "soak exit $?"; tail -2 $OUT/soak.log | cut -c1-400
