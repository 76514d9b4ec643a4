#!/bin/bash
# forced-collective 1-rank run of the N > 1 flow over RCCL (the only RCCL contact a 1-GPU box allows) on the final tree
cd "$(dirname "$0")/.."
OUT=gpurun_out/r05m
mkdir -p $OUT
export TMPDIR=/tmp
MSAE_FORCE_COLLECTIVES=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 10 --warmup 3 > $OUT/bench_forced.json 2> $OUT/bench_forced.err; echo "forced exit $?"
python - <<'PY'
import json
r=json.loads(open('gpurun_out/r05m/bench_forced.json').read().strip().splitlines()[-1])
keep={k: r.get(k) for k in ('ms_per_step','value','n_gpus','scaling','rccl_world','collective_backend','collective_ms','ms_per_step_no_recon_gather','config')}
print(json.dumps(keep)[:1500])
PY
tail -3 $OUT/bench_forced.err | cut -c1-300
