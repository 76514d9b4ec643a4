#!/bin/bash
# round 6, final visit: suite, smoke, bench line (+ bf16), rocprof kernel stats, PMC passes, re-score statistics, shapes,
# latencies, shard emulation, training step, cache loop, real-input runner.  Everything lands in gpurun_out/.
cd "$(dirname "$0")/.."
OUT=gpurun_out
R=r06
mkdir -p $OUT
export TMPDIR=/tmp
(nproc; grep -m1 "model name" /proc/cpuinfo; free -g | head -2; rocm-smi --showproductname 2>/dev/null | head -8) > $OUT/host.txt 2>&1
echo "== pytest -m gpu =="
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?" | tee -a $OUT/pytest_gpu.log
grep -E "passed|failed|error" $OUT/pytest_gpu.log | tail -3
echo "== smoke =="
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke exit $?" | tee -a $OUT/smoke.log; tail -2 $OUT/smoke.log
echo "== bench =="
timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/${R}_bench.json 2> $OUT/bench.err; echo "bench exit $?"; cut -c1-400 $OUT/${R}_bench.json
MSAE_COARSE=bf16 timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary > $OUT/${R}_bench_bf16.json 2>> $OUT/bench.err; echo "bench bf16 exit $?"
echo "== rocprof kernel stats =="
rm -rf $OUT/prof
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o $R -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary > $OUT/${R}_bench_under_rocprof.json 2> $OUT/rocprof.err; echo "rocprof exit $?"
for f in $(find $OUT/prof -name "*kernel_stats*.csv" | head -1); do cp $f $OUT/${R}_kernel_stats.csv; head -12 $f | cut -c1-160; done
find $OUT/prof -name "*kernel_trace*" -size +2M -delete
echo "== PMC passes =="
timeout 1500 bash tools/gpu_pmc.sh > $OUT/pmc.log 2>&1; tail -8 $OUT/pmc.log
python tools/pmc_traffic.py $OUT/pmc_summary.json $OUT/pmc_traffic.json $OUT/pmc/p4/p4_kernel_trace.csv > /dev/null 2>&1
echo "== rescore stats =="
timeout 600 python tools/rescore_stats.py bench trained_like > $OUT/${R}_rescore_stats.txt 2>&1; tail -4 $OUT/${R}_rescore_stats.txt
echo "== shapes / latency / shard emulation / training =="
timeout 600 python tools/sanity_shapes.py > $OUT/${R}_other_shapes.txt 2>&1; cat $OUT/${R}_other_shapes.txt | grep "T="
timeout 300 python tools/latency_small_T.py > $OUT/${R}_latency_small_T.txt 2>&1; grep "T=" $OUT/${R}_latency_small_T.txt
timeout 300 python tools/latency_hook_S1.py >> $OUT/${R}_latency_small_T.txt 2>&1; grep "S=1" $OUT/${R}_latency_small_T.txt
timeout 600 python tools/emulate_shard.py > $OUT/${R}_emulate_shard.txt 2>&1; grep "G=" $OUT/${R}_emulate_shard.txt
timeout 300 python tools/train_step_bench.py > $OUT/${R}_train_step.txt 2>&1; tail -1 $OUT/${R}_train_step.txt
timeout 300 python tools/cache_throughput.py 2>/dev/null | grep "tokens/s" > $OUT/${R}_cache_throughput.txt; cat $OUT/${R}_cache_throughput.txt
echo "== real-input runner (checkpoint dir + activation file written here) =="
timeout 600 python - > $OUT/${R}_real_inputs.txt 2>&1 <<'PY'
import json, subprocess, sys, torch
sys.path[:0] = ["multimodal-sae_amd", "tests"]
import hostile
from safetensors.torch import save_file
from msae import Sae, SaeConfig
dev = torch.device("cuda:0")
d, N, k = 4096, 131072, 32
W, b, bd = hostile.weights("trained_like", N, d, dev, seed=77)
sae = Sae(d, SaeConfig(num_latents=N, k=k), device=dev)
with torch.no_grad():
    sae.encoder.weight.copy_(W); sae.encoder.bias.copy_(b); sae.b_dec.copy_(bd)
    sae.W_dec.copy_(W / (W.norm(dim=1, keepdim=True) + 1e-6))
sae.save_to_disk("/tmp/ckpt/layers.24")
save_file({"acts": hostile.activations(16384, d, dev, seed=78).cpu()}, "/tmp/acts.safetensors")
del sae, W
torch.cuda.empty_cache()
r = subprocess.run([sys.executable, "bench.py", "--sae_path", "/tmp/ckpt/layers.24", "--acts", "/tmp/acts.safetensors",
                    "--tokens", "8192", "--steps", "5", "--warmup", "2"], capture_output=True, text=True)
print("bench --sae_path --acts:", r.stdout.strip()[-1500:], r.stderr[-300:])
r = subprocess.run([sys.executable, "tools/parity_real.py", "--sae_path", "/tmp/ckpt/layers.24", "--acts", "/tmp/acts.safetensors",
                    "--out", "gpurun_out/parity_real.json"], capture_output=True, text=True)
print("parity_real:", r.stdout.strip().splitlines()[-1][:1200], r.stderr[-300:])
PY
tail -3 $OUT/${R}_real_inputs.txt | cut -c1-700
