#!/bin/bash
# One full GPU-box visit of a round: parity tests, smoke, bench (+bf16), rocprof kernel stats, PMC passes,
# re-score statistics, soak, other shapes, small-T latency, per-rank shard emulation, training step,
# real-input runner.  Everything lands in gpurun_out/ (merged back by gpurun); each step has its own timeout.
cd "$(dirname "$0")/.."
OUT=gpurun_out
R=${ROUND:-r06}
mkdir -p $OUT
export TMPDIR=/tmp
(nproc; grep -m1 "model name" /proc/cpuinfo; free -g | head -2; rocm-smi --showproductname 2>/dev/null | head -8) > $OUT/host.txt 2>&1
if [ -z "$SKIP_TESTS" ]; then
echo "== pytest -m gpu =="
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -s > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?" | tee -a $OUT/pytest_gpu.log
grep -E "passed|failed|error" $OUT/pytest_gpu.log | tail -5
fi
echo "== smoke =="
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke exit $?" | tee -a $OUT/smoke.log; tail -2 $OUT/smoke.log
echo "== bench =="
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/${R}_bench.json 2> $OUT/bench.err; echo "bench exit $?"; cat $OUT/${R}_bench.json
MSAE_COARSE=bf16 timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary > $OUT/${R}_bench_bf16.json 2>> $OUT/bench.err; echo "bench bf16 exit $?"
echo "== rocprof kernel stats =="
rm -rf $OUT/prof
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o $R -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary > $OUT/${R}_bench_under_rocprof.json 2> $OUT/rocprof.err; echo "rocprof exit $?"
for f in $(find $OUT/prof -name "*kernel_stats*.csv" | head -1); do head -12 $f | cut -c1-160; done
find $OUT/prof -name "*kernel_trace*" -size +2M -delete
echo "== PMC passes =="
timeout 1500 bash tools/gpu_pmc.sh > $OUT/pmc.log 2>&1; tail -8 $OUT/pmc.log
python tools/pmc_traffic.py $OUT/pmc_summary.json $OUT/pmc_traffic.json $OUT/pmc/p4/p4_kernel_trace.csv > /dev/null 2>&1
echo "== rescore stats =="
timeout 600 python tools/rescore_stats.py bench trained_like > $OUT/${R}_rescore_stats.txt 2>&1; tail -4 $OUT/${R}_rescore_stats.txt
timeout 200 python tools/cluster_rows.py 2>&1 | grep cluster > $OUT/${R}_cluster_rows.txt; cat $OUT/${R}_cluster_rows.txt

echo "== soak =="
timeout 900 python tools/soak_fused.py --tokens 1048576 --N 32768 --d 1024 --out $OUT/${R}_soak_1M_trained_like_n32768.json > $OUT/soak.log 2>&1; echo "soak exit $?"; tail -1 $OUT/soak.log | cut -c1-400
timeout 900 python tools/soak_fused.py --tokens 4194304 --N 131072 --d 4096 --out $OUT/${R}_soak_4M_trained_like_c2.json >> $OUT/soak.log 2>&1; echo "soak c2 exit $?"
timeout 900 python tools/soak_fused.py --tokens 2097152 --N 262144 --d 4096 --out $OUT/${R}_soak_2M_trained_like_n262144.json >> $OUT/soak.log 2>&1; echo "soak N=262144 exit $?"
timeout 900 python tools/soak_fused.py --tokens 1048576 --N 131072 --d 4096 --coarse bf16 --out $OUT/${R}_soak_1M_trained_like_c2_bf16.json >> $OUT/soak.log 2>&1; echo "soak c2 bf16 exit $?"
timeout 900 python tools/soak_fused.py --tokens 524288 --N 131072 --d 4096 --coarse certified --out $OUT/${R}_soak_512k_trained_like_c2_certified.json >> $OUT/soak.log 2>&1; echo "soak c2 certified exit $?"
MSAE_DITHER=0 timeout 900 python tools/soak_fused.py --tokens 1048576 --N 131072 --d 4096 --out $OUT/${R}_soak_1M_trained_like_c2_dither_off.json >> $OUT/soak.log 2>&1; echo "soak c2 dither-off exit $?"
echo "== feature-major re-score: A/B, probe, k = 256 soak, fuzz with the route forced on / off =="
timeout 400 bash tools/gpu_fm_ab.sh > $OUT/${R}_fm_rescore.txt 2>&1; cat $OUT/${R}_fm_rescore.txt | cut -c1-220
KS=256 timeout 250 bash tools/gpu_fm.sh > /dev/null 2>&1; cp $OUT/fm/kernel_stats_k256.csv $OUT/${R}_k256_kernel_stats.csv 2>/dev/null
[ -x tools/bin/fm_rescore_probe ] && timeout 200 tools/bin/fm_rescore_probe > $OUT/${R}_fm_rescore_probe.txt 2>&1 < /dev/null
timeout 600 python tools/soak_fused.py --tokens 262144 --N 131072 --d 4096 --k 256 --out $OUT/${R}_soak_k256_trained_like_c2.json >> $OUT/soak.log 2>&1; echo "soak k256 exit $?"
(MSAE_FM=1 timeout 400 python tools/fuzz_fused.py 1500 11; MSAE_FM=0 timeout 400 python tools/fuzz_fused.py 1500 11; timeout 400 python tools/fuzz_fused.py 1500 12; timeout 600 python tools/fuzz_fused.py 2500 21 int8,bf16,fp8,certified,int8_rn) 2>&1 | grep -i "cases" > $OUT/${R}_fuzz.txt; cat $OUT/${R}_fuzz.txt
echo "== exact path kernel (pre_acts_f32) =="
timeout 120 python tools/f32_probe.py 10 2>&1 | tail -1 | tee $OUT/${R}_f32_rate.txt
echo "== shapes / latency / shard emulation / training =="
timeout 600 python tools/sanity_shapes.py > $OUT/${R}_other_shapes.txt 2>&1; cat $OUT/${R}_other_shapes.txt | grep "T="
timeout 300 python tools/latency_small_T.py > $OUT/${R}_latency_small_T.txt 2>&1; grep "T=" $OUT/${R}_latency_small_T.txt
timeout 600 python tools/emulate_shard.py > $OUT/${R}_emulate_shard.txt 2>&1; grep "G=" $OUT/${R}_emulate_shard.txt
timeout 300 python tools/train_step_bench.py > $OUT/${R}_train_step.txt 2>&1; tail -1 $OUT/${R}_train_step.txt
timeout 300 python tools/cache_throughput.py 2>/dev/null | grep "tokens/s" > $OUT/${R}_cache_throughput.txt; cat $OUT/${R}_cache_throughput.txt
echo "== T=1 step kernel timeline =="
rm -rf $OUT/prof_t1
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$OUT/prof_t1 -o t1 -- python $OLDPWD/tools/t1_trace.py > $OLDPWD/$OUT/t1.log 2>&1)
for f in $(find $OUT/prof_t1 -name "*kernel_stats.csv" | head -1); do cp $f $OUT/${R}_t1_kernel_stats.csv; head -8 $f | cut -c1-140; done
find $OUT/prof_t1 -name "*kernel_trace*" -delete
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
