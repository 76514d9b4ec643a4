#!/bin/bash
# Is the candidate GEMM clock- or power-limited?  Runs bench.py (many steps) with the given library in the background and samples
# the SMU's power / clock readings beside it; then prints the tile timeline's cycle count of the same build (if it is a timeline
# build) so that cycles / time = the effective shader clock.  usage: power_probe.sh name lib.so [steps]
cd "$(dirname "$0")/.."
name=$1; lib=$2; steps=${3:-400}
OUT=gpurun_out/power; mkdir -p $OUT
( [ -n "$lib" ] && export MSAE_HIP_LIB=$lib; timeout 300 python bench.py --steps $steps --warmup 5 --no-cpu-baseline > $OUT/$name.json 2>$OUT/$name.err ) &
BP=$!
sleep 20   # import + input generation
: > $OUT/$name.smi
while kill -0 $BP 2>/dev/null; do
  rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -E "Power|sclk|fclk|mclk|Temperature \(Sensor (junction|edge)" >> $OUT/$name.smi
  echo "--" >> $OUT/$name.smi
  sleep 0.3
done
wait $BP
python - <<PY
import json,re
r=json.load(open("$OUT/$name.json"))
print("$name: step %.3f ms main_gemm %.3f" % (r["ms_per_step"], r["stage_ms"]["main_gemm"]))
t=open("$OUT/$name.smi").read()
for key,pat in (("power W", r"Power[^:]*: ([0-9.]+)"), ("sclk MHz", r"sclk clock level[^(]*\(([0-9]+)Mhz\)"), ("fclk MHz", r"fclk clock level[^(]*\(([0-9]+)Mhz\)"), ("temp C", r"junction\) \(C\): ([0-9.]+)")):
    v=[float(x) for x in re.findall(pat,t)]
    if v:
        v2=sorted(v); print("  %-9s n=%d  median %.0f  p10 %.0f  p90 %.0f  max %.0f" % (key,len(v),v2[len(v)//2],v2[len(v)//10],v2[len(v)*9//10],v2[-1]))
PY
tail -12 $OUT/$name.smi | head -8
