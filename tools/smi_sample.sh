#!/bin/bash
# usage: smi_sample.sh name skip_seconds -- command...   Runs the command in the background and samples rocm-smi power / clocks
# beside it (after skip_seconds, or once the command has printed "ready" if skip_seconds is the word ready); prints the medians.  Output under gpurun_out/power/.
cd "$(dirname "$0")/.."
name=$1; skip=$2; shift 3
OUT=gpurun_out/power; mkdir -p $OUT
( "$@" > $OUT/$name.out 2>&1 ) &
BP=$!
if [ "$skip" = ready ]; then while kill -0 $BP 2>/dev/null && ! grep -q ready $OUT/$name.out 2>/dev/null; do sleep 0.2; done; sleep 1; else sleep $skip; fi
: > $OUT/$name.smi
while kill -0 $BP 2>/dev/null; do
  rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk|fclk" >> $OUT/$name.smi
  sleep 0.2
done
wait $BP
python - <<PY
import re
t=open("$OUT/$name.smi").read()
line="$name:"
for key,pat in (("W", r"Power[^:]*: ([0-9.]+)"), ("sclk", r"sclk clock level[^(]*\(([0-9]+)Mhz\)"), ("fclk", r"fclk clock level[^(]*\(([0-9]+)Mhz\)")):
    v=sorted(float(x) for x in re.findall(pat,t))
    if v: line += "  %s median %.0f (p10 %.0f, p90 %.0f, n=%d)" % (key, v[len(v)//2], v[len(v)//10], v[len(v)*9//10], len(v))
print(line)
PY
tail -3 $OUT/$name.out
