# stage clocks of the step with the token-major (MSAE_FM=0) and the feature-major (MSAE_FM=1) first round of the re-score
for k in ${KS:-256 32}; do for fm in 0 1; do
  echo "== k=$k MSAE_FM=$fm"; MSAE_FM=$fm timeout 120 python bench.py --k $k --steps 10 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null < /dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(r['ms_per_step'],3), {a: round(b,3) for a,b in r['stage_ms'].items()}, r.get('rows_rescored_per_token'), r.get('rescore_feature_major'))"
done; done
