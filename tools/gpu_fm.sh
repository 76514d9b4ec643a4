# per-kernel times of the feature-major re-score (MSAE_FM=1) under rocprofv3
mkdir -p gpurun_out/fm
for k in ${KS:-256 32}; do
rm -rf gpurun_out/fm/prof_$k
MSAE_FM=${FM:-1} timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/fm/prof_$k -o fm -- python bench.py --k $k --steps 5 --warmup 2 --no-cpu-baseline --no-secondary > /dev/null 2> gpurun_out/fm/rocprof_$k.err < /dev/null
for f in $(find gpurun_out/fm/prof_$k -name "*kernel_stats.csv" | head -1); do cp $f gpurun_out/fm/kernel_stats_k$k.csv; head -14 $f | cut -c1-220; done
find gpurun_out/fm/prof_$k -name "*kernel_trace*" -delete
done
