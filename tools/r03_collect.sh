#!/bin/bash
# Turns what tools/r03_profile.sh left under gpurun_out/r03/ into the committed summaries under profiles/ (run here, no GPU needed).
set -e
O=gpurun_out/r03
P=profiles
cp $O/bench.json $P/r03_bench.json
cp $O/bench_c4.json $P/r03_bench_c4.json
cp $O/bench_c5.json $P/r03_bench_c5.json
cp $O/bench_under_rocprof.json $P/r03_bench_under_rocprof.json
cp $O/stats_c2/s_kernel_stats.csv $P/r03_kernel_stats.csv
cp $O/stats_c3/s_kernel_stats.csv $P/r03_c3_kernel_stats.csv
cp $O/stats_c4/s_kernel_stats.csv $P/r03_c4_kernel_stats.csv
cp $O/stats_c5/s_kernel_stats.csv $P/r03_c5_kernel_stats.csv
S="python tools/pmc_summary.py"
passes() { echo $O/$1/sq1 $O/$1/sq2 $O/$1/tcc $O/$1/tcp $O/$1/fetch $O/$1/write; }
$S --kernel "trace_grid_kernel<false, false, true, true, true" --out $P/r03_pmc.json --alg-bytes 24883904 --min-us 500 \
   --command "tools/pmc_passes.sh <dir> python3 bench.py --no-other-configs --cpu-spp 0 --steps 5 --warmup 1" \
   --workload "C2 1920x1080 spp 64 (the bench frame)" $(passes pmc_c2)
$S --kernel "trace_grid_sched_kernel<true, false, true, true" --out $P/r03_c3_glass_bunny_pmc.json --alg-bytes 50685980 --min-us 500 \
   --command "tools/pmc_passes.sh <dir> python3 tools/one_frame.py c3 64" \
   --workload "C3 2048x2048 spp 64 glass bunny + ChessBoard floor; scheduled launch only (light tiles beside it)" $(passes pmc_c3)
$S --kernel "trace_grid_sched_kernel<true, false, true, false" --out $P/r03_c4_dragon_pmc.json --alg-bytes 210362044 --min-us 200 \
   --command "tools/pmc_passes.sh <dir> python3 tools/one_frame.py c4 64" \
   --workload "C4 4096x4096 spp 64 (a quarter of the configuration's samples), dragon 100 000 triangles; the scheduled kernel (what is left of it: the heavy units are completed by primary_walk_kernel)" $(passes pmc_c4)
$S --kernel "primary_walk_kernel" --out $P/r03_c4_primary_walk_pmc.json --alg-bytes 210362044 --min-us 500 \
   --command "tools/pmc_passes.sh <dir> python3 tools/one_frame.py c4 64" \
   --workload "C4 4096x4096 spp 64: primary_walk_kernel (walk-only kernel with lane refill; completes the heavy units)" $(passes pmc_c4)
$S --kernel "trace_grid_kernel<false, false, true, false, false" --out $P/r03_c4_light_pmc.json --alg-bytes 210362044 --min-us 500 \
   --command "tools/pmc_passes.sh <dir> python3 tools/one_frame.py c4 64" \
   --workload "C4 4096x4096 spp 64: the light-tile launch (planes only)" $(passes pmc_c4)
$S --kernel "trace_grid_sched_kernel<true, true, true" --out $P/r03_c5_bezier_band_pmc.json --alg-bytes 37000000 --min-us 500 \
   --command "tools/pmc_passes.sh <dir> python3 tools/one_frame.py c5band 16" \
   --workload "C5 band 8192x256 rows through the vase, spp 16; scheduled launch only" $(passes pmc_c5)
python tools/resource_table.py --out $P/r03_resource_usage.json > /dev/null
