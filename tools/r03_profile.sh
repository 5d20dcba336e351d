#!/bin/bash
# Round-3 evidence run on the GPU box (one gpurun call): bench lines, rocprofv3 kernel stats and PMC passes of the same commands.
#   bash tools/r03_profile.sh        (writes under gpurun_out/r03/)
set -x
O=gpurun_out/r03
mkdir -p $O
export TMPDIR=/tmp
python bench.py > $O/bench.json 2> $O/bench.err
python bench.py --config c4 > $O/bench_c4.json 2>> $O/bench.err
python bench.py --config c5 > $O/bench_c5.json 2>> $O/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c2 -o s -- python3 bench.py --no-other-configs --cpu-spp 0 > $O/bench_under_rocprof.json 2> $O/stats_c2.log
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c3 -o s -- python3 tools/one_frame.py c3 64 > $O/stats_c3.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c4 -o s -- python3 tools/one_frame.py c4 256 > $O/stats_c4.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c5 -o s -- python3 tools/one_frame.py c5band 16 > $O/stats_c5.log 2>&1
bash tools/pmc_passes.sh $O/pmc_c2 python3 bench.py --no-other-configs --cpu-spp 0 --steps 5 --warmup 1 > $O/pmc_c2.log 2>&1
bash tools/pmc_passes.sh $O/pmc_c3 python3 tools/one_frame.py c3 64 > $O/pmc_c3.log 2>&1
bash tools/pmc_passes.sh $O/pmc_c4 python3 tools/one_frame.py c4 64 > $O/pmc_c4.log 2>&1
bash tools/pmc_passes.sh $O/pmc_c5 python3 tools/one_frame.py c5band 16 > $O/pmc_c5.log 2>&1
find $O -name "*.csv" -size +8M -delete
du -sh $O
