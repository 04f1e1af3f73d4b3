#!/bin/bash
# The measurements behind DESIGN.md 6 and profiles/<tag>_*: run on the GPU box from the repository root
#   bash tools/final_profiles.sh <tag>        (e.g. r02) -> gpurun_out/<tag>/..., copied into profiles/ by hand
set -e -o pipefail
tag=${1:-r03}
R=$(pwd)
O=$R/gpurun_out/$tag
mkdir -p $O
export TMPDIR=/tmp
# 1. the bench lines: pipelined default (with the CPU legs) and the stages back to back
timeout -k 10 600 python3 bench.py --steps 20 > $O/bench_line.json 2> $O/bench.err
timeout -k 10 300 python3 bench.py --steps 20 --pipeline 0 --cpu-sample 0 > $O/bench_line_serial.json 2> $O/bench_serial.err
# 2. rocprofv3 kernel trace of the same command (program directly after --)
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python3 $R/bench.py --steps 10 --cpu-sample 0 > $O/kt.log 2>&1)
python3 tools/launches_by_grid.py $O/kt > $O/launches_by_grid.txt
python3 tools/timeline.py $O/kt 7 > $O/timeline_pipelined_step.txt
rm -rf $O/kt/*/*.db
cp $O/kt/kt_kernel_stats.csv $O/kernel_stats_batch256.csv 2>/dev/null || cp $(find $O/kt -name '*kernel_stats.csv' | head -1) $O/kernel_stats_batch256.csv
# 3. HBM traffic of the HBM-bound kernels (separate --pmc passes inside the script)
# (ten minutes: SKIP_TRAFFIC=1 leaves it to a call of its own)
if [ -z "$SKIP_TRAFFIC" ]; then
  timeout -k 10 1100 python3 tools/collect_traffic.py 256 > $O/traffic.log 2>&1
  cp gpurun_out/dwt_l1_traffic.json gpurun_out/hbm_traffic_other.json $O/
fi
# 4. the other BASELINE configurations and the colour model change
timeout -k 10 600 python3 tools/other_configs.py > $O/other_configs.txt 2> $O/other_configs.err
timeout -k 10 300 python3 tools/color_timing.py > $O/color_timing.txt 2> $O/color_timing.err
echo done
