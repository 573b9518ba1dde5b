#!/bin/bash
# rocprofv3 kernel stats + PMC traffic passes + the round's bench line, in that order: bench.py reads the PMC file of THIS
# kernel source (hash-checked) from profiles/, so the traffic passes run first and their result is copied there on the box.
# Afterwards copy gpurun_out/<tag>_{bench.json,pmc_traffic.json,rocprofv3_kernel_stats.csv,kernel_event_times.json} into profiles/ and commit.
#   usage: tools/bench_profile.sh [tag]     (default r04)
set -o pipefail
TAG=${1:-r04}
mkdir -p gpurun_out/prof
rm -rf gpurun_out/prof/stats gpurun_out/prof/fetch gpurun_out/prof/write
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/stats -- python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-train-leg --no-sweep > gpurun_out/prof_stats.out 2>&1 || true
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof/fetch -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-train-leg --no-sweep > gpurun_out/prof_fetch.out 2>&1 || true
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/prof/write -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-train-leg --no-sweep > gpurun_out/prof_write.out 2>&1 || true
python tools/pmc_traffic.py gpurun_out/prof/fetch gpurun_out/prof/write "mg_step_kernel<false,16,false,true,true>" > gpurun_out/${TAG}_pmc_traffic.json && cp gpurun_out/${TAG}_pmc_traffic.json profiles/${TAG}_pmc_traffic.json; cat gpurun_out/${TAG}_pmc_traffic.json
f=$(find gpurun_out/prof/stats -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/${TAG}_rocprofv3_kernel_stats.csv; head -8 gpurun_out/${TAG}_rocprofv3_kernel_stats.csv
python bench.py --profile-out gpurun_out/${TAG}_kernel_event_times.json > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err || { tail -5 gpurun_out/${TAG}_bench.err; exit 1; }
python -c "import json;d=json.load(open('gpurun_out/${TAG}_bench.json'));print(d['value'], d['ms_per_step'], d['roofline'], d.get('with_policy'), [ (x['envs'], round(x['frac'],3)) for x in d['roofline_large_b']])"
