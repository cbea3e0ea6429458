#!/bin/bash
# Run on the GPU box from the repo root: kernel-trace statistics and the two HBM-traffic PMC
# passes for bench.py's default workload; raw outputs under gpurun_out/prof_<tag>/, summaries
# written by tools/summarize_profile.py.   usage: tools/profile_round.sh r01
set -e
tag=${1:-r01}
out=gpurun_out/prof_$tag
rm -rf $out && mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $out/bench_stats.json 2> $out/stats.err
echo "stats pass done"
timeout -s KILL 240 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2> $out/fetch.err
echo "fetch pass done"
timeout -s KILL 240 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2> $out/write.err
echo "write pass done"
python3 tools/summarize_profile.py $out $tag
# keep only what the summary needs (the traces are tens of MB)
find $out -name "*kernel_trace.csv" -delete
find $out -name "*counter_collection.csv" -delete
