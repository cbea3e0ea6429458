#!/bin/bash
# Run on the GPU box from the repo root: kernel-trace statistics and the two HBM-traffic PMC
# passes for bench.py's default workload, plus kernel statistics of the float32-accurate precision
# and of the training update; raw outputs under gpurun_out/prof_<tag>/, summaries written by
# tools/summarize_profile.py.   usage: tools/profile_round.sh r02
set -e
tag=${1:-r03}
out=gpurun_out/prof_$tag
rm -rf $out && mkdir -p $out
export TMPDIR=/tmp
B="--no-cpu-baseline --no-extras"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py --steps 3 --warmup 1 $B > $out/bench_stats.json 2> $out/stats.err
echo "stats pass done"
timeout -s KILL 240 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -- python3 bench.py --steps 1 --warmup 1 $B > /dev/null 2> $out/fetch.err
echo "fetch pass done"
timeout -s KILL 240 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -- python3 bench.py --steps 1 --warmup 1 $B > /dev/null 2> $out/write.err
echo "write pass done"
python3 tools/summarize_profile.py $out $tag
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/x3 -- python3 bench.py --steps 2 --warmup 1 --precision bf16x3 $B > $out/bench_x3.json 2> $out/x3.err
python3 tools/top_kernels.py $out/x3 14 > $out/${tag}_x3_kernels.txt
echo "x3 pass done"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/f16 -- python3 bench.py --steps 2 --warmup 1 --precision f16 $B > $out/bench_f16.json 2> $out/f16.err
python3 tools/top_kernels.py $out/f16 14 > $out/${tag}_f16_kernels.txt
echo "f16 pass done"
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $out/train -- python3 tools/profile_train.py 128 bf16 native graphed > /dev/null 2> $out/train.err
python3 tools/last_window.py $out/train 7.9 60 > $out/${tag}_train_update.txt
echo "train pass done"
# keep only what the summaries need (the traces are tens of MB)
find $out -name "*kernel_trace.csv" -delete
find $out -name "*counter_collection.csv" -delete
