#!/bin/bash
# Run on the GPU box from the repo root: SQ / GRBM counters of the two-image chain kernel (dynamics shape, 5 layers,
# 1024 images) and of the 18-layer representation-tail launch, and the chain's time per layer against its length.
# usage: tools/pmc_pair.sh <tag>      raw data under gpurun_out/pmc_<tag>/, summary in gpurun_out/pmc_<tag>/summary.txt
set -e
tag=${1:-r03}
out=gpurun_out/pmc_$tag
rm -rf $out && mkdir -p $out
export TMPDIR=/tmp
S=$out/summary.txt
echo "== time per layer against chain length (tools/bench_conv.py --shape chainN --batch 1024 --iters 40, back-to-back launches)" > $S
for n in 1 2 5 10 18 20; do
  timeout -k 10 120 python3 tools/bench_conv.py --shape chain$n --batch 1024 --iters 40 >> $S
done
for n in 5 18; do
  timeout -s KILL 120 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU --output-format csv -d $out/sq$n -- python3 tools/bench_conv.py --shape chain$n --batch 1024 --iters 10 > /dev/null 2> $out/sq$n.err
  echo "== conv_chain_pair_kernel, $n layers, 1024 images: rocprofv3 --pmc (SQ block), mean per dispatch" >> $S
  python3 tools/pmc_summary.py $out/sq$n conv_chain_pair >> $S
  timeout -s KILL 120 rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT --output-format csv -d $out/grbm$n -- python3 tools/bench_conv.py --shape chain$n --batch 1024 --iters 10 > /dev/null 2> $out/grbm$n.err
  echo "== the same, GRBM block" >> $S
  python3 tools/pmc_summary.py $out/grbm$n conv_chain_pair >> $S
done
find $out -name "*counter_collection.csv" -delete
cat $S
