# Re-measures the bf16-vs-fp32 search agreement at an older tree and at HEAD on one box.
# Set-up (build container): git worktree add -f _r1 3ba4d59 && make -C _r1/muzero_amd/csrc -j8   (then run this on the GPU box)
set -e
for A in 18 6; do
 (cd _r1 && timeout -k 10 200 python tools/bf16_vs_fp32_search.py 128 $A 50) >> gpurun_out/r3_tv_r1.log 2>&1
 timeout -k 10 300 python tools/bf16_vs_fp32_search.py 128 $A 50 0 >> gpurun_out/r3_tv_head.log 2>&1
done
(cd _r1 && timeout -k 10 300 python tools/bf16_vs_fp32_search.py 1024 18 50) >> gpurun_out/r3_tv_r1.log 2>&1
timeout -k 10 400 python tools/bf16_vs_fp32_search.py 1024 18 50 0 >> gpurun_out/r3_tv_head.log 2>&1
timeout -k 10 400 python tools/bf16_vs_fp32_search.py 1024 18 50 1 >> gpurun_out/r3_tv_head.log 2>&1
timeout -k 10 400 python tools/bf16_vs_fp32_search.py 1024 6 50 0 >> gpurun_out/r3_tv_head.log 2>&1
