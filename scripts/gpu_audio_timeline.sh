cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/at
timeout 400 rocprofv3 --kernel-trace -d gpurun_out/at -o a -- python scripts/audio_trace.py 32 bf16 > gpurun_out/at/run.log 2>&1
python scripts/audio_trace.py 32 bf16 $(find gpurun_out/at -name "a_results.db" | head -1) > gpurun_out/at/timeline.txt
find gpurun_out/at -name "*.db" -delete
grep -E "tgemm256d|attn_kernel|ln_tm|pass:" gpurun_out/at/timeline.txt | sed -n 1,14p
tail -1 gpurun_out/at/timeline.txt
