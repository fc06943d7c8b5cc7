cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r6t3
for m in fp32_strict fp32; do
timeout 600 python tests/debug_stages.py 2 600 600 trained $m > gpurun_out/r6t3/stages_trained_$m.txt 2>&1; echo "exit=$?"
grep -v amdgpu.ids gpurun_out/r6t3/stages_trained_$m.txt | tail -42
done
