# round 5: XCD-affine block order in attn_kernel: cfg4 (T = 1800), headline and 32-clip fp32 A/B on one box
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r5; rm -f gpurun_out/r5/attnxcd.txt
for rep in 1 2; do for lib in libsaid_hip ab_attnxcd; do
  echo "== $lib cfg4" | tee -a gpurun_out/r5/attnxcd.txt
  timeout 600 python scripts/bench_lib.py said_amd/lib/$lib.so --seconds 30 --num_steps 100 --edit --steps 2 --warmup 1 --no_cpu_baseline --no_secondary --no_roofline 2>&1 | tail -1 | cut -c1-200 | tee -a gpurun_out/r5/attnxcd.txt
  echo "== $lib headline" | tee -a gpurun_out/r5/attnxcd.txt
  timeout 600 python scripts/bench_lib.py said_amd/lib/$lib.so --steps 2 --warmup 1 --no_cpu_baseline --no_secondary --no_roofline 2>&1 | tail -1 | cut -c1-200 | tee -a gpurun_out/r5/attnxcd.txt
done; done
for lib in libsaid_hip ab_attnxcd; do
  echo "== $lib 32 clips x 100" | tee -a gpurun_out/r5/attnxcd.txt
  timeout 600 python scripts/bench_lib.py said_amd/lib/$lib.so --batch 32 --num_steps 100 --steps 1 --warmup 1 --no_cpu_baseline --no_secondary --no_roofline 2>&1 | tail -1 | cut -c1-200 | tee -a gpurun_out/r5/attnxcd.txt
done
