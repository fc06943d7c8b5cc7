# when do the waves of a workgroup start (scripts/ubench/wave_launch.hip)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r6t14
hipcc --offload-arch=gfx950 -O3 -o /tmp/wave_launch scripts/ubench/wave_launch.hip && /tmp/wave_launch | tee gpurun_out/r6t14/wave_launch.txt
