# round 6: battn_kernel's K / V copy without integer divisions (idx / npc through one float multiply): bit-identity against the previous build, then configs[2] A/B
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
bash scripts/gpu_ab.sh EQ=1 cfg2 base lib:said_amd/lib/ab_prev.so
