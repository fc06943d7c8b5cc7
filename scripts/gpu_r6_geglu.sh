# round 6: A/B of the GEGLU epilogue's arithmetic (gemm_common.h geglu_f: SAID_GEGLU_FORM 0 = value * gelu_f(gate), 1 = the shipped short form, 2 = packed pairs in stchain;
# stchain.hip SAID_GEGLU_BIAS_INIT 1 = the pair's accumulators start from the bias (shipped), 0 = bias added in the epilogue), then the -m gpu suite on the shipped build
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for cfg in cfg2 headline cfg3 cfg4; do bash scripts/gpu_ab.sh $cfg base lib:said_amd/lib/ab_nobias.so; done
mkdir -p gpurun_out/geglu
timeout 1200 python -m pytest tests -m gpu -q -x > gpurun_out/geglu/suite.log 2>&1; echo "suite exit=$?"; tail -3 gpurun_out/geglu/suite.log
