# attn_kernel head_dim 64 (audio encoder) in the vector-register form too: bit-identity against the previous build (audio embedding, both modes) + configs[2] A/B
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r6t31
cat > /tmp/enc_eq.py <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
import torch
from said_amd import _engine
_engine._LIB_PATH = os.path.abspath(sys.argv[1])
from said_amd.model.diffusion import SAID_UNet1D
from said_amd.util import synth
from oracle import pipeline as op
torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
m = SAID_UNet1D(); m.load_state_dict(synth.said_state_dict(), strict=True); m.to(dev).eval()
out = {}
for mode in ("fp32", "bf16"):
    m.set_mfma_dtype(mode)
    for B, Ta in ((1, 160000), (32, 160000), (3, 48000)):
        proc = op.process_audio([synth.synth_waveform(900 + i, Ta).numpy() for i in range(B)]).to(dev)
        out[f"{mode}_{B}x{Ta}"] = m.get_audio_embedding(proc, int(Ta / 16000 * 60)).cpu()
if sys.argv[2] == "save": torch.save(out, sys.argv[3]); print("saved")
else:
    ref = torch.load(sys.argv[3]); bad = [k for k in out if not torch.equal(out[k], ref[k])]
    for k in out: print(k, f"max |diff| {float((out[k] - ref[k]).abs().max()):.3e}")
    print("BIT-IDENTICAL" if not bad else f"DIFFERENT {bad}")
PY
timeout 600 python /tmp/enc_eq.py said_amd/lib/ab_prev.so save /tmp/enc_ref.pt 2>&1 | tail -1
timeout 600 python /tmp/enc_eq.py said_amd/lib/libsaid_hip.so cmp /tmp/enc_ref.pt 2>&1 | tail -7 | tee gpurun_out/r6t31/equal.txt
for rep in 1 2; do for lib in "--ab_lib said_amd/lib/ab_prev.so" ""; do
  echo "== cfg2 (32 clips x 50 steps, bf16) $lib" | tee -a gpurun_out/r6t31/ab.txt
  timeout 600 python bench.py --batch 32 --num_steps 50 --dtype bf16 --steps 3 --warmup 1 --no_cpu_baseline --no_secondary --no_roofline $lib 2>&1 | tail -1 | cut -c1-160 | tee -a gpurun_out/r6t31/ab.txt
done; done
for lib in "--ab_lib said_amd/lib/ab_prev.so" ""; do
  echo "== headline $lib" | tee -a gpurun_out/r6t31/ab.txt
  timeout 600 python bench.py --steps 5 --warmup 2 --no_cpu_baseline --no_secondary --no_roofline $lib 2>&1 | tail -1 | cut -c1-160 | tee -a gpurun_out/r6t31/ab.txt
done
