cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -x -k "opt_in" > gpurun_out/r3_tm_tests.log 2>&1
echo "exit=$?" >> gpurun_out/r3_tm_tests.log
tail -3 gpurun_out/r3_tm_tests.log
cat > /tmp/insitu.py <<'PY'
import sys, time, torch
sys.path.insert(0, '.')
from said_amd.model.diffusion import SAID_UNet1D
from said_amd.util import synth
torch.set_grad_enabled(False)
dev = torch.device('cuda:0')
m = SAID_UNet1D(); m.load_state_dict(synth.said_state_dict(), strict=True); m.to(dev).eval()
B, T, N = 32, 600, 50
ctx = synth.synth_latents(1, (B, T, 768)).to(dev); lat = synth.synth_latents(2, (B, T, 32)).to(dev)
wav = torch.zeros(B, T * 16000 // 60, device=dev)
for dt in ('bf16', 'fp32'):
    m.set_mfma_dtype(dt)
    for tm, ntw in ((0, 0), (1, 0), (1, 2), (1, 3), (1, 6), (0, 0)):
        eng = m._get_engine(2 * B, T); eng.debug_option('tm_acts', tm); eng.debug_option('xgemm_ntw', ntw)
        m.inference(wav, audio_embedding=ctx, num_inference_steps=10, guidance_scale=2.0, init_latents=lat)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(3): m.inference(wav, audio_embedding=ctx, num_inference_steps=N, guidance_scale=2.0, init_latents=lat)
        torch.cuda.synchronize(); dtm = (time.perf_counter() - t0) / 3 / N * 1e3
        print(f'{dt} tm_acts={tm} ntw={ntw}: {dtm:.3f} ms per step in situ, nodes {eng.graph_num_nodes()}', flush=True)
PY
python /tmp/insitu.py 2>&1 | grep "in situ"
for cfg in "bf16 tm_acts=1 xgemm_ntw=6"; do
  timeout 300 python scripts/profile_stages.py 32 600 $cfg 2>&1 | tail -65
done > gpurun_out/r3_tm_stages2.log 2>&1
grep "^sum" gpurun_out/r3_tm_stages2.log
