cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -x -s -k "bf16 or large_batch or ragged or batch32 or edge_shapes or opt_in" > gpurun_out/r3_hybrid_tests.log 2>&1
echo "exit=$?" >> gpurun_out/r3_hybrid_tests.log
grep -E "bf16|passed|failed|exit" gpurun_out/r3_hybrid_tests.log | cut -c1-170 | tail -40
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
m.set_mfma_dtype('bf16')
for hy, tm in ((1, 0), (0, 0), (0, 1), (1, 0), (0, 0), (1, 0), (0, 0)):
    eng = m._get_engine(2 * B, T); eng.debug_option('hybrid', hy); eng.debug_option('tm_acts', tm)
    m.inference(wav, audio_embedding=ctx, num_inference_steps=10, guidance_scale=2.0, init_latents=lat)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3): m.inference(wav, audio_embedding=ctx, num_inference_steps=N, guidance_scale=2.0, init_latents=lat)
    torch.cuda.synchronize(); dtm = (time.perf_counter() - t0) / 3 / N * 1e3
    print(f'bf16 hybrid={hy} tm_acts={tm}: {dtm:.3f} ms per step in situ, nodes {eng.graph_num_nodes()}', flush=True)
PY
python /tmp/insitu.py 2>&1 | grep "in situ"
