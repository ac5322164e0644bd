import sys, time, torch
sys.path.insert(0, '.')
from magicdrive_amd.networks import spec
from magicdrive_amd import synthetic
from oracle import denoiser as D
cfg = spec.SD15_CONFIG
usd = {k: torch.randn(s)*0.02 for k,s in spec.unet_param_shapes(cfg).items()}
sc = synthetic.make_scene_batch(1, max_len=None, zero_map=True)
lat = torch.stack([sc['latents']]*6, 1).reshape(-1,4,28,50)
ctx = torch.randn(6, 78, 768)
for n in (16, 32, 64, 128):
    torch.set_num_threads(n)
    with torch.no_grad():
        D.unet_forward(usd, cfg, lat, 981, ctx, None, None)
        t0=time.perf_counter(); D.unet_forward(usd, cfg, lat, 961, ctx, None, None); dt=time.perf_counter()-t0
    print(n, 'threads: unet fwd', round(dt,2), 's', flush=True)
