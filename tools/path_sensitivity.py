import sys, time; sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import torch, copy
from magicdrive_amd.networks import spec
from magicdrive_amd import synthetic
from oracle import denoiser as D
cfg = spec.TINY_CONFIG if len(sys.argv)<2 else spec.SD15_CONFIG
usd = spec.random_state_dict(spec.unet_param_shapes(cfg), 0)
csd = spec.random_state_dict(spec.controlnet_param_shapes(cfg), 1)
nb_=1 if len(sys.argv)>1 else 2
sc = synthetic.make_scene_batch(nb_, ctx_dim=cfg['cross_attention_dim'], max_len=5, latent_hw=(28,50))
lat = torch.stack([sc['latents']]*6, 1)
t = torch.tensor([981]*nb_)
def run(usd, csd, lat, cfg=cfg):
    d, m, ctx = D.controlnet_forward(csd, cfg, lat, t, sc['camera_param'], sc['bboxes_3d_data'], sc['prompt_embeds'], sc['bev_map'])
    e = D.unet_forward(usd, cfg, lat.reshape(-1,4,28,50), 981, ctx, d, m)
    return d, m, ctx, e
rel = lambda a,b: ((a-b).norm()/b.norm()).item()
with torch.no_grad():
    t0=time.time(); d, m, ctx, e = run(usd, csd, lat); print('time', time.time()-t0)
    print('ctx mag', ctx.abs().max().item(), 'cam tok', ctx[:,0].std().item(), 'text', ctx[:,1:78].std().item(), 'box tok', ctx[:,78:].std().item(), 'eps std', e.std().item(), 'mid std', m.std().item(), 'd0 std', d[0].std().item())
    bf = lambda sd: {k: v.to(torch.bfloat16).float() for k,v in sd.items()}
    d2, m2, ctx2, e2 = run(bf(usd), bf(csd), lat)
    print('bf16-weights rel err: eps', rel(e2,e), 'mid', rel(m2,m))
    print('views differ rel', rel(e[0],e[1]))
    cfg2 = copy.deepcopy(cfg); cfg2['neighboring_view_pair'] = {0:[1,2],1:[2,3],2:[3,4],3:[4,5],4:[5,0],5:[0,1]}
    print('other neighbours rel', rel(run(usd, csd, lat, cfg2)[3], e))
    nb = {k: (torch.zeros_like(v)) for k,v in sc['bboxes_3d_data'].items()}
    d3, m3, ctx3 = D.controlnet_forward(csd, cfg, lat, t, sc['camera_param'], nb, sc['prompt_embeds'], sc['bev_map'])
    print('no boxes rel', rel(D.unet_forward(usd, cfg, lat.reshape(-1,4,28,50), 981, ctx3, d3, m3), e))
    d4, m4, ctx4 = D.controlnet_forward(csd, cfg, lat, t, sc['camera_param'], sc['bboxes_3d_data'], sc['prompt_embeds'], torch.zeros_like(sc['bev_map']))
    print('zero map rel', rel(D.unet_forward(usd, cfg, lat.reshape(-1,4,28,50), 981, ctx4, d4, m4), e), 'on d0', rel(d4[0], d[0]))
    print('no controlnet rel', rel(D.unet_forward(usd, cfg, lat.reshape(-1,4,28,50), 981, ctx, None, None), e))
    d5, m5, ctx5 = D.controlnet_forward(csd, cfg, lat, t, D.uncond_cam_param(csd, nb_, 6), sc['bboxes_3d_data'], sc['prompt_embeds'], sc['bev_map'])
    print('uncond cam rel', rel(D.unet_forward(usd, cfg, lat.reshape(-1,4,28,50), 981, ctx5, d5, m5), e))
    e6 = D.unet_forward(usd, cfg, lat.reshape(-1,4,28,50), 981, ctx*0.5, d, m); print('ctx halved rel', rel(e6,e))
