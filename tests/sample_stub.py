"""TEST INFRASTRUCTURE for tests/test_distributed.py::test_sample_driver_two_ranks_gloo: a stand-in for the HIP pipeline that tools/sample.py loads
through its --pipe-factory hook, so the driver's multi-rank control flow (batch j -> rank j mod N, the reference's two seeding branches, the per-batch exchange, rank 0's
index) runs under gloo on a box without GPUs.  It samples nothing: a view's "image" is a 4 x 6 uint8 picture filled with a value derived from
the generator's seed, the camera index and the scene's BEV map, so a test can tell which rank / seed / scene produced a file."""
import numpy as np
import torch


class _Out:
    def __init__(self, images):
        self.images = images


class _Unet:
    cfg = {"cross_attention_dim": 8}


class StubPipe:
    text_encoder = None
    unet = _Unet()

    def __call__(self, generator=None, image=None, camera_param=None, prompt_embeds=None, **kw):
        g = generator[0] if isinstance(generator, (list, tuple)) else generator
        seed = 0 if g is None else int(g.initial_seed())
        draw = int(torch.randint(0, 97, [1], generator=g)) if g is not None else 0        # advances the generator like a sampler would
        out = []
        for bi in range(image.shape[0]):
            tag = int(image[bi].float().abs().sum().item() * 1000) % 251
            out.append([np.full((4, 6, 3), (seed * 7 + cam * 13 + tag + draw) % 256, dtype=np.uint8) for cam in range(camera_param.shape[1])])
        return _Out(out)


def make(ckpt, sd15, scheduler, device, given_view):
    return StubPipe()
