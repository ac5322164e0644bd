"""TEST INFRASTRUCTURE for tests/test_distributed.py::test_bench_two_ranks_gloo_prints_one_json_line: a stand-in for the HIP pipeline that
bench.py loads through its --pipe-factory hook, so the script's multi-rank control flow (scene sharding, the result gather, barrier +
max-over-ranks timing, process-group teardown on every rank before rank 0 prints) runs under gloo on a box without GPUs.  It samples
nothing: a scene's "latents" are its input latents scaled by 2."""
import torch


class _Out:
    def __init__(self, images):
        self.images = images


class _Net:
    def num_parameters(self):
        return 0


class StubPipe:
    use_graph = True
    streams = 1

    def __init__(self):
        self._plans = {}
        self.calls = 0

    def __call__(self, prompt=None, image=None, camera_param=None, latents=None, **kw):
        self.calls += 1
        assert kw["output_type"] == "latent" and latents.dim() == 4 and image.shape[0] == latents.shape[0]
        return _Out(torch.stack([latents.float() * 2.0] * 6, dim=1))


def make(cfg, device):
    return StubPipe(), _Net(), _Net()
