"""BEVControlNetOutput (magicdrive/networks/output_cls.py)."""
from dataclasses import dataclass
from typing import List

import torch


@dataclass
class BEVControlNetOutput:
    down_block_res_samples: List[torch.Tensor]
    mid_block_res_sample: torch.Tensor
    encoder_hidden_states_with_cam: torch.Tensor
