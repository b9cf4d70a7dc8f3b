"""First command of every gpurun script: exit non-zero at once when the box's GPU is not usable (a box whose first HIP call
aborts otherwise burns the whole time limit of the call in per-command timeouts)."""
import sys

import torch

try:
    assert torch.cuda.is_available(), "no GPU visible"
    x = torch.ones(1 << 20, device="cuda:0")
    assert float((x * 2).sum()) == float(2 << 20)
    torch.cuda.synchronize()
    print("gpu ok:", torch.cuda.get_device_name(0))
except Exception as ex:  # noqa: BLE001
    print("gpu NOT usable:", ex)
    sys.exit(3)
