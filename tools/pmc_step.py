"""Driver for rocprofv3 counter passes: a few fused training steps (toad_mil_step_f32) on one resident bag, so that every
kernel of the step appears with its real operands (abs-max arrays, pooling addend, ...). Usage (one --pmc set per pass):
    rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES \
              SQ_INSTS_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d OUT -o p -- python tools/pmc_step.py [N] [steps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from toad_amd import TOAD_fc_mtl_concat
from toad_amd.dp import SlideShardedDP
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = torch.device("cuda:0")
torch.manual_seed(1)
model = TOAD_fc_mtl_concat(n_classes=18); model.relocate(); model.train()
dp = SlideShardedDP(model, {"lr": 1e-4, "weight_decay": 1e-5})
g = torch.Generator(device=dev).manual_seed(1000)
slide = (torch.randn(n, 1024, device=dev, generator=g), torch.tensor([1.0], device=dev), torch.tensor([3], device=dev), torch.tensor([1], device=dev))
if os.environ.get("TOAD_BAG", "fp32") == "prepared":
    from toad_amd import ops
    slide = (ops.prepare_bag(slide[0]),) + slide[1:]
for _ in range(steps):
    dp.step([slide], 1)
torch.cuda.synchronize()
