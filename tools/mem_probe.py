"""Where the device memory of one ViT-L 4K image pass goes (torch allocator figures, GiB): parameters, packed engine, Winograd arenas, activations.
usage: python tools/mem_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def gib(x):
    return f"{x / 2**30:.2f}"


def main():
    from patchfusion_amd import hip_ops
    from patchfusion_amd.config import make_config
    from patchfusion_amd.model import PatchFusion
    from patchfusion_amd.spec import patchfusion_spec, synthetic_state_dict
    dev = torch.device("cuda", 0)
    cfg = make_config("vitl", (392, 518), (2160, 3840), (4, 4))
    sd = synthetic_state_dict(patchfusion_spec(cfg), 0)
    print("| stage | allocated GiB | reserved GiB | peak allocated GiB |\n|---|---|---|---|")

    def row(name):
        torch.cuda.synchronize()
        print(f"| {name} | {gib(torch.cuda.memory_allocated())} | {gib(torch.cuda.memory_reserved())} | {gib(torch.cuda.max_memory_allocated())} |", flush=True)

    img = torch.rand(1, 3, 2160, 3840, generator=torch.Generator().manual_seed(1234)).to(dev)
    m = PatchFusion(cfg, compute_dtype="fp32").eval()
    m.load_state_dict(sd, strict=True)
    m = m.to(dev)
    row("model.to(device): nn.Parameters (f32 checkpoints) + image")
    m._ensure_engine()
    row("engine built (packed f32 layers, split planes, Winograd filters)")
    lr = m.resizer(img)
    for i in range(2):
        torch.cuda.reset_peak_memory_stats()
        d, _ = m(mode="infer", image_lr=lr, image_hr=img, cai_mode="m1", process_num=8)
        row(f"after image pass {i} (workspaces {gib(hip_ops.workspace_bytes())} GiB)")
    if hasattr(m, "free_parameters"):
        m.free_parameters()
        torch.cuda.empty_cache()
        torch.cuda.reset_peak_memory_stats()
        d2, _ = m(mode="infer", image_lr=lr, image_hr=img, cai_mode="m1", process_num=8)
        row(f"after free_parameters() + one more pass (max |d - d2| = {float((d - d2).abs().max()):.1e})")


if __name__ == "__main__":
    main()
