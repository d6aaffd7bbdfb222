"""TEST INFRASTRUCTURE: a LARGE-DYNAMIC-RANGE variant of a synthetic state dict (round-3 review: every parity number was measured on O(1)-scaled
activations; real Depth-Anything / PatchFusion activations span several decades per channel).

`widen_dynamic_range(sd)` rescales pairs of layers so that the FUNCTION is unchanged in exact arithmetic while the tensors between the two layers of
a pair carry per-channel scales drawn log-uniformly from [10^lo, 10^hi] (default 1e-3 ... 1e3):
  * ViT blocks (both branches): q channel c x s_c, k channel c / s_c (every q.k product unchanged), v channel c x t_c, projection input column c / t_c;
  * ResidualConvUnits of the DPT heads (blocks.py:69-92): conv1 output channel x s_c (weight and bias), conv2 input channel / s_c (ReLU is
    positively homogeneous);
  * the double convolutions of the guided-fusion U-Net (guided_fusion_model.py:34-69): first conv's output (its BatchNorm affine, or its bias) x s_c,
    second conv's input channel / s_c.
The split-precision planes (q / k / v, attention output), the Winograd transforms and every epilogue then see operands whose magnitude varies over
six decades across channels; engine and oracle are compared on the SAME rescaled weights.
"""
import re

import torch


def widen_dynamic_range(sd, seed=0, lo=-3.0, hi=3.0):
    out = {k: v.clone() for k, v in sd.items()}
    g = torch.Generator().manual_seed(seed)

    def scales(n):
        return 10.0 ** (torch.rand(n, generator=g, dtype=torch.float64) * (hi - lo) + lo)

    touched = 0
    for k in sorted(out):
        if re.search(r"\.pretrained\.blocks\.\d+\.attn\.qkv\.weight$", k):
            p = k[:-len("qkv.weight")]
            W, b, Wp = out[k].double(), out[p + "qkv.bias"].double(), out[p + "proj.weight"].double()
            D = W.shape[1]
            s, t = scales(D), scales(D)
            W[:D] *= s[:, None]
            b[:D] *= s
            W[D:2 * D] /= s[:, None]
            b[D:2 * D] /= s
            W[2 * D:] *= t[:, None]
            b[2 * D:] *= t
            Wp /= t[None, :]
            out[k], out[p + "qkv.bias"], out[p + "proj.weight"] = W.float(), b.float(), Wp.float()
            touched += 1
        elif re.search(r"resConfUnit[12]\.conv1\.weight$", k):
            p = k[:-len("conv1.weight")]
            s = scales(out[k].shape[0])
            out[k] = (out[k].double() * s[:, None, None, None]).float()
            out[p + "conv1.bias"] = (out[p + "conv1.bias"].double() * s).float()
            out[p + "conv2.weight"] = (out[p + "conv2.weight"].double() / s[None, :, None, None]).float()
            touched += 1
        elif re.search(r"guided_fusion\..*double_conv\.0\.weight$", k):
            p = k[:-len("0.weight")]
            s = scales(out[k].shape[0])
            if p + "1.running_mean" in out:                      # conv -> BN -> ReLU -> conv: scale the BN affine
                out[p + "1.weight"] = (out[p + "1.weight"].double() * s).float()
                out[p + "1.bias"] = (out[p + "1.bias"].double() * s).float()
                nxt = p + "3.weight"
            else:                                                # conv(+bias) -> ReLU -> conv
                out[k] = (out[k].double() * s[:, None, None, None]).float()
                out[p + "0.bias"] = (out[p + "0.bias"].double() * s).float()
                nxt = p + "2.weight"
            out[nxt] = (out[nxt].double() / s[None, :, None, None]).float()
            touched += 1
    assert touched > 0
    return out
