"""Build-time guard: no gfx950 kernel of the MFMA family may use scratch (register spills).  The 3x3 halo
kernels run at the 256-VGPR limit; an innocent-looking edit (e.g. an inline-asm bf16 pack in the epilogue)
once made hipcc spill 832 B/lane and slowed the whole image pass 2.5x without failing any numerical test."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="hipcc not available")
def test_mfma_kernels_have_no_scratch(tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    src = os.path.join(ROOT, "patchfusion_amd", "csrc", "igemm.hip")
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value", "-c", src,
                        "-o", str(tmp_path / "igemm.o"), "-Rpass-analysis=kernel-resource-usage", "-save-temps=obj"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    names = re.findall(r"Function Name: (\S+)", r.stderr)
    scratch = [int(x) for x in re.findall(r"ScratchSize \[bytes/lane\]: (\d+)", r.stderr)]
    assert len(names) == len(scratch) and len(names) >= 20
    bad = {n: s for n, s in zip(names, scratch) if s > 0}
    assert not bad, f"kernels with register spills: {bad}"

    # hand-counted LDS reads (inline-asm ds_read + s_waitcnt lgkmcnt(N) in the halo kernels): no instruction may touch
    # a destination register before the wait that releases it (tools/asm_lds_audit.py simulates the in-order queue)
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import asm_lds_audit
    listing = [f for f in os.listdir(tmp_path) if f.endswith("gfx950.s")]
    assert listing, os.listdir(tmp_path)
    n, viol = asm_lds_audit.audit(str(tmp_path / listing[0]))
    assert n >= 8 and not viol, viol[:5]


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="hipcc not available")
def test_winograd_transform_kernels_have_no_scratch(tmp_path):
    """the F(4x4,3x3) transforms hold a 6x6 tile of 4-channel vectors in registers (170 / 226 VGPRs): a spill would turn two
    HBM-bound passes into scratch-bound ones without failing any numerical test"""
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    src = os.path.join(ROOT, "patchfusion_amd", "csrc", "winograd.hip")
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value", "-c", src,
                        "-o", str(tmp_path / "winograd.o"), "-Rpass-analysis=kernel-resource-usage"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    names = re.findall(r"Function Name: (\S+)", r.stderr)
    scratch = [int(x) for x in re.findall(r"ScratchSize \[bytes/lane\]: (\d+)", r.stderr)]
    vgprs = [int(x) for x in re.findall(r"VGPRs: (\d+)", r.stderr)]
    assert len(names) == len(scratch) == 5, names        # input m = 2 / 4, input m = 4 with split-plane output, output m = 2 / 4
    assert not any(scratch), dict(zip(names, scratch))
    assert max(vgprs) <= 256, dict(zip(names, vgprs))


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="hipcc not available")
def test_fused_winograd_kernel_resources_and_hand_counted_vmem(tmp_path):
    """csrc/wino_fused.hip: 144 accumulator registers + filter fragments + a 6x6 transform tile at two waves per SIMD (256 registers):
    no scratch; and its inline-asm global loads / LDS-DMA with hand-counted s_waitcnt vmcnt(N) replayed over the emitted assembly
    (tools/asm_vm_audit.py): no instruction may touch a register that is still in flight, waves 0-3 keep the nine filter fragments of a
    chunk in flight (vmcnt(8)), the DMA waves 9 fragments + 5 DMA pieces (vmcnt(13), vmcnt(9)); both super-tile shapes."""
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    src = os.path.join(ROOT, "patchfusion_amd", "csrc", "wino_fused.hip")
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value", "-c", src,
                        "-o", str(tmp_path / "wino_fused.o"), "-Rpass-analysis=kernel-resource-usage", "-save-temps=obj"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    scratch = [int(x) for x in re.findall(r"ScratchSize \[bytes/lane\]: (\d+)", r.stderr)]
    vgprs = [int(x) for x in re.findall(r"VGPRs: (\d+)", r.stderr)]
    assert scratch == [0, 0] and vgprs and max(vgprs) <= 256, (scratch, vgprs)
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import asm_vm_audit
    listing = [f for f in os.listdir(tmp_path) if f.endswith("gfx950.s")]
    assert listing, os.listdir(tmp_path)
    report, viol = asm_vm_audit.audit(str(tmp_path / listing[0]))
    assert not viol, viol[:5]
    loops = [(what, st) for _, what, st, _ in report if what.startswith("loop") and st.get("asm_loads")]
    assert len(loops) == 4, report
    shapes = sorted((st["depth"], tuple(sorted(set(st["waits"])))) for _, st in loops)
    assert shapes == [(9, (8,)), (9, (8,)), (14, (9, 13)), (14, (9, 13))], shapes


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="hipcc not available")
@pytest.mark.parametrize("src_name,min_kernels", [("gemm_split3.hip", 8), ("vit.hip", 14), ("attn_split3.hip", 3), ("conv1x1_split3.hip", 1), ("swin.hip", 10)])
def test_split_precision_and_vit_kernels_have_no_scratch(tmp_path, src_name, min_kernels):
    """csrc/gemm_split3.hip (210 - 251 VGPRs: two 72-register fragment sets + accumulators at two waves per SIMD) and csrc/vit.hip (the split
    attention kernel runs three blocks per CU = 168 registers and spilled 12 B/lane in round 3): no scratch anywhere, at most 256 registers; csrc/attn_split3.hip (round 6: the pipelined attention holds Q, O, two S and two P
    register sets and two fragment sets -- 250 registers at two blocks per CU; its softmax slices must stay between its MFMAs)"""
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    src = os.path.join(ROOT, "patchfusion_amd", "csrc", src_name)
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value", "-Wno-unused-result", "-c", src,
                        "-o", str(tmp_path / "k.o"), "-Rpass-analysis=kernel-resource-usage", "-save-temps=obj"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    names = re.findall(r"Function Name: (\S+)", r.stderr)
    scratch = [int(x) for x in re.findall(r"ScratchSize \[bytes/lane\]: (\d+)", r.stderr)]
    vgprs = [int(x) for x in re.findall(r"VGPRs: (\d+)", r.stderr)]
    assert len(names) == len(scratch) == len(vgprs) and len(names) >= min_kernels, names
    assert not any(scratch), {n: s for n, s in zip(names, scratch) if s}
    assert max(vgprs) <= 256, dict(zip(names, vgprs))
    if src_name == "vit.hip":
        att = [v for n, v in zip(names, vgprs) if "vit_attention_split3_kernel" in n]
        assert att and att[0] <= 168, att          # three blocks per CU
    if src_name == "attn_split3.hip":
        # the steady step of the pipelined kernel must come out INTERLEAVED: between two barriers of the tile loop no run of more than 5 MFMAs without
        # another instruction between them (an un-fenced build put all 48 MFMAs first and the softmax after them: same results, no overlap)
        listing = [f for f in os.listdir(tmp_path) if f.endswith("gfx950.s")]
        assert listing, os.listdir(tmp_path)
        txt = open(tmp_path / listing[0]).read()
        body = txt[txt.index("vit_attention_split3_pipe_kernel"):]
        body = body[:body.index("s_endpgm")]
        ops = [ln.split()[0] for ln in body.splitlines() if ln.startswith("\t") and ln.strip() and not ln.strip().startswith((";", "."))]
        regions, cur = [], []
        for op in ops:
            if op == "s_barrier":
                regions.append(cur)
                cur = []
            else:
                cur.append(op)
        steady = [r for r in regions if sum(o.startswith("v_mfma_f32_32x32x16") for o in r) == 48 and sum(o == "v_exp_f32_e32" for o in r) >= 16]
        assert len(steady) >= 2, [sum(o.startswith("v_mfma") for o in r) for r in regions]
        for r in steady:
            run = worst = 0
            for o in r:
                run = run + 1 if o.startswith("v_mfma") else 0
                worst = max(worst, run)
            assert worst <= 6, worst
