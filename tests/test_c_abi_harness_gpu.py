"""The C ABI without PyTorch: examples/abi_smoke.cpp (hipMalloc + include/vitpe.h + libvitpe.so) is compiled on the
GPU box and its block-forward checksum is compared with the same computation issued through the Python host."""
import math
import os
import re
import shutil
import subprocess

import pytest
import torch

from conftest import REPO

pytestmark = pytest.mark.gpu


def _wave(n, salt, amp):
    # float32 sinf(0.37f * i + salt) as the C program computes it
    i = torch.arange(n, dtype=torch.float32)
    return amp * torch.sin(torch.tensor(0.37, dtype=torch.float32) * i + torch.tensor(salt, dtype=torch.float32))


def test_standalone_c_caller_matches_python_host(tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available on this box")
    lib_dir = os.path.join(REPO, "vit-rpe-rope_amd", "lib")
    exe = str(tmp_path / "abi_smoke")
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O2", "-I" + os.path.join(REPO, "include"), os.path.join(REPO, "examples", "abi_smoke.cpp"),
                    "-L" + lib_dir, "-lvitpe", "-Wl,-rpath," + lib_dir, "-o", exe], check=True, capture_output=True, timeout=600)
    run = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert run.returncode == 0, run.stdout + run.stderr
    m = re.search(r"abi_smoke ok sum=(\S+) wsq=(\S+) first=(\S+) last=(\S+)", run.stdout)
    assert m, run.stdout
    c_sum, c_wsq, c_first, c_last = (float(v) for v in m.groups())
    assert "unsupported geometry -> 801" in run.stdout

    # the same block forward through the Python host (same kernels, same inputs)
    from vitpe import kernels as K
    B, N, D, H, HD, HID, G = 4, 65, 192, 6, 32, 768, 8
    M = B * N
    bf = torch.bfloat16
    x = _wave(M * D, 0.1, 1.0).view(M, D).to(bf).cuda()
    wqkv = _wave(3 * D * D, 0.7, 0.08).view(3 * D, D).cuda()
    wp = _wave(D * D, 1.3, 0.07).view(D, D).cuda()
    w1 = _wave(HID * D, 2.1, 0.07).view(HID, D).cuda()
    w2 = _wave(D * HID, 2.9, 0.04).view(D, HID).cuda()
    g1, b1 = (1 + _wave(D, 3.3, 0.1)).cuda(), _wave(D, 3.9, 0.1).cuda()
    g2, b2 = (1 + _wave(D, 4.4, 0.1)).cuda(), _wave(D, 5.0, 0.1).cuda()
    bp, bf2_, bf1_ = _wave(D, 5.5, 0.05).cuda(), _wave(D, 6.1, 0.05).cuda(), _wave(HID, 6.6, 0.05).cuda()
    inv = torch.tensor([1.0 / math.pow(100.0, 4 * i / HD) for i in range(HD // 4)], dtype=torch.float32).cuda()
    pe = K.PETables("rope-axial", G)
    pe.cos, pe.sin = K.rope_axial_tables(inv, G)
    _, m1, r1 = K.layernorm_fwd(x.view(B, N, D), g1, b1, stats_only=True)
    att = K.fused_attention_fwd_wide(x.view(B, N, D), K.pack_qkv_weights_wide(wqkv, bf, H), H, pe, ln=(g1, b1, m1, r1))
    out = K.block_tail2_fwd(att.view(M, D), x, K.pack_weight_frags(wp, bf, 192, 0), bp, g2, b2, K.pack_weight_frags(w1, bf, 192, 1), bf1_,
                            K.pack_weight_frags(w2, bf, 32, 1), bf2_)[0].float().cpu().double().flatten()
    w = torch.tensor([(i % 7) + 1 for i in range(out.numel())], dtype=torch.float64)
    assert abs(float(out.sum()) - c_sum) <= 1e-3 * max(1.0, abs(c_sum)) + 0.5     # host-side sinf vs torch.sin: last-ulp input differences
    assert abs(float((out * out * w).sum()) - c_wsq) <= 2e-3 * c_wsq
    assert abs(float(out[0]) - c_first) < 0.05 and abs(float(out[-1]) - c_last) < 0.05
