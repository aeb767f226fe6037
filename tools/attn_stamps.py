#!/usr/bin/env python
"""Where a (b, h) of the fused ViT attention backward spends its cycles: s_memtime stamps of wave 0 of every workgroup at entry,
after the first staging barrier, between the two parts and at exit (diagnostic build path of mha_bwd_fused_mfma_v_k; shares, not run time)."""
import sys, os, math, ctypes, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from garbage_classification_rca_amd import lib as L
lib = L.load()
B, H, S, dh = 256, 12, 197, 64
qkv = torch.randn(B * S, 3 * H * dh, device="cuda").bfloat16()
out = torch.empty(B * S, H * dh, device="cuda", dtype=torch.bfloat16)
dout = torch.randn_like(out)
dqkv = torch.empty_like(qkv)
lse = torch.empty(B, H, S, device="cuda")
sc = 1 / math.sqrt(dh)
L.mha_fwd(qkv, None, out, lse, B, H, S, dh, sc, L.BF16)
for _ in range(3):
    L.mha_bwd(qkv, None, out, dout, lse, dqkv, B, H, S, dh, sc, L.BF16)
st = torch.zeros(B * H * 4, dtype=torch.int64, device="cuda")
lib.mmrca_debug_attn_stamps(ctypes.c_void_p(st.data_ptr()))
L.mha_bwd(qkv, None, out, dout, lse, dqkv, B, H, S, dh, sc, L.BF16)
torch.cuda.synchronize()
lib.mmrca_debug_attn_stamps(None)
t = st.view(-1, 4).double().cpu()
d = (t[:, 1:] - t[:, :-1])
print("cycles per workgroup (wave 0; mean / median): first staging (K|V + row constants) %.0f / %.0f   first part (dQ) %.0f / %.0f   restaging (Q|dO) + second part (dK,dV) %.0f / %.0f   total %.0f" % (
    d[:, 0].mean(), d[:, 0].median(), d[:, 1].mean(), d[:, 1].median(), d[:, 2].mean(), d[:, 2].median(), (t[:, 3] - t[:, 0]).mean()))
print("(default form MMRCA_ATTN_BWD_FUSED=1: two 8-wave workgroups per CU; =2: one 16-wave workgroup, all four images resident, parts = dQ then dK,dV without restaging)")
