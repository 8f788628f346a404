import ctypes as C, sys, os, torch
sys.path.insert(0, "/root/repo")
from rich_text_to_image_amd.engine import _ptr, load_library
lib = load_library(); dev = "cuda:0"
for (B, H, N) in ((4, 2, 1024), (7, 20, 1024)):
    HD, P = H * 64, 3
    g = torch.Generator(device=dev).manual_seed(1)
    bf = lambda *s: torch.randn(*s, generator=g, device=dev).to(torch.bfloat16)
    Q, K, VT = bf(B * N, HD), bf(P * 96, HD), bf(HD, P * 96)
    K.view(P, 96, HD)[:, 77:] = 0; VT.view(HD, P, 96)[:, :, 77:] = 0
    wabs = torch.zeros(2, 96, device=dev); wabs[:, :77] = 1.0; wsgn = torch.ones(2, 96, device=dev)
    def run(Bn):
        O = torch.zeros(Bn * N, HD, device=dev, dtype=torch.bfloat16)
        ia = lambda v: (C.c_int * Bn)(*v)
        src, prm, ws = ia(range(Bn)), ia([0, 2, 2, 1, 0, 1, 2][:Bn]), ia([-1] * Bn)
        rc = lib.rt_op_attention(_ptr(Q), HD, _ptr(K), HD, _ptr(VT), P * 96, _ptr(O), HD, src, prm, prm, ws, _ptr(wabs), _ptr(wsgn), Bn, H, N, 96, 77, 64, 1, None)
        assert rc == 0
        torch.cuda.synchronize()
        return O
    outs = [run(B) for _ in range(6)]
    print((B, H, N), "repeat equal:", [bool(torch.equal(outs[0], o)) for o in outs[1:]], "alone vs batch:", bool(torch.equal(run(1), outs[0][:N])),
          "max diff", max(float((outs[0].float() - o.float()).abs().max()) for o in outs[1:]), "nan", bool(torch.isnan(outs[0].float()).any()))
