"""Tile -> XCD order of the gemm16 launches (rt_op_gemm16_variant's w_stationary argument: 0 = groups of 4 tile rows x all tile columns, 1 = every XCD
owns a share of the W columns, 2 = groups of 8 tile rows) on the step's shapes: time per launch, bit-identity.  TILE_WS=<n> restricts the run to
one order and the two GEGLU shapes (the target of a rocprofv3 --pmc FETCH_SIZE pass: profiles/r6_tile_order.txt)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rich_text_to_image_amd.engine import load_library, _ptr
lib = load_library(); DEV = "cuda:0"
def run(M, N, K, epi, v, ws, res=None):
    g = torch.Generator().manual_seed(1)
    A = torch.randn(M, K, generator=g).to(DEV).to(torch.bfloat16); W = (torch.randn(N, K, generator=g) * K ** -0.5).to(DEV).to(torch.bfloat16)
    bias = torch.randn(N, generator=g).to(DEV)
    out = torch.empty(M, N // 2 if epi == 3 else N, device=DEV, dtype={4: torch.float16}.get(epi, torch.bfloat16))
    r = torch.randn(M, N, generator=g).to(DEV).to(torch.float16) if res else None
    def go(w):
        rc = lib.rt_op_gemm16_variant(_ptr(A), _ptr(W), _ptr(bias), _ptr(out), _ptr(r), epi, M, N, K, K, K, out.stride(0), r.stride(0) if r is not None else 0, 0, v, w, None); assert rc == 0
    best = {}
    outs = {}
    for rnd in range(4):
        for w in ws:
            go(w); go(w)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): go(w)
            e1.record(); torch.cuda.synchronize()
            best[w] = min(best.get(w, 1e9), e0.elapsed_time(e1) / 20 * 1e3)
            outs[w] = out.clone()
    same = all(torch.equal(outs[w], outs[ws[0]]) for w in ws)
    print(f"{M}x{N}x{K} epi {epi} v{v}: " + "  ".join(f"wstat {w}: {best[w]:.1f} us" for w in ws) + f"   identical {same}", flush=True)
ONLY = os.environ.get("TILE_WS")
if ONLY is not None:
    run(28672, 5120, 640, 3, 3, (int(ONLY),)); run(7168, 10240, 1280, 3, 2, (int(ONLY),)); sys.exit(0)
run(28672, 5120, 640, 3, 3, (0, 2))       # GEGLU 640 level
run(28672, 640, 2560, 4, 4, (0, 2), True)  # ff.net.2 640 level
run(28672, 640, 640, 4, 4, (0, 2), True)   # to_out 640
run(7168, 1280, 5120, 4, 0, (0, 2), True)  # ff.net.2 1280
run(7168, 1280, 1280, 4, 0, (0, 2), True)  # to_out 1280
run(7168, 2560, 1280, 0, 4, (0, 2))        # Q|K 1280
run(7168, 10240, 1280, 3, 2, (1, 0, 2))    # GEGLU 1280
