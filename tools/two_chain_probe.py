"""Does running the 7 streams of a step as TWO concurrent chains (3 + 4 streams, separate HIP streams) beat one batch-7 forward?
Two engines with the same random weights, one Python thread each (ctypes releases the GIL)."""
import sys, time, threading, torch
sys.path.insert(0, '/root/repo')
from rich_text_to_image_amd.engine import Engine, SDXL_CONFIG
DEV = 'cuda:0'
P = 5
def mk():
    e = Engine(SDXL_CONFIG, 128, 128, device=0, max_streams=8, max_prompts=8)
    e.init_random_weights(0)
    e.set_prompts(torch.randn(P, 77, 2048, device=DEV), torch.randn(P, 1280, device=DEV), torch.tensor([[1024., 1024, 0, 0, 1024, 1024]]))
    e.set_fontsize(torch.tensor([5, 6]), torch.tensor([20.0, 20.0]))
    return e
x = torch.randn(7, 4, 128, 128, device=DEV)
full = dict(prompt_idx=[0, 4, 0, 4, 1, 2, 3], fontsize=[0, 1, 0, 0, 0, 0, 0], qk_src=[0, 1, 2, 3, 3, 3, 3], res_src=[-1, -1, -1, -1, 3, 3, 3])
chainA = dict(prompt_idx=[0, 4, 0], fontsize=[0, 1, 0], qk_src=[0, 1, 2], res_src=[-1, -1, -1])                 # uncond, base, uncond_ref
chainB = dict(prompt_idx=[4, 1, 2, 3], fontsize=[0, 0, 0, 0], qk_src=[0, 0, 0, 0], res_src=[-1, 0, 0, 0])       # text_ref + 3 regions
e1, e2 = mk(), mk()
def run(e, xs, kw, n):
    for _ in range(n):
        e.unet_forward(xs, 801.0, **kw)
def timeit(fn, n=6):
    fn(2); torch.cuda.synchronize(); t0 = time.perf_counter(); fn(n); torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
t_full = timeit(lambda n: run(e1, x, full, n))
t_a = timeit(lambda n: run(e1, x[:3], chainA, n))
t_b = timeit(lambda n: run(e2, x[3:], chainB, n))
def both(n):
    ta = threading.Thread(target=run, args=(e1, x[:3], chainA, n)); tb = threading.Thread(target=run, args=(e2, x[3:], chainB, n))
    ta.start(); tb.start(); ta.join(); tb.join()
t_ab = timeit(both)
print(f"one batch-7 forward {t_full:.1f} ms | chain A alone (3) {t_a:.1f} ms | chain B alone (4, injected) {t_b:.1f} ms | A and B concurrently {t_ab:.1f} ms")
