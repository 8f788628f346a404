import sys, time, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from rich_text_to_image_amd.engine import Engine, SDXL_CONFIG
DEV='cuda:0'
eng = Engine(SDXL_CONFIG, 128, 128, device=0, max_streams=16, max_prompts=8)
eng.init_random_weights(0)
P=5
eng.set_prompts(torch.randn(P,77,2048,device=DEV), torch.randn(P,1280,device=DEV), torch.tensor([[1024.,1024,0,0,1024,1024]]))
eng.set_fontsize(torch.tensor([5,6]), torch.tensor([20.0,20.0]))
x = torch.randn(14,4,128,128,device=DEV)
def t(B, iters=4):
    pr=([0,4,0,4,1,2,3]*2)[:B]
    f=lambda: eng.unet_forward(x[:B], 801.0, pr)
    f(); f()
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(iters): f()
    torch.cuda.synchronize(); return (time.perf_counter()-t0)/iters*1e3
for B in (7,14,4,10):
    ms=t(B); print(f"B={B}: {ms:.1f} ms  -> {ms/B:.2f} ms per stream, {B*6.7612/ms:.0f} TF")
