import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kokoro_ruslan_amd import lib as kk
rows, H = 4096, 512
x = torch.randn(rows, H, device="cuda"); dy = torch.randn(rows, H, device="cuda").bfloat16()
g = torch.ones(H, device="cuda"); mean = torch.zeros(rows, device="cuda"); rstd = torch.ones(rows, device="cuda")
dx = torch.zeros(rows, H, device="cuda"); dg = torch.zeros(H, device="cuda"); db = torch.zeros(H, device="cuda")
PART = torch.zeros(256, 2 * H, device="cuda") if os.environ.get("KK_PART") else None
def f(): kk.call("kk_layernorm_bwd", dy, x, g, mean, rstd, dx, 1, dg, db, PART, rows, H, 1)
for _ in range(5): f()
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(50): f()
e.record(); torch.cuda.synchronize()
print(os.environ.get("KK_LN_WPB"), os.environ.get("KK_LN_TRIPS"), "us", s.elapsed_time(e) / 50 * 1e3)
