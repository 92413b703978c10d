"""kk_bucket_embed_add_bwd at the step's shapes: us per launch, graph replays over 6 operand sets (random and speech-like bin sequences)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from kokoro_ruslan_amd import lib as kk
kk.use_library("tuning")
dev = "cuda"
def gtime(fns, reps=20):
    def run():
        for f in fns: f()
    run(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g): run()
    for _ in range(3): g.replay()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): g.replay()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps / len(fns) * 1e3
H, nb = 512, 256
for B, T, smooth in ((8, 512, 0), (8, 512, 1), (8, 1024, 0), (16, 700, 0)):
    fns, sorts, segs = [], [], []
    for i in range(6):
        d = torch.randn(B * T, H, device=dev)
        if smooth:                                     # neighbouring frames in neighbouring bins (speech-like contours)
            base = torch.cumsum(torch.randn(B * T, device=dev) * 2, 0)
            pi = (base.long() % nb).int(); ei = ((base * 0.7).long() % nb).int()
        else:
            pi, ei = torch.randint(0, nb, (B * T,), device=dev, dtype=torch.int32), torch.randint(0, nb, (B * T,), device=dev, dtype=torch.int32)
        fm = (torch.rand(B * T, device=dev) < 0.1).to(torch.uint8)
        gp, ge = torch.zeros(nb, H, device=dev), torch.zeros(nb, H, device=dev)
        fns.append(lambda d=d, pi=pi, ei=ei, fm=fm, gp=gp, ge=ge: kk.call("kk_bucket_embed_add_bwd", d, pi, ei, fm, gp, ge, B, T, H, nb))
        order = torch.empty(2, B * T, dtype=torch.int32, device=dev)
        items = torch.empty(2, kk.load().kk_bucket_sort_items(B * T, nb), 4, dtype=torch.int32, device=dev)
        sorts.append(lambda pi=pi, ei=ei, fm=fm, order=order, items=items: kk.call("kk_bucket_sort", pi, ei, fm, B * T, nb, order, items))
        sorts[-1]()
        segs.append(lambda d=d, order=order, items=items, gp=gp, ge=ge: kk.call("kk_bucket_embed_add_bwd_sorted", d, order, items, gp, ge, B * T, H, nb))
    print(f"rows {B*T} smooth {smooth}: LDS form {gtime(fns):7.2f} us   segmented {gtime(segs):7.2f} us   (+ kk_bucket_sort in the forward {gtime(sorts):7.2f} us)", flush=True)
