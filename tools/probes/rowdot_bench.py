"""kk_rowdot_bwd at the step's shapes (KK_ROWDOT_VEC=0/1 in the tools flavour): us per launch, graph replays over 6 operand sets."""
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
for rows, C, L, xbf, want_dx, msk in ((4096, 256, 512, 0, 1, 1), (4096, 512, 512, 1, 1, 0), (512, 256, 64, 0, 1, 1), (8192, 256, 1024, 0, 1, 1), (8192, 512, 1024, 1, 1, 0)):
    fns = []
    for i in range(6):
        x = torch.randn(rows, C, device=dev); x = x.bfloat16() if xbf else x
        w, dout = torch.randn(C, device=dev), torch.randn(rows, device=dev)
        mask = (torch.rand(rows, device=dev) < 0.2).to(torch.uint8) if msk else None
        dx, dw, db = torch.empty(rows, C, device=dev), torch.zeros(C, device=dev), torch.zeros(1, device=dev)
        part = torch.empty(kk.load().kk_rowdot_bwd_blocks(rows), C + 4, device=dev) if os.environ.get("PART") == "1" else None
        fns.append(lambda x=x, w=w, dout=dout, mask=mask, dx=dx, dw=dw, db=db, part=part: kk.call("kk_rowdot_bwd", dout, x, w, mask, dx if want_dx else None, dw, db, rows, C, L, 0, xbf, part))
    print(f"KK_ROWDOT_VEC={os.environ.get('KK_ROWDOT_VEC','1')} partial rows={os.environ.get('PART','0')} rows {rows} C {C} x_bf16 {xbf}: {gtime(fns):7.2f} us", flush=True)
