"""Grid-wide timeline of the attention launches (KK_ATTN_DBG=4096, tools flavour): every workgroup's (entry, exit) on the constant 100 MHz
clock and its CU — dispatch ramp, spread of workgroup durations, tail — beside the launch's duration by events.
    python tools/probes/attn_grid_timeline.py [S] [causal]"""
import os, sys, ctypes, torch
os.environ["KK_ATTN_DBG"] = "4096"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from kokoro_ruslan_amd import lib as kk
from kokoro_ruslan_amd.spec import rope_tables
kk.use_library("tuning")
S = int(sys.argv[1]) if len(sys.argv) > 1 else 512
causal = int(sys.argv[2]) if len(sys.argv) > 2 else 0
B, h, H, P = 8, 8, 512, 0.2
bf, dev = torch.bfloat16, "cuda"
q, kv = torch.randn(B * S, H, device=dev).to(bf), torch.randn(B * S, 2 * H, device=dev).to(bf)
raw_q, raw_kv, do = torch.randn_like(q), torch.randn_like(kv), torch.randn(B * S, H, device=dev).to(bf)
o, lse = torch.empty_like(q), torch.empty(B, h, S, device=dev)
seed = torch.tensor([7], dtype=torch.int32, device=dev)
keep = torch.empty(kk.load().kk_attn_keep_bytes(B, h, S, S), dtype=torch.uint8, device=dev)
NW = 4096
buf = torch.zeros(512 + 4 * NW, dtype=torch.int64, device=dev)
kk._tuning_hook("kk_attn_trace")(ctypes.c_void_p(buf.data_ptr()))
fwd = lambda: kk.call("kk_attn_fwd_kb", q, kv, kv[:, H:], o, lse, B, h, S, S, H, 2 * H, 2 * H, H, None, causal, 0.125, seed, 5, P, 1, 1, keep)
fwd_rb = lambda: kk.call("kk_attn_fwd_rb", q, kv, kv[:, H:], o, lse, B, h, S, S, H, 2 * H, 2 * H, H, None, causal, 0.125, seed, 5, P, 1, 1, keep)
fwd()
delta = torch.empty(B, h, S, device=dev)
kk.call("kk_attn_delta", o, do, delta, B, h, S, H, H, 1)
nb = kk.load().kk_attn_bwd_blocks(B, h, S)
gains = [torch.ones(64, device=dev) for _ in range(3)]
c, s = (t.cuda() for t in rope_tables(S, 64))
pq, pkv = torch.zeros(1, nb, 64, device=dev), torch.zeros(2, nb, 64, device=dev)
hq = kk.attn_headnorm([(raw_q, gains[0], pq[0], c, s)])
hkv = kk.attn_headnorm([(raw_kv, gains[1], pkv[0], c, s), (raw_kv[:, H:], gains[2], pkv[1], None, None)])
dq, dkv = torch.empty_like(q), torch.empty_like(kv)
a = (q, kv, kv[:, H:], do, lse, delta, dq, dkv, dkv[:, H:], B, h, S, S, H, 2 * H, 2 * H, H, H, 2 * H, 2 * H, None, causal, 0.125, seed, 5, P, 1, 1, hq, hkv)
bwd = lambda: kk.call("kk_attn_bwd_kb", *a, keep)


def reset():
    v = buf.view(-1)[512:].view(NW, 4)
    v.zero_()
    v[:, 0] = torch.iinfo(torch.int64).max


def pct(x, p):
    x = sorted(x)
    return x[min(len(x) - 1, int(p * len(x)))]


for name, run in (("forward storing keep bits", fwd), ("forward reading keep bits", fwd_rb), ("backward pair reading keep bits", bwd)):
    for _ in range(3):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(20):
        run()
    e1.record()
    torch.cuda.synchronize()
    per = e0.elapsed_time(e1) * 1000 / 20
    reset()
    torch.cuda.synchronize()
    run()
    torch.cuda.synchronize()
    t = buf.cpu()[512:].view(NW, 4)
    rows = [(i, int(t[i, 0]), int(t[i, 1]), int(t[i, 2])) for i in range(NW) if int(t[i, 1]) != 0]
    t0 = min(r[1] for r in rows)
    us = lambda ticks: ticks / 100.0
    starts = [us(r[1] - t0) for r in rows]
    ends = [us(r[2] - t0) for r in rows]
    durs = [us(r[2] - r[1]) for r in rows]
    print(f"== {kk.last_kernel()}  ({name}) S={S} causal={causal}: {len(rows)} workgroups, back-to-back launches {per:.1f} us each")
    print(f"   entries after the first: median {pct(starts, .5):.2f}  p90 {pct(starts, .9):.2f}  last {max(starts):.2f} us;   exits: first {min(ends):.2f}  median {pct(ends, .5):.2f}  p90 {pct(ends, .9):.2f}  last {max(ends):.2f} us")
    n = len(rows)
    groups = (("all", rows),) if "pair" not in kk.last_kernel() else (("first half (z = 0: dK/dV)", rows[: n // 2]), ("second half (z = 1: dQ)", rows[n // 2:]))
    for gname, g in groups:
        d = [us(r[2] - r[1]) for r in g]
        st = [us(r[1] - t0) for r in g]
        print(f"   {gname}: duration min {min(d):.2f}  median {pct(d, .5):.2f}  p90 {pct(d, .9):.2f}  max {max(d):.2f} us; entries {min(st):.2f} .. {max(st):.2f} us")
    cus = {}
    for r in rows:
        hw = r[3] & 0xFFFFFFFF
        key = (r[3] >> 32 & 0xF, (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 15)
        cus.setdefault(key, []).append(r)
    per_cu = sorted(len(v) for v in cus.values())
    busy = [us(max(r[2] for r in v) - min(r[1] for r in v)) for v in cus.values()]
    print(f"   {len(cus)} CUs hold workgroups ({per_cu[0]} .. {per_cu[-1]} each); a CU's first entry to last exit: median {pct(busy, .5):.2f}  max {max(busy):.2f} us")
    if causal and "pair" in kk.last_kernel():
        nx = (S + 127) // 128
        for z, nm in ((0, "dK/dV"), (1, "dQ")):
            for x in range(nx):
                d = [us(r[2] - r[1]) for r in rows if r[0] // (nx * B * h) == z and r[0] % nx == x]
                print(f"      {nm} block {x}: median {pct(d, .5):.2f}  max {max(d):.2f} us")
    if "pair" in kk.last_kernel() and len(rows) == 512:
        half = len(rows) // 2
        same = sum(1 for v in cus.values() if len(v) == 2 and abs(v[0][0] - v[1][0]) == half)
        mixed = sum(1 for v in cus.values() if len(v) == 2 and (v[0][0] < half) != (v[1][0] < half))
        print(f"   CUs holding one workgroup of each half: {mixed}; holding workgroups i and i + {half}: {same}")
        if causal:
            spans = sorted((us(max(r[2] for r in v) - min(r[1] for r in v)), sorted(r[0] for r in v), [round(us(r[2] - r[1]), 1) for r in sorted(v)]) for v in cus.values())
            print("   shortest CUs:", spans[:4])
            print("   longest CUs:", spans[-4:])
