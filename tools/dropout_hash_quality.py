#!/usr/bin/env python3
"""Statistical screen of candidate attention-dropout hashes on the (query, key/2) counter lattice the kernels use
(numpy, CPU): keep rate, correlation between neighbouring keys / queries / diagonals, row and column keep-count variance
against the binomial, chi-square of 8-key patterns.  lowbias32 (two 32-bit multiplies, quarter rate on CDNA) and the
two-round 24-bit-multiplier variant the kernels use ("u24 x2") are indistinguishable here; single-multiply and
64-bit-product shortcuts fail badly (kept as a record of what not to use)."""
import numpy as np
def lowbias32(x):
    x = x.astype(np.uint64)
    x ^= x >> 16; x = (x * 0x7feb352d) & 0xffffffff; x ^= x >> 15; x = (x * 0x846ca68b) & 0xffffffff; x ^= x >> 16
    return x.astype(np.uint32)
def candB(x, key):
    # one 32x32->64 multiply, 4 fields of 16 bits
    x = (x ^ key).astype(np.uint64)
    x ^= x >> 16
    y = x * np.uint64(0x9E3779B1)
    hi = (y >> np.uint64(32)) & np.uint64(0xffffffff); lo = y & np.uint64(0xffffffff)
    hi ^= lo >> np.uint64(7)          # fold
    lo ^= hi << np.uint64(9) & np.uint64(0xffffffff)
    lo &= np.uint64(0xffffffff)
    return [(hi & 0xffff), (hi >> 16) & 0xffff, (lo >> 16) & 0xffff, ((lo ^ (lo >> 13)) & 0xffff)]
def candC(x, key):
    # two full-rate-ish rounds: xorshift-multiply once (1 mul) + second cheap round with mul_u24
    x = (x ^ key).astype(np.uint64)
    x ^= x >> 16; x = (x * 0x7feb352d) & 0xffffffff; x ^= x >> 15
    return [x & 0xffff, (x >> 16) & 0xffff]
def stats(fields, shape, thr, name):
    # fields: list of arrays of 16-bit values laid over (q, slot); build mask matrix [q, key]
    Q, S = shape
    nf = len(fields)
    M = np.zeros((Q, S * nf), dtype=np.uint8)
    for f, fv in enumerate(fields):
        M[:, f::nf] = (fv.reshape(Q, S) < thr)
    p = M.mean()
    # correlations: adjacent keys, adjacent queries, diagonal, key+2
    def corr(a, b):
        a = a.astype(np.float64) - p; b = b.astype(np.float64) - p
        return (a * b).mean() / (p * (1 - p))
    c1 = corr(M[:, :-1], M[:, 1:]); c2 = corr(M[:-1], M[1:]); c3 = corr(M[:-1, :-1], M[1:, 1:]); c4 = corr(M[:, :-2], M[:, 2:]); c5=corr(M[:, :-4], M[:, 4:]); c6=corr(M[:-2], M[2:])
    # row / column sums variance vs binomial
    n = M.shape[1]
    rv = M.sum(1).var() / (n * p * (1 - p)); cv = M.sum(0).var() / (M.shape[0] * p * (1 - p))
    # 8-bit pattern chi-square along keys
    pat = np.packbits(M[:, : (n // 8) * 8].reshape(Q, -1, 8), axis=2).ravel()
    cnt = np.bincount(pat, minlength=256).astype(np.float64)
    pop = np.array([bin(i).count("1") for i in range(256)])
    exp = len(pat) * (p ** pop) * ((1 - p) ** (8 - pop))
    chi = ((cnt - exp) ** 2 / exp).sum() / 255
    print(f"{name:10s} p={p:.5f} corr key+1 {c1:+.4f} q+1 {c2:+.4f} diag {c3:+.4f} key+2 {c4:+.4f} key+4 {c5:+.4f} q+2 {c6:+.4f} rowvar {rv:.3f} colvar {cv:.3f} chi8 {chi:.2f}")
Q, S = 512, 256      # 512 queries x 256 key pairs (Sk = 512)
q = np.arange(Q, dtype=np.uint64)[:, None]; j = np.arange(S, dtype=np.uint64)[None, :]
thr = int(0.2 * 65536 + 0.5)
for key in (0x12345678, 0xdeadbeef, 0x0, 0x9abcdef1):
    x = (q * S + j) & 0xffffffff
    h = lowbias32(x ^ np.uint64(key))
    stats([h & 0xffff, h >> 16], (Q, S), thr, "lowbias32")
    x4 = (q * (S // 2) + j[:, : S // 2]) & 0xffffffff
    stats(candB(x4, np.uint64(key)), (Q, S // 2), thr, "mul64x4")
    stats(candC(x, np.uint64(key)), (Q, S), thr, "1mul x2")
print("---- 24-bit multiply variants")
def mul24(a, c):
    return ((a & np.uint64(0xffffff)) * np.uint64(c)) & np.uint64(0xffffffff)
def cand24(x, key, c1=0xb5352d, c2=0xca68b5, s1=16, s2=13, s3=16):
    x = (x ^ key).astype(np.uint64)
    x ^= x >> np.uint64(s1); x = mul24(x, c1); x ^= x >> np.uint64(s2); x = mul24(x, c2); x ^= x >> np.uint64(s3)
    return [x & 0xffff, (x >> 16) & 0xffff]
def cand24b(x, key):
    # 3 rounds of u24 multiply (still all full rate)
    x = (x ^ key).astype(np.uint64)
    x ^= x >> np.uint64(16); x = mul24(x, 0xb5352d); x ^= x >> np.uint64(12); x = mul24(x, 0xca68b5); x ^= x >> np.uint64(15); x = mul24(x, 0x2c1b3d); x ^= x >> np.uint64(16)
    return [x & 0xffff, (x >> 16) & 0xffff]
for key in (0x12345678, 0xdeadbeef, 0x0):
    x = (q * S + j) & 0xffffffff
    stats(cand24(x, np.uint64(key)), (Q, S), thr, "u24 x2")
    stats(cand24b(x, np.uint64(key)), (Q, S), thr, "u24 x3")
# larger index ranges (b,h offset) and different S
x = ((q + 100000) * 2048 + j * 7) & 0xffffffff
stats(cand24(x, np.uint64(0x5555)), (Q, S), thr, "u24x2 big")
stats(cand24b(x, np.uint64(0x5555)), (Q, S), thr, "u24x3 big")
h = lowbias32(x ^ np.uint64(0x5555)); stats([h & 0xffff, h >> 16], (Q, S), thr, "lowb big")
