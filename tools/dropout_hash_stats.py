"""statistics of the counter-based dropout hash of ralf_amd/csrc/common.h (attn_rng2x16) in numpy: keep rate at p = 0.1, correlation of the keep
bits along the keys of a row (lags 1 .. 64), between rows, and within the pair of 16-bit fields of one hash -- for the three-round form (rounds 1-4)
and the two-round form (round 5), with random row keys (attention) and with one constant key over sequential groups (element dropout of the
GEMM epilogues: all variation comes from the group index).      python tools/dropout_hash_stats.py"""
import numpy as np

M32 = np.uint64(0xffffffff)


def mul24(a, b):
    return (a.astype(np.uint64) & np.uint64(0xffffff)) * np.uint64(b & 0xffffff)


def lo32(p):
    return (p & M32).astype(np.uint32)


def three_rounds(rk, p):
    x = rk ^ lo32(mul24(p, 0x9E3779)); x ^= x >> np.uint32(16); x = lo32(mul24(x, 0xEB352D)); x ^= x >> np.uint32(15)
    x = lo32(mul24(x, 0xA68B6B)); x ^= x >> np.uint32(15)
    return x


def two_rounds(rk, p):
    x = rk ^ lo32(mul24(p, 0x9E3779)); x ^= x >> np.uint32(16); x = lo32(mul24(x, 0xEB352D)); x ^= x >> np.uint32(15)
    return x


def corr(a, b):
    a = a - a.mean(); b = b - b.mean()
    return float((a * b).mean() / np.sqrt((a * a).mean() * (b * b).mean() + 1e-30))


rng = np.random.default_rng(1)
R, P, thr = 2048, 1024, int(0.1 * 65536)
for rkname, rk in (("random row keys", rng.integers(0, 2 ** 32, R, dtype=np.uint64).astype(np.uint32)), ("one constant key", np.full(R, 0x1234567, dtype=np.uint32))):
    pr = np.arange(P, dtype=np.uint32)[None, :] + (np.arange(R, dtype=np.uint32)[:, None] * np.uint32(P) if "constant" in rkname else np.uint32(0))
    for hn, hf in (("three rounds", three_rounds), ("two rounds", two_rounds)):
        h = hf(rk[:, None], pr)
        f0, f1 = h & np.uint32(0xffff), h >> np.uint32(16)
        k0, k1 = (f0 >= thr).astype(np.float64), (f1 >= thr).astype(np.float64)
        keep = np.stack([k0, k1], -1).reshape(R, -1)
        lags = [corr(keep[:, :-l], keep[:, l:]) for l in (1, 2, 3, 4, 5, 8, 16, 64)]
        rows = [corr(keep[:-l], keep[l:]) for l in (1, 2)]
        hist = np.bincount((h.ravel() >> np.uint32(24)).astype(np.int64), minlength=256); e = h.size / 256
        print(f"{rkname:18s} {hn:13s} keep {keep.mean():.4f}  pair {corr(k0, k1):+.4f}  key lags {' '.join(f'{c:+.4f}' for c in lags)}  row lags {' '.join(f'{c:+.4f}' for c in rows)}"
              f"  chi2/dof(top byte) {((hist - e) ** 2 / e).sum() / 255:.2f}")

# masks of two STREAMS (different keys, the same sequential indices) at p = 0.5 and 0.1: what decided against the two-round form
P2 = 1 << 22
idx = np.arange(P2, dtype=np.uint32)
for hn, hf in (("three rounds", three_rounds), ("two rounds", two_rounds)):
    worst = {0.5: 0.0, 0.1: 0.0}
    for _ in range(6):
        k0, k1 = rng.integers(0, 2 ** 32, 2, dtype=np.uint64).astype(np.uint32)
        a, b = hf(np.uint32(k0), idx), hf(np.uint32(k1), idx)
        for p in worst:
            t = int(p * 65536)
            for fa, fb in (((a & np.uint32(0xffff)), (b & np.uint32(0xffff))), (a >> np.uint32(16), b >> np.uint32(16))):
                worst[p] = max(worst[p], abs(corr((fa >= t).astype(np.float64), (fb >= t).astype(np.float64))))
    print(f"cross-stream |corr|, worst of 6 key pairs x 2 fields: {hn:13s} p = 0.5: {worst[0.5]:.4f}   p = 0.1: {worst[0.1]:.4f}")
