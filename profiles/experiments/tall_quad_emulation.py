"""Wave-level emulation of the 'four tall Gaussians per wave' expansion vs. the straightforward per-Gaussian loop."""
import random
random.seed(1)
ROWS_BIG = 6
def widths_for(g):   # deterministic pseudo row widths for Gaussian g: h rows
    rnd = random.Random(g["seed"]); return [rnd.randint(0, 9) for _ in range(g["h"])]
def reference(wave):
    out = {}
    for lane, g in enumerate(wave):
        if g["h"] > ROWS_BIG:
            pos = g["cur"]; ent = []
            for row, w in enumerate(widths_for(g)):
                for t in range(w): ent.append((pos, g["gid"], row, t)); pos += 1
            out[lane] = (ent, pos - g["cur"])
    return out
def shfl(vals, idx): return [vals[i] for i in idx]
def quad(wave):
    h = [g["h"] for g in wave]; cur = [g["cur"] for g in wave]; gid = [g["gid"] for g in wave]
    big = [l for l in range(64) if h[l] > ROWS_BIG]
    cnt = [0] * 64; entries = {}
    while big:
        srcs = [(big.pop(0) if big else -1) for _ in range(4)]
        src = [srcs[l >> 4] for l in range(64)]; sl = [max(s, 0) for s in src]
        bh_any = shfl(h, sl); bh = [bh_any[l] if src[l] >= 0 else 0 for l in range(64)]
        bgid = shfl(gid, sl); base = shfl(cur, sl)
        max_bh = max([h[s] for s in srcs if s >= 0] + [0])
        total = [0] * 64
        r0 = 0
        while r0 < max_bh:
            c = []
            for l in range(64):
                j = l & 15
                if r0 + j < bh[l]:
                    c.append(widths_for(wave[sl[l]])[r0 + j])
                else: c.append(0)
            incl = c[:]
            for sh in (1, 2, 4, 8):   # row_shr within 16-lane rows, zero fill
                incl = [incl[l] + (incl[l - sh] if (l & 15) >= sh else 0) for l in range(64)]
            for l in range(64):
                pos = base[l] + incl[l] - c[l]
                for t in range(c[l]):
                    entries.setdefault(sl[l], []).append((pos, bgid[l], r0 + (l & 15), t)); pos += 1
            chunk = [incl[(l & 48) | 15] for l in range(64)]
            base = [base[l] + chunk[l] for l in range(64)]; total = [total[l] + chunk[l] for l in range(64)]
            r0 += 16
        for g in range(4):
            t = total[g * 16]
            if srcs[g] >= 0: cnt[srcs[g]] = t
    return entries, cnt
bad = 0
for trial in range(300):
    wave = []; cur = 0
    for l in range(64):
        hh = random.choice([0, 1, 3, 6, 7, 9, 16, 17, 33, 40]) if random.random() < 0.5 else random.randint(0, 5)
        g = dict(h=hh, gid=1000 + l, seed=random.randint(0, 10**9), cur=cur)
        cur += sum(widths_for(g)) if hh > ROWS_BIG else random.randint(0, 20)
        wave.append(g)
    ref = reference(wave); ent, cnt = quad(wave)
    for lane, (e, n) in ref.items():
        if sorted(ent.get(lane, [])) != sorted(e) or cnt[lane] != n: bad += 1
print("mismatches:", bad)
