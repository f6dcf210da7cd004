import itertools
groups = [[0,1,2,3,12,13,14,15,20,21,22,23,24,25,26,27],[4,5,6,7,8,9,10,11,16,17,18,19,28,29,30,31]]
groups += [[l+32 for l in g] for g in groups]
def cycles(W, pitch16, f, chunks_per_pos=4, ksteps=1):
    tot = 0; worst = 0
    for w_h in (0,1):
      for ky in range(5):
        for kx in range(5):
          for ks in range(ksteps):
            for grp in groups:
                banks = {}
                for l in grp:
                    m, g = l & 15, l >> 4
                    wl, q = m >> 2, m & 3
                    pos = (q >> 1) * W + 2 * (4 * w_h + wl) + (q & 1) + ky * W + kx
                    chunk = ks * 4 + g
                    addr16 = pos * pitch16 + (chunk ^ f(pos, W))
                    # 16B chunk covers 4 banks: bank group index = addr16 mod 16
                    banks.setdefault(addr16 % 16, set()).add(addr16)
                c = max(len(v) for v in banks.values())
                tot += c; worst = max(worst, c)
    return tot / (2*25*ksteps*4), worst
fs = {"none": lambda p, W: 0}
for s in range(0, 5):
    fs[f"(p>>{s})&3"] = (lambda s: lambda p, W: (p >> s) & 3)(s)
    fs[f"(y>>{s})&3"] = (lambda s: lambda p, W: ((p // W) >> s) & 3)(s)
fs["(y^x>>1)&3"] = lambda p, W: ((p // W) ^ ((p % W) >> 1)) & 3
fs["(y+x>>1)&3"] = lambda p, W: ((p // W) + ((p % W) >> 1)) & 3
fs["(2y+x>>1)&3"] = lambda p, W: (2*(p // W) + ((p % W) >> 1)) & 3
fs["(y*2^x>>1)&3"] = lambda p, W: ((2*(p // W)) ^ ((p % W) >> 1)) & 3
res = []
for W in (18, 19, 20, 21, 22):
    for pitch16 in (4, 5, 6):
        for name, f in fs.items():
            if pitch16 != 4 and name != "none": continue
            avg, worst = cycles(W, pitch16, f)
            res.append((avg, worst, W, pitch16, name))
res.sort()
for r in res[:15]: print(r)
print([r for r in res if r[2]==18 and r[3]==4 and r[4]=="none"])
