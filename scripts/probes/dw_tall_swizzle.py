# brute-force: conflict-free slot swizzle for B-fragment ds_read_b128 reads of the tall dw kernel
# LDS image: pixel slot P = r*17 + cc (cc 0..15 real, 16 zero pixel), 128 B per pixel = 8 slots of 16 B; slot' = w ^ f(r, cc)
# bank16 (16-byte bank group of 16) = (P*8 + slot') % 16
import itertools
groups = [[0,1,2,3,12,13,14,15,20,21,22,23,24,25,26,27],[4,5,6,7,8,9,10,11,16,17,18,19,28,29,30,31]]
groups += [[l+32 for l in g] for g in groups]
def conflicts(f, pitch=17):
    worst = 0; tot = 0
    for w in range(8):
        for kw in range(7):
            for rb in range(0, 11):   # row base (any, to be safe)
                for g in groups:
                    seen = {}
                    for l in g:
                        n, q = l & 15, l >> 4
                        r = rb + q
                        c = n + kw - 3
                        cc = c if 0 <= c < 16 else 16
                        P = r * pitch + cc
                        addr = P * 8 + (w ^ f(r, cc))
                        seen.setdefault(addr % 16, set()).add(addr)
                    m = max(len(v) for v in seen.values())
                    worst = max(worst, m); tot += sum(len(v) - 1 for v in seen.values())
    return worst, tot
cands = {
 'cc>>1': lambda r, cc: (cc >> 1) & 7,
 '(cc>>1)^(r&1)': lambda r, cc: ((cc >> 1) ^ (r & 1)) & 7,
 '((cc+(r&1))>>1)': lambda r, cc: ((cc + (r & 1)) >> 1) & 7,
 '((cc+r)>>1)': lambda r, cc: ((cc + r) >> 1) & 7,
 '(P>>1)': lambda r, cc: ((r * 17 + cc) >> 1) & 7,
 '0': lambda r, cc: 0,
}
for k, f in cands.items():
    print(k, conflicts(f))
for pitch in (16, 18):
    for k, f in cands.items():
        print('pitch', pitch, k, conflicts(f, pitch))
def conflicts2():
    worst = 0; tot = 0
    for w in range(8):
        for kw in range(7):
            for rb in range(0, 11):
                for g in groups:
                    seen = {}
                    for l in g:
                        n, q = l & 15, l >> 4
                        r = rb + q
                        c = n + kw - 3
                        cc = c if 0 <= c < 16 else 16 + (c & 1)
                        P = r * 18 + cc
                        addr = P * 8 + (w ^ ((c >> 1) & 7))
                        seen.setdefault(addr % 16, set()).add(addr)
                    m = max(len(v) for v in seen.values())
                    worst = max(worst, m); tot += sum(len(v) - 1 for v in seen.values())
    return worst, tot
print("pitch 18, two zero pixels:", conflicts2())

def conflicts_pair():
    """two 8-wide images side by side in one 16-column tile (stage 3, 8 x 8 maps): lane n -> image n >> 3, column (n & 7) + kw - 3; slot = image * 8 + column"""
    worst = 0; tot = 0
    for w in range(8):
        for kw in range(7):
            for rb in range(0, 7):
                for g in groups:
                    seen = {}
                    for l in g:
                        n, q = l & 15, l >> 4
                        r = rb + q
                        c = (n & 7) + kw - 3
                        cs = (n >> 3) * 8 + c
                        cc = cs if 0 <= c < 8 else 16 + (cs & 1)
                        P = r * 18 + cc
                        addr = P * 8 + (w ^ ((cs >> 1) & 7))
                        seen.setdefault(addr % 16, set()).add(addr)
                    m = max(len(v) for v in seen.values())
                    worst = max(worst, m); tot += sum(len(v) - 1 for v in seen.values())
    return worst, tot
print("pair tiles (two 8-wide images), pitch 18:", conflicts_pair())
