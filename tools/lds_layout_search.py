"""Search padded, additive LDS layouts for the N=2048 transform (M=1024 points, 128 threads x 8 registers).

slot(j) = j + sum_k pad_k * (j >> shift_k): linear in the index bits, hence `thread base + m * const` on both sides.
A layout (reg bits, thread bits) maps thread t (two waves of 64 lanes) and register m to index j.
"""
import itertools, sys
RG = [[0,1,2,3,12,13,14,15,20,21,22,23,24,25,26,27],[4,5,6,7,8,9,10,11,16,17,18,19,28,29,30,31]]
RG = RG + [[x+32 for x in g] for g in RG]
WG = [list(range(8*g, 8*g+8)) for g in range(8)]

def ways(slots, nbanks):
    cnt = {}
    for s in set(slots):
        for b in range(4):
            bank = (4*s + b) % nbanks
            cnt[bank] = cnt.get(bank, 0) + 1
    return max(cnt.values())

def make_layout(reg_bits, thread_bits):
    # reg_bits: list of index-bit positions for register bits (LSB first); thread_bits likewise for thread id bits
    def lay(t, m):
        j = 0
        for i, b in enumerate(reg_bits): j |= ((m >> i) & 1) << b
        for i, b in enumerate(thread_bits): j |= ((t >> i) & 1) << b
        return j
    return lay

def evaluate(f, wl, rl):
    w = max(ways([f(wl(64*wv + l, m)) for l in g], 32) for wv in range(2) for m in range(8) for g in WG)
    r = max(ways([f(rl(64*wv + l, m)) for l in g], 64) for wv in range(2) for m in range(8) for g in RG)
    return w, r

A = make_layout([7,8,9], [0,1,2,3,4,5,6])
B = make_layout([4,5,6], [0,1,2,3,7,8,9])
C1 = make_layout([1,2,3], [0,4,5,6,7,8,9])          # thread = 2u + e
C2 = make_layout([1,2,3], [4,5,6,7,8,9,0])          # thread = 64 e + u  (e = wave)
D = make_layout([0,1,2], [3,4,5,6,7,8,9])

def search(name, X, Y, shifts, maxpad=9):
    best = None
    for pads in itertools.product(range(maxpad), repeat=len(shifts)):
        f = lambda j: j + sum(p * (j >> s) for p, s in zip(pads, shifts))
        img = [f(j) for j in range(1024)]
        if len(set(img)) != 1024: continue
        size = max(img) + 1
        ok = True
        for (wl, rl) in ((X, Y), (Y, X)):
            w, r = evaluate(f, wl, rl)
            if w > 1 or r > 1: ok = False; break
        if ok and (best is None or size < best[0]):
            best = (size, pads)
    print(name, "shifts", shifts, "->", best)

search("A<->B ", A, B, [7])
search("B<->C1", B, C1, [4, 7])
search("B<->C2", B, C2, [4, 7])
search("C1<->D", C1, D, [3, 4, 7], maxpad=5)
search("C2<->D", C2, D, [3, 4, 7], maxpad=5)
