"""Bank-conflict check of the LDS exchange layouts used by the negacyclic FFT (N=1024, 64 lanes x 8 points).

Lane groups and bank moduli follow /opt/skills/guides/MI355X_MICROARCH.md (LDS section):
  ds_read_b128 : 4 groups of 16 lanes {0-3,12-15,20-27},{4-11,16-19,28-31} (+32), bank = (addr/4) % 64
  ds_write_b128: 8 groups of 8 contiguous lanes,                             bank = (addr/4) % 32
An element is 16 bytes (double2); `slot` = element index.  A group is conflict-free when its lanes touch
disjoint banks.  Prints the worst-case way-ness per exchange.
"""
RG = [[0,1,2,3,12,13,14,15,20,21,22,23,24,25,26,27],[4,5,6,7,8,9,10,11,16,17,18,19,28,29,30,31]]
RG = RG + [[x+32 for x in g] for g in RG]
WG = [list(range(8*g, 8*g+8)) for g in range(8)]

def ways(slots, nbanks):
    # each 16-B slot covers 4 consecutive banks
    cnt = {}
    for s in set(slots):
        for b in range(4):
            bank = (4*s + b) % nbanks
            cnt[bank] = cnt.get(bank, 0) + 1
    return max(cnt.values())

def layA(t, m): return (m << 6) | t
def layB(t, m): return ((t >> 3) << 6) | (m << 3) | (t & 7)
def layC(t, m): return (t << 3) | m

def sig1(j): return j ^ (((j >> 6) & 7) << 3)
def sig2(j): return j ^ ((j >> 4) & 7)
def tau(j):  return (((j >> 8) & 1) << 8) | (((j >> 6) & 1) << 7) | ((j & 7) << 4) | (((j >> 7) & 1) << 3) | (((j >> 3) & 7) ^ (j & 7))
def ident(j): return j

def check(name, wl, rl, f):
    assert sorted(f(j) for j in range(512)) == list(range(512)), name + " not a bijection"
    w = max(ways([f(wl(t, m)) for t in g], 32) for m in range(8) for g in WG)
    r = max(ways([f(rl(t, m)) for t in g], 64) for m in range(8) for g in RG)
    print("%-28s write %d-way  read %d-way" % (name, w, r))

check("fwd A->B sigma1", layA, layB, sig1)
check("fwd B->C sigma2", layB, layC, sig2)
check("inv C->B tau", layC, layB, tau)
check("inv B->A identity", layB, layA, ident)
check("fwd A->B identity", layA, layB, ident)
check("fwd B->C identity", layB, layC, ident)
check("inv C->B sigma2", layC, layB, sig2)

# ---- padded (additive) layouts: address = lane base + m * constant, so every DS access uses an immediate offset
def f1(j): return j + 8 * (j >> 6)        # A <-> B   (576 slots)
def f2(j): return j + (j >> 3)            # B <-> C   (576 slots)

def check2(name, wl, rl, f, size):
    img = [f(j) for j in range(512)]
    assert len(set(img)) == 512 and max(img) < size, name
    w = max(ways([f(wl(t, m)) for t in g], 32) for m in range(8) for g in WG)
    r = max(ways([f(rl(t, m)) for t in g], 64) for m in range(8) for g in RG)
    # additivity in the register index on both sides
    for lay in (wl, rl):
        for t in range(64):
            d = [f(lay(t, m)) - f(lay(t, 0)) for m in range(8)]
            assert d == [m * d[1] for m in range(8)], (name, "not additive", d)
    print("%-28s write %d-way  read %d-way  (additive, %d slots)" % (name, w, r, size))

print()
check2("fwd A->B padded f1", layA, layB, f1, 576)
check2("inv B->A padded f1", layB, layA, f1, 576)
check2("fwd B->C padded f2", layB, layC, f2, 576)
check2("inv C->B padded f2", layC, layB, f2, 576)
