# bank-conflict model of k_stem2x's intermediate-tile writes (phase A) and B-fragment reads (phase B)
G128 = [[0,1,2,3,12,13,14,15,20,21,22,23,24,25,26,27],[4,5,6,7,8,9,10,11,16,17,18,19,28,29,30,31]]
G128 = G128 + [[l+32 for l in g] for g in G128]
W128 = [list(range(8*i, 8*i+8)) for i in range(8)]
def cycles(addrs, nbytes, groups, nbanks):
    tot = 0
    for g in groups:
        bank = {}
        for l in g:
            a = addrs[l]
            if a is None: continue
            for d in range(nbytes // 4):
                bank.setdefault(((a // 4) + d) % nbanks, set()).add((a // 4) + d)
        tot += max([len(v) for v in bank.values()] + [1])
    return tot
IWh, IWs = 33, 66
def slot(my, mx):
    rem = (mx & 1) * IWh + (mx >> 1)
    return (my * IWs + rem) * 128, (rem >> 1) & 7
def write_cycles(mapping, key):
    tot = 0; n = 0
    for grp in range(18):
        my, half = grp >> 1, grp & 1
        for o in range(4):
            addrs = []
            for lane in range(64):
                pix, hh = lane & 31, lane >> 5
                mx = (half << 5) + mapping(pix)
                dst, fk = slot(my, mx)
                fk = key(my, mx)
                addrs.append(dst + (((2 * o + hh) ^ fk) * 16))
            tot += cycles(addrs, 16, W128, 32); n += 1
    return tot / n
def read_cycles(key):
    tot = 0; n = 0
    for wave in range(4):
        for r in range(3):
            for s in range(3):
                for q in range(4):
                    addrs = []
                    for lane in range(64):
                        pix, hh = lane & 31, lane >> 5
                        mx = 2 * pix + s
                        my = 2 * wave + r
                        dst, _ = slot(my, mx)
                        addrs.append(dst + (((2 * q + hh) ^ key(my, mx)) * 16))
                    tot += cycles(addrs, 16, G128, 64); n += 1
    return tot / n
ident = lambda p: p
perm = [0, 2, 1, 3]
newmap = lambda p: 4 * (p & 7) + perm[p >> 3]
key0 = lambda my, mx: (((mx & 1) * IWh + (mx >> 1)) >> 1) & 7
print('writes (ideal 8): current mapping', write_cycles(ident, key0), ' permuted pixels', write_cycles(newmap, key0))
print('reads  (ideal 4):', read_cycles(key0))
# alternative keys that keep reads conflict-free?
for name, key in [('rem>>1 ^ 4*(mx&1)', lambda my, mx: ((((mx & 1) * IWh + (mx >> 1)) >> 1) ^ (4 * (mx & 1))) & 7),
                  ('rem>>1 ^ (my&1)*2', lambda my, mx: ((((mx & 1) * IWh + (mx >> 1)) >> 1) ^ ((my & 1) * 2)) & 7)]:
    print(name, 'writes', write_cycles(ident, key), 'reads', read_cycles(key))
