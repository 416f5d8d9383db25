"""Which GF(2)-linear shuffles of the low five index bits make the kd-tree network's LDS stage pairs free of bank conflicts
(align3d_amd/csrc/kdtree_select.hip, bitonic_sort, `phys`).

A stage pair of distances (j, j / 2), hh = j / 2 = 2^lh, has quad q touch the 64-bit words base(q) | {0, hh, j, j + hh} with
base(q) = (q >> lh) << (lh + 2) | (q & (hh - 1)): q's bits below lh stay, the others move up by two.  One ds_read_b64 is served in
two groups of 32 lanes over 32 8-byte slots, one ds_write_b64 in four groups of 16 lanes over 16 slots (MI355X_MICROARCH.md, LDS):
the map q -> slot has to be a bijection on q's low five (four) bits for every lh.  slot = xor of cols[p] over the set address bits
p = 0 .. 6; bits 7 and up are the same for a group's lanes.  Prints the maps with identity on bits 0-2 that pass, sparsest first:
the first one, cols = [1, 2, 4, 8, 21, 10, 16], is  i ^ 5 * (bits 4-5 of i) ^ (bit 6 of i) << 4."""


def independent(vecs):
    basis = []
    for v in vecs:
        for b in basis:
            v = min(v, v ^ b)
        if v == 0:
            return False
        basis.append(v)
    return True


def conflicts(cols):
    bad = 0
    for lh in range(6):
        pos = lambda nq: [k if k < lh else k + 2 for k in range(nq)]
        bad += not independent([cols[p] for p in pos(5) if p <= 6])            # 32 lanes of a read group
        bad += not independent([cols[p] & 15 for p in pos(4) if p <= 6])       # 16 lanes of a write group
    return bad


if __name__ == "__main__":
    print("plain layout:", conflicts([1, 2, 4, 8, 16, 0, 0]), "of 12 (stage, access) cases conflict")
    good = [[1, 2, 4, c3, c4, m5, m6] for c3 in range(8, 16) for c4 in range(16, 32) for m5 in range(32) for m6 in range(32)
            if independent([1, 2, 4, c3, c4]) and conflicts([1, 2, 4, c3, c4, m5, m6]) == 0]
    good.sort(key=lambda c: sum(bin(x).count("1") for x in c))
    print(len(good), "conflict-free maps; sparsest:", good[:5])
