#!/usr/bin/env python3
"""Exhaustive search for bank-conflict-free LDS halo layouts of conv_data_kernel (csrc/conv_mfma.hip).

ds_read_b128 is served in fixed 16-lane groups; a group is conflict-free when its 16 fragments (16 B each) fall into 16
distinct slots of the 256-byte bank row.  Free parameters: the bit permutation mapping an MFMA row (lane & 31) to the
(h, w) position inside its 32-position sub-tile, and the halo row pitch RS in 16-byte slots (stride-2 halos are split into
even-x / odd-x planes).  Prints, per kernel flavour, the conflict degree of the naive layout and the first conflict-free
(RS, permutation).  The chosen values are hard-coded in SubTile<> / HaloPitch<>."""
import itertools

GROUPS = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]


def worst(RS, mapfn, stride):
    wv = 0
    for kh in range(4 if stride == 2 else 3):
        for kx in range(2 if stride == 2 else 3):
            for grp in GROUPS:
                cnt = {}
                for r in grp:
                    hh, w = mapfn(r)
                    y, xi = (2 * hh + kh, w + kx) if stride == 2 else (hh + kh, w + kx)
                    s = (y * RS + xi) % 16
                    cnt[s] = cnt.get(s, 0) + 1
                wv = max(wv, max(cnt.values()))
    return wv


def search(name, sw_bits, sh_bits, stride, min_rs):
    naive = worst(min_rs, lambda r: (r >> sw_bits, r & ((1 << sw_bits) - 1)), stride)
    for RS in range(min_rs, min_rs + 8):
        for perm in itertools.permutations(range(5)):
            def mapfn(r, perm=perm):
                b = [(r >> perm[i]) & 1 for i in range(5)]
                return sum(b[sw_bits + i] << i for i in range(sh_bits)), sum(b[i] << i for i in range(sw_bits))
            if worst(RS, mapfn, stride) == 1:
                print(f"{name}: naive layout {naive}-way; conflict-free at RS={RS}, w bits <- lane bits {perm[:sw_bits]}, h bits <- {perm[sw_bits:]}")
                return
    print(name, "no conflict-free layout found")


if __name__ == "__main__":
    search("down 3D (sub-tile 4x8, stride 2)", 3, 2, 2, 9)
    search("up   3D (sub-tile 4x8, stride 1)", 3, 2, 1, 10)
    search("down 2D (sub-tile 2x16, stride 2)", 4, 1, 2, 17)
    search("up   2D (sub-tile 2x16, stride 1)", 4, 1, 1, 18)
