"""Seeded (reference window, read) pairs for Needleman-Wunsch: reads cut from the window with substitutions, indels, repeats."""
import numpy as np


def nw_pairs(seed, n=40, ref_len=(120, 320), read_len=(30, 150), repeats=True, ns=True):
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n):
        L1 = int(rng.integers(ref_len[0], ref_len[1] + 1))
        ref = list(rng.choice(list("ACGT"), L1))
        if repeats and rng.random() < 0.7:                       # an STR inside the window: many equally good gap placements
            p = int(rng.integers(1, 7)); motif = list(rng.choice(list("ACGT"), p)); c = int(rng.integers(4, 16))
            at = int(rng.integers(10, max(11, L1 - p * c - 10)))
            ref[at:at + p * c] = motif * c
            ref = ref[:L1]
        L2 = int(rng.integers(read_len[0], min(read_len[1], L1 - 10) + 1))
        st = int(rng.integers(0, L1 - L2 + 1))
        read = ref[st:st + L2]
        i = 0
        while i < len(read):
            u = rng.random()
            if u < 0.01:
                read[i] = str(rng.choice(list("ACGT")))
            elif u < 0.02:
                del read[i:i + int(rng.integers(1, 9))]
            elif u < 0.03:
                read[i:i] = list(rng.choice(list("ACGT"), int(rng.integers(1, 9))))
            elif ns and u < 0.033:
                read[i] = "N"
            i += 1
        read = read[:max(256, read_len[1])]
        if len(read) == 0:
            read = ["A"]
        if ns and rng.random() < 0.1:
            ref[int(rng.integers(L1))] = "N"
        out.append(("".join(ref), "".join(read)))
    return out
