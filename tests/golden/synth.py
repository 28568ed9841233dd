"""Seeded synthetic live positions (numpy only): random walls placed under the static
overlap rule, pawns anywhere a live game allows (P1 not on row 8, P2 not on row 0), random
wall counts.  Used by the parity tests at sizes the oracle finishes in seconds and by
bench.py's position sets (S-open / S-mid / S-dense)."""
import numpy as np

from _stubs import PACKED_DTYPE


def synth_positions(n, seed=0, min_walls=0, max_walls=20, mover_has_walls=None, adjacent_frac=0.3):
    rng = np.random.RandomState(seed)
    out = np.zeros(n, dtype=PACKED_DTYPE)
    for i in range(n):
        k = rng.randint(min_walls, max_walls + 1)
        hb = vb = 0
        placed = 0
        tries = 0
        while placed < k and tries < 400:
            tries += 1
            ix = int(rng.randint(64))
            horiz = bool(rng.randint(2))
            if ((hb | vb) >> ix) & 1:
                continue
            if horiz:
                if ix % 8 != 0 and (hb >> (ix - 1)) & 1:
                    continue
                if ix % 8 != 7 and (hb >> (ix + 1)) & 1:
                    continue
                hb |= 1 << ix
            else:
                if ix // 8 != 0 and (vb >> (ix - 8)) & 1:
                    continue
                if ix // 8 != 7 and (vb >> (ix + 8)) & 1:
                    continue
                vb |= 1 << ix
            placed += 1
        while True:
            p1 = int(rng.randint(0, 72))
            if rng.rand() < adjacent_frac:
                d = [9, -9, 1, -1][rng.randint(4)]
                p2 = p1 + d
                if not (9 <= p2 <= 80) or (abs(d) == 1 and p1 // 9 != p2 // 9):
                    continue
            else:
                p2 = int(rng.randint(9, 81))
            if p1 != p2:
                break
        cur = int(rng.randint(1, 3))
        w1 = int(rng.randint(0, 11))
        w2 = int(rng.randint(0, 11))
        if mover_has_walls is True:
            if cur == 1:
                w1 = max(w1, 1)
            else:
                w2 = max(w2, 1)
        elif mover_has_walls is None and rng.rand() < 0.7:
            if cur == 1:
                w1 = max(w1, 1)
            else:
                w2 = max(w2, 1)
        out[i]["hbits"], out[i]["vbits"] = hb, vb
        out[i]["p1"], out[i]["p2"], out[i]["w1"], out[i]["w2"], out[i]["cur"] = p1, p2, w1, w2, cur
    return out
