#!/usr/bin/env python3
"""Round model of the pipelined batch-1 step (profiles/r2_stamps_timeline.md, section 7): the launch chains of the lanes advance in
lock-step -- one kernel of every lane per round, round = longest kernel of the round + launch gap.  Given the measured durations of
the eight launches of a layer (beside two other chains) it reproduces the measured step (0.486 ms modelled, 0.49 measured at three
lanes) and prices restructurings BEFORE they are built: e.g. a persistent FFN pair (W1 -> in-kernel hand-off -> W2 in one launch)
removes two rounds of eight but puts a 10 us kernel into the rounds it takes part in.
usage: python3 tests/micro/round_model.py"""
import itertools
import statistics

GAP = 2.2                                                    # us, end of the round's last kernel -> start of the next round
LAYER = [4.9, 2.8, 4.4, 6.1, 4.1, 5.9, 4.5, 2.8]            # LN+W1, W2, LN+QKV, attn+Wo, LN+pw1, dw+pw2, LN+W1', W2' (us, three lanes)


def step_ms(d, lanes, n_layers=24):
    """mean / best / worst over the relative phases of the lanes"""
    n, tot = len(d), []
    for offs in itertools.product(range(n), repeat=lanes - 1):
        tot.append(sum(max([d[r]] + [d[(r + o) % n] for o in offs]) + GAP for r in range(n)))
    f = n_layers / lanes / 1000.0
    return statistics.mean(tot) * f, min(tot) * f, max(tot) * f


if __name__ == "__main__":
    cases = [("8 launches per layer (shipped)", LAYER),
             ("FFN pairs as one persistent launch, hand-off 3 us (guide: 2.4-3.0 us all-gather of 8 KB)", [4.9 + 3.0 + 2.8, 4.4, 6.1, 4.1, 5.9, 4.5 + 3.0 + 2.8]),
             ("the same with a 2 us hand-off", [4.9 + 2.0 + 2.8, 4.4, 6.1, 4.1, 5.9, 4.5 + 2.0 + 2.8]),
             ("the same with the W2 weights prefetched under the hand-off (pair = 8.5 us)", [8.5, 4.4, 6.1, 4.1, 5.9, 8.3]),
             ("attention and conv blocks fused as well: 4 launches per layer", [10.7, 4.4 + 6.1 + 2.5, 4.1 + 5.9 + 2.5, 10.3])]
    for name, d in cases:
        m, lo, hi = step_ms(d, 3)
        print(f"{name:95s} three lanes: {m:.3f} ms per step (phases aligned {lo:.3f}, worst {hi:.3f})")
