"""What could feed-forward work placed on XCDs the recurrence has left buy on the bench workload?  (round-3 verdict item 1)

Host-only model of the BASELINE configs[1] pass with the library's own plan (182 clips, LPT into 128 slots, 49 152-row chunks):
per chunk the recurrence holds ceil(live slots / 16) XCDs (slots re-dealt contiguously at every launch) for its sequential
steps (measured 1.67 us + 0.0102 us per live column), and the feed-forward of the NEXT chunk (58.5 ms per pass on the whole chip:
projections 52.5 + LayerNorm 4 + head 2) may use the other XCDs for that long at `eff` of its whole-chip rate; what does not
fit runs afterwards on the whole chip.  Everything else about such a scheme (dynamic tile queue, ready flags, a co-residable or
single-launch feed-forward kernel) is assumed free.

    python scripts/overlap_sim.py
"""
import heapq
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from prego_amd.workloads import assembly101_eval_lengths

S, CAP, FF_MS = 128, 49152, 58.5


def main():
    lens = assembly101_eval_lengths(seed=20)
    heap = [(0, i) for i in range(S)]
    heapq.heapify(heap)
    load = [0] * S
    for i in sorted(range(len(lens)), key=lambda i: -lens[i]):
        l, s = heapq.heappop(heap)
        load[s] += lens[i]
        heapq.heappush(heap, (load[s], s))
    load = sorted(load, reverse=True)
    smax, total = load[0], sum(lens)
    nact = np.zeros(smax, int)
    for l in load:
        nact[:l] += 1
    rowoff = np.concatenate([[0], np.cumsum(nact)])
    chunks, t0 = [], 0
    while t0 < smax:
        t1 = min(smax, max(t0 + 1, int(np.searchsorted(rowoff, rowoff[t0] + CAP, side="right") - 1)))
        chunks.append((t0, t1))
        t0 = t1
    print(f"{len(lens)} clips, {total} frames, {smax} sequential steps, slot loads at every 16th slot: {load[::16]}")
    print(f"all 128 slots live for {load[-1]} steps; <= 16 slots live from step {load[16]} on")
    rows = []
    for a, b in chunks:
        rec = sum(1.67 + 0.0102 * min(16, nact[t]) for t in range(a, b)) / 1000.0
        rows.append((int(np.ceil(nact[a] / 16)), rec, (rowoff[b] - rowoff[a]) * FF_MS / total))
    rec_ms, ff_ms = sum(r[1] for r in rows), sum(r[2] for r in rows)
    print(f"serial: recurrence {rec_ms:.1f} + feed-forward {ff_ms:.1f} = {rec_ms + ff_ms:.1f} ms per pass (model; measured 125)")
    for eff in (1.0, 0.85, 0.7):
        t = rows[0][2]
        hidden = 0.0
        for i, (gd, rec, _) in enumerate(rows):
            nxt = rows[i + 1][2] if i + 1 < len(rows) else 0.0
            done = min(nxt, rec * (8 - gd) / 8 * eff)
            hidden += done
            t += rec + nxt - done
        print(f"feed-forward of chunk c+1 on the XCDs the recurrence of chunk c does not hold, at {eff:.2f} of the whole-chip rate: "
              f"{t:.1f} ms per pass ({hidden:.1f} ms hidden)")
    idle = sum(r[1] * (8 - r[0]) / 8 for r in rows)
    print(f"idle XCD time under the recurrence: {idle:.1f} chip-ms, of which {sum(r[1] * (8 - r[0]) / 8 for r in rows if r[0] <= 1):.1f} "
          f"while at most one group is alive (the feed-forward left for those steps: {sum(r[2] for r in rows if r[0] <= 1):.1f} ms)")


if __name__ == "__main__":
    main()
