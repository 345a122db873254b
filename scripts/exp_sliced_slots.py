"""Offline slot statistics of the feature-sliced product (no GPU): how many padded lane-steps per CSR entry the
format of csrc/gcn_sliced.hip spends on the C4 graph under different assignments of rows to 64-row slots, and what bounds it.
    python scripts/exp_sliced_slots.py            -> profiles/r06_sliced_slots.txt (table quoted in profiles/r06_experiments.md)
Format rules restated from csrc/gcn_sliced.hip: a (panel, wave) pair owns R slots (rounds; snake deal over the pairs); per source
tile a round is as long as its longest row, in blocks of 8 steps; within a pair the round lengths are padded to a non-increasing
sequence (round 0 the longest); every lane of a slot steps through the slot's blocks (lock step)."""
import sys
import numpy as np

N, PAIRS, TILE = 132534, 39561252, 10208
rng = np.random.default_rng(0)
a = rng.integers(0, N, PAIRS, dtype=np.int64)
b = rng.integers(0, N, PAIRS, dtype=np.int64)
dst = np.concatenate([a, b, np.arange(N)])
src = np.concatenate([b, a, np.arange(N)])
NT = -(-N // TILE)
c3 = np.bincount((dst * NT + src // TILE) * 16 + src % 16, minlength=N * NT * 16).reshape(N, NT, 16).astype(np.int32)   # per (row, tile, bank quad)
cnt = c3.sum(axis=2)
nnz = int(cnt.sum())
G = -(-N // 64)
PW, R = 240, 9                      # the C4 plan: 16 panels x 15 waves, 9 rounds (profiles/r03_experiments.md)
assert PW * R >= G


def steps(order, quads=False):
    """padded lane-steps of the product when slot s holds rows order[64 s : 64 s + 64].  quads: a step of a 16-row lane group reads
    16 DISTINCT bank quads (source row mod 16: conflict-free ds_read_b128), so a round is also at least as long as the group's
    largest bank-quad column (the sum over its 16 rows of the entries in one quad) -- the format's real rule."""
    pad = np.full(G * 64, -1, dtype=np.int64)
    pad[:N] = order
    if quads:
        c = np.where(pad[:, None, None] >= 0, c3[np.clip(pad, 0, N - 1)], 0).reshape(G, 4, 16, NT, 16)
        env = np.maximum(c.sum(axis=4).max(axis=2), c.sum(axis=2).max(axis=3)).max(axis=1)
    else:
        env = np.where(pad[:, None] >= 0, cnt[np.clip(pad, 0, N - 1)], 0).reshape(G, 64, NT).max(axis=1)
    nb = -(-env // 8)                                             # blocks per (slot, tile)
    tot = 0
    for pw in range(PW):
        rounds = []
        for j in range(R):
            s = j * PW + (PW - 1 - pw if j & 1 else pw)
            rounds.append(nb[s] if s < G else np.zeros(NT, dtype=nb.dtype))
        r = np.stack(rounds)                                      # [R, NT]
        r = np.maximum.accumulate(r[::-1], axis=0)[::-1]          # non-increasing over rounds
        tot += int(r.sum()) * 8 * 64
    return tot


deg = cnt.sum(axis=1)
rows = np.arange(N)
out = []
out.append(("natural order, rows AND bank-quad columns (= the built format: 1.556 measured)", steps(rows, True)))
out.append(("natural order, row envelopes alone", steps(rows)))
out.append(("rows by descending degree", steps(np.argsort(-deg, kind="stable"))))
out.append(("rows by descending largest per-tile count (VERDICT r5 item 5, first idea)", steps(np.argsort(-cnt.max(axis=1), kind="stable"))))
blk = -(-cnt // 8)
top = blk.max(axis=1)
hot = ((blk == top[:, None]) * (1 << np.arange(NT)).astype(np.int64)).sum(axis=1)
best = np.lexsort((hot, -top))
out.append(("rows by (largest block count, set of tiles that reach it), row envelopes alone", steps(best)))
out.append(("  ... the same order WITH the bank-quad columns (built on the GPU: 1.569)", steps(best, True)))
key = np.lexsort(tuple(-(cnt[:, t] // 8) for t in range(NT - 1, -1, -1)))
out.append(("rows sorted lexicographically by their per-tile block counts", steps(key)))
# greedy envelope fill: seed a slot with the unused row of the largest degree, then add the 63 unused rows that raise the slot's
# per-tile block envelope the least (exact search over a candidate window of the 4,096 rows next in degree order)
left = list(np.argsort(-deg, kind="stable"))
greedy = []
import heapq
while left:
    seed = left.pop(0)
    env = -(-cnt[seed] // 8)
    slot = [seed]
    window = left[:4096]
    w = np.array(window, dtype=np.int64)
    while len(slot) < 64 and len(w):
        nbw = -(-cnt[w] // 8)
        cost = np.maximum(nbw, env[None, :]).sum(axis=1) - env.sum()
        i = int(np.argmin(cost))
        env = np.maximum(env, nbw[i])
        slot.append(int(w[i]))
        w = np.delete(w, i)
    chosen = set(slot[1:])
    left = [r for r in left if r not in chosen]
    greedy.extend(slot)
    if len(greedy) % 6400 == 0:
        print(len(greedy), file=sys.stderr)
out.append(("greedy envelope fill (window of 4,096 rows by degree)", steps(np.array(greedy, dtype=np.int64))))
# bounds
blocks_row = (-(-cnt // 8)).sum()
out.append(("BOUND: every row alone in its lane, only the 8-step blocks (no lock step between rows)", int(blocks_row) * 8))
out.append(("BOUND: no padding at all (= entries)", nnz))
with open("profiles/r06_sliced_slots.txt", "w") as f:
    f.write(f"# C4 synthetic graph: {N} rows, {nnz} entries, {NT} source tiles of {TILE} rows, {G} slots of 64 rows, plan {PW} pairs x {R} rounds\n")
    f.write(f"# per-(row, tile) count: mean {cnt.mean():.2f}, std {cnt.std():.2f}; largest of 64 independent rows ~ {np.sort(cnt[:6400, 0].reshape(100, 64).max(axis=1)).mean():.1f}\n")
    for name, s in out:
        f.write(f"{s / nnz:7.3f} lane-steps per entry  ({s / 1e6:7.1f} M)  {name}\n")
print(open("profiles/r06_sliced_slots.txt").read())
