"""The argument behind k_tree's batched merge (himg_amd/csrc/kernels_enc.hip), checked on the CPU.

The reference builds its Huffman tree by joining, again and again, the two lightest nodes under
the total order (count ascending, node index DESCENDING) -- huffman_enc.cpp:183-238, trap T5: the
tie-breaking is part of the format.  k_tree does not take those merges one by one: with x1, x2
the two lightest nodes and s = count(x1) + count(x2), every node created from now on weighs at
least s, so all nodes lighter than s are consumed first, in that order and in consecutive pairs;
an odd last one is left over for the next batch.  This file states that batching in plain Python
(same thresholds and the same 64-per-kind cap as the kernel; what a batch does not take goes to
the serial loop, like there) and compares the sequence of merges with the O(n^2) definition on
tie-heavy, geometric, heavy-tailed, uniform and all-equal histograms.  The kernel itself is
checked byte for byte by the GPU parity tests; this pins the reasoning it rests on."""
import numpy as np


def merges_by_definition(counts):
    cnt = list(counts)
    alive = list(range(len(cnt)))
    out = []
    while len(alive) > 1:
        alive.sort(key=lambda i: (cnt[i], -i))
        a, b = alive[0], alive[1]
        cnt.append(cnt[a] + cnt[b])
        out.append((a, b))
        alive = alive[2:] + [len(cnt) - 1]
    return out


def merges_in_batches(counts, cap=64):
    n = len(counts)
    cnt = list(counts)
    leaves = sorted(range(n), key=lambda i: (cnt[i], -i))
    lh, qh, nxt, y = 0, n, n, None
    out, rounds, serial = [], 0, 0
    while (n - lh) + (nxt - qh) + (y is not None) >= 2:
        heads = ([cnt[y]] if y is not None else []) + [cnt[leaves[k]] for k in range(lh, min(lh + 2, n))] \
            + [cnt[q] for q in range(qh, min(qh + 2, nxt))]
        heads.sort()
        s = heads[0] + heads[1]
        if lh + cap < n and cnt[leaves[lh + cap]] < s:
            s = cnt[leaves[lh + cap]]
        if qh + cap < nxt and cnt[qh + cap] < s:
            s = cnt[qh + cap]
        L = [leaves[k] for k in range(lh, min(lh + cap, n)) if cnt[leaves[k]] < s]
        I = [q for q in range(qh, min(qh + cap, nxt)) if cnt[q] < s]
        yin = y is not None and cnt[y] < s
        m = int(yin) + len(L) + len(I)
        if m < 2:      # the serial loop takes the rest
            rest = ([y] if y is not None else []) + [leaves[k] for k in range(lh, n)] + list(range(qh, nxt))
            while len(rest) > 1:
                rest.sort(key=lambda i: (cnt[i], -i))
                a, b = rest[0], rest[1]
                cnt.append(cnt[a] + cnt[b])
                out.append((a, b))
                rest = rest[2:] + [len(cnt) - 1]
                serial += 1
            return out, rounds, serial
        # rank inside the batch exactly as the kernel computes it
        ranked = {}
        base = int(yin)
        if yin:
            ranked[0] = y
        for a, leaf in enumerate(L):
            le = sum(1 for q in I if cnt[q] <= cnt[leaf])
            ranked[base + a + le] = leaf
        for b, q in enumerate(I):
            run_b = b
            while run_b > 0 and cnt[I[run_b - 1]] == cnt[q]:
                run_b -= 1
            run_e = b + 1
            while run_e < len(I) and cnt[I[run_e]] == cnt[q]:
                run_e += 1
            kb = run_b + (run_e - 1 - b)
            lt = sum(1 for leaf in L if cnt[leaf] < cnt[q])
            ranked[base + kb + lt] = q
        assert sorted(ranked) == list(range(m)), "ranks are a permutation"
        B = [ranked[k] for k in range(m)]
        for k in range(m // 2):
            cnt.append(cnt[B[2 * k]] + cnt[B[2 * k + 1]])
            out.append((B[2 * k], B[2 * k + 1]))
        y = B[m - 1] if m & 1 else None
        lh, qh, nxt, rounds = lh + len(L), qh + len(I), nxt + m // 2, rounds + 1
    return out, rounds, serial


def _hist(kind, n, rng):
    if kind == 0:
        return [int(x) for x in rng.integers(1, 4, n)]                 # ties everywhere
    if kind == 1:
        return [int(x) for x in rng.integers(1, 1000000, n)]
    if kind == 2:
        return [int(1.3 ** k) + 1 for k in range(n)]                    # geometric: one merge per batch
    if kind == 3:
        return [7] * n                                                  # all equal: the serial loop's case beyond 64
    return [int(x) for x in (rng.pareto(1.0, n) * 10 + 1)]              # heavy tail


def test_batches_reproduce_the_definition():
    rng = np.random.default_rng(5)
    batched_steps = serial_steps = 0
    for t in range(300):
        n = int(rng.integers(2, 262))
        h = _hist(t % 5, n, rng)
        want = merges_by_definition(h)
        got, rounds, serial = merges_in_batches(h)
        assert got == want, (t % 5, n)
        if t % 5 in (1, 2, 4):     # (kinds 0 and 3 are the long runs of equal counts the serial loop is kept for)
            batched_steps += len(want) - serial
            serial_steps += serial
    assert batched_steps > 10 * serial_steps      # elsewhere the batches do the bulk of the work


def test_token_like_histogram_takes_few_batches():
    """Counts like a frame's token histogram (a few huge bins, a long tail of small ones)."""
    rng = np.random.default_rng(9)
    h = sorted(int(x) for x in (rng.pareto(0.7, 180) * 3 + 1))
    want = merges_by_definition(h)
    got, rounds, serial = merges_in_batches(h)
    assert got == want and serial == 0 and rounds < len(want) // 3
