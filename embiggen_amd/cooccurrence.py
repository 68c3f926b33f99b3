"""Co-occurrence counts of random walks for the GloVe model, kept in HBM.

The slots come from the ``gn2v_cooc_slots`` kernel; summing them by key (sort + segmented integer
sum), merging batches and ordering the training entries is torch tensor plumbing on the device.
Counts are fixed point (2^20 / distance per co-occurrence), so every sum is exact and independent
of the order of accumulation; the oracle (oracle/gn2v_oracle.c, "GloVe") produces the same
integers.  Reference call site: models.GloVe in embedders/ensmallen_embedders/node2vec.py:16-26.
"""
from typing import Optional, Tuple

UNUSED = 0x7FFFFFFFFFFFFFFF
_TAG_GLOVE = 0x610FE00000C00C01
_M64 = (1 << 64) - 1


def _signed(x: int) -> int:
    x &= _M64
    return x - (1 << 64) if x >= (1 << 63) else x


def mix64_int(z: int) -> int:
    z &= _M64
    z ^= z >> 30
    z = (z * 0xBF58476D1CE4E5B9) & _M64
    z ^= z >> 27
    z = (z * 0x94D049BB133111EB) & _M64
    return z ^ (z >> 31)


def _lsr(z, s: int):
    return (z >> s) & ((1 << (64 - s)) - 1)


def mix64_tensor(z):
    """splitmix64 finaliser on an int64 tensor (two's complement wrap-around = u64 arithmetic)."""
    z = z ^ _lsr(z, 30)
    z = z * _signed(0xBF58476D1CE4E5B9)
    z = z ^ _lsr(z, 27)
    z = z * _signed(0x94D049BB133111EB)
    return z ^ _lsr(z, 31)


def reduce_slots(keys, weights) -> Tuple["torch.Tensor", "torch.Tensor"]:
    """(distinct keys, summed weights) of the used slots; keys ascending as int64.  One radix sort,
    then segment sums as differences of an exact int64 running sum (no atomics, no inverse map)."""
    import torch

    keys, order = torch.sort(keys.view(torch.int64).flatten())
    weights = weights.view(torch.int64).flatten()[order]
    del order
    unused = torch.tensor([UNUSED], dtype=torch.int64, device=keys.device)
    used = int(torch.searchsorted(keys, unused)[0])  # unused slots sort behind every key
    keys, weights = keys[:used], weights[:used]
    if used == 0:
        return keys.clone(), weights.clone()
    last = torch.ones(used, dtype=torch.bool, device=keys.device)
    torch.ne(keys[1:], keys[:-1], out=last[:-1])
    ends = torch.nonzero(last).flatten()
    del last
    running = torch.cumsum(weights, 0)
    del weights
    sums = running[ends]
    del running
    sums[1:] -= sums[:-1].clone()
    return keys[ends], sums


SORT_LIMIT = 2 ** 31 - 1  # torch.sort refuses longer dimensions


def merge(a: Optional[tuple], b: tuple) -> tuple:
    """Sum of two reduced (keys ascending and distinct, counts) sets.  Sets too long for one sort
    are cut at a pivot key (both are sorted: two binary searches, no masks) and merged by halves."""
    import torch

    if a is None:
        return b
    if a[0].numel() + b[0].numel() <= SORT_LIMIT or min(a[0].numel(), b[0].numel()) == 0:
        if min(a[0].numel(), b[0].numel()) == 0:
            return a if b[0].numel() == 0 else b
        return reduce_slots(torch.cat([a[0], b[0]]), torch.cat([a[1], b[1]]))
    big, small = (a, b) if a[0].numel() >= b[0].numel() else (b, a)
    cut = big[0].numel() // 2
    pivot = big[0][cut:cut + 1]
    other = int(torch.searchsorted(small[0], pivot)[0])
    left = merge((big[0][:cut], big[1][:cut]), (small[0][:other], small[1][:other]))
    right = merge((big[0][cut:], big[1][cut:]), (small[0][other:], small[1][other:]))
    return torch.cat([left[0], right[0]]), torch.cat([left[1], right[1]])


class Accumulator:
    """Running sum of reduced batches.  Batches are kept as sorted runs and merged like a binary
    counter (two runs merge when the older is at most twice the newer), so every entry takes part
    in O(log batches) merges instead of one merge per batch."""

    def __init__(self):
        self.runs = []

    def _merge_last_two(self):
        import torch

        b, a = self.runs.pop(), self.runs.pop()
        if a[0].numel() + b[0].numel() > SORT_LIMIT:
            self.runs.append(merge(a, b))
            return
        keys, counts = torch.cat([a[0], b[0]]), torch.cat([a[1], b[1]])
        del a, b  # the inputs are dead before the sort allocates its buffers
        self.runs.append(reduce_slots(keys, counts))

    def add(self, run: tuple):
        self.runs.append(run)
        while len(self.runs) > 1 and self.runs[-2][0].numel() <= 2 * self.runs[-1][0].numel():
            self._merge_last_two()

    def result(self) -> tuple:
        while len(self.runs) > 1:
            self._merge_last_two()
        return self.runs.pop() if self.runs else None


RECORD = 16
_GOLDEN = 0x9E3779B97F4A7C15
_SIGN = -(1 << 63)


def _row_pieces(keys):
    """[lo, hi) ranges of ascending keys that hold whole rows and at most SORT_LIMIT entries each
    (torch.sort / nonzero / cumsum work on one range at a time, so the entry set itself may be
    longer than INT_MAX: a default GloVe fit of a 1 M-node graph holds > 2^31 distinct pairs)."""
    import torch

    n, lo, out = keys.numel(), 0, []
    while lo < n:
        hi = min(n, lo + SORT_LIMIT)
        if hi < n:  # back to the first entry of the row that holds keys[hi]
            first = (keys[hi:hi + 1] >> 32) << 32
            hi = lo + int(torch.searchsorted(keys[lo:hi], first)[0])
            if hi <= lo:
                raise RuntimeError(f"one row holds more than {SORT_LIMIT} co-occurrence entries")
        out.append((lo, hi))
        lo = hi
    return out


def entries(keys, counts, seed: int, alpha: float):
    """Training slots (rows i32, cols i32, log X f32, f(X) f32) laid out as records of RECORD
    consecutive slots that share their central row (the engine trains a record per wavefront with
    that row in registers): every row's entries in ascending mix64(key ^ salt) are cut into records,
    the records are shuffled (ascending draw(mix64(row ^ salt), record index in the row)); unused
    slots of a row's last record hold col = -1.  X = count / max count.  Same order as the oracle
    (o_glove_entries).  `keys` ascending and distinct (reduce_slots / merge leave them so)."""
    import torch

    dev = keys.device
    n = keys.numel()
    if n == 0:
        empty_i = torch.empty(0, dtype=torch.int32, device=dev)
        empty_f = torch.empty(0, dtype=torch.float32, device=dev)
        return empty_i, empty_i.clone(), empty_f, empty_f.clone()
    salt = _signed(mix64_int(seed ^ _TAG_GLOVE))
    # pass 1, a range of whole rows at a time: (row, hash of key) order, the rows' records
    pieces, rec_rows, rec_hash, n_rec, top = [], [], [], 0, None
    for lo, hi in _row_pieces(keys):
        piece = keys[lo:hi]
        m = hi - lo
        # stable sort by the hash, then stable sort by the row
        order = torch.argsort(mix64_tensor(piece ^ salt) ^ _SIGN, stable=True)
        order = order[torch.argsort(_lsr(piece[order], 32), stable=True)]
        pkeys, pcounts = piece[order], counts[lo:hi][order]
        del order, piece
        row = _lsr(pkeys, 32)
        start = torch.ones(m, dtype=torch.bool, device=dev)
        torch.ne(row[1:], row[:-1], out=start[1:])
        run_pos = torch.nonzero(start).flatten()
        run_id = torch.cumsum(start, 0) - 1
        del start
        rank = torch.arange(m, dtype=torch.int64, device=dev) - run_pos[run_id]
        run_len = torch.diff(run_pos, append=torch.tensor([m], dtype=torch.int64, device=dev))
        recs = torch.div(run_len + (RECORD - 1), RECORD, rounding_mode="floor")
        first_rec = torch.cumsum(recs, 0) - recs
        rec_id = first_rec[run_id] + torch.div(rank, RECORD, rounding_mode="floor") + n_rec
        del run_id, run_len
        piece_recs = int(recs.sum())
        rec_run = torch.repeat_interleave(torch.arange(recs.numel(), device=dev), recs)
        rec_row = row[run_pos][rec_run]
        rec_q = torch.arange(piece_recs, dtype=torch.int64, device=dev) - first_rec[rec_run]
        del rec_run, first_rec, recs, run_pos, row
        rec_hash.append(mix64_tensor(mix64_tensor(rec_row ^ salt) + (rec_q + 1) * _signed(_GOLDEN)) ^ _SIGN)
        rec_rows.append(rec_row)
        del rec_q
        pieces.append(((pkeys & 0xFFFFFFFF).to(torch.int32), pcounts, rec_id,
                       (rank % RECORD).to(torch.int8)))
        piece_top = pcounts.max()
        top = piece_top if top is None else torch.maximum(top, piece_top)
        del pkeys, rank
        n_rec += piece_recs
    if n_rec > SORT_LIMIT:
        raise RuntimeError(f"{n_rec} records of co-occurrence entries are more than one sort takes")
    rec_row = torch.cat(rec_rows) if len(rec_rows) > 1 else rec_rows[0]
    h = torch.cat(rec_hash) if len(rec_hash) > 1 else rec_hash[0]
    del rec_rows, rec_hash
    rec_order = torch.argsort(h, stable=True)  # ties keep the (row, index) order
    del h
    new_pos = torch.empty(n_rec, dtype=torch.int64, device=dev)
    new_pos[rec_order] = torch.arange(n_rec, dtype=torch.int64, device=dev)
    rows = rec_row[rec_order].to(torch.int32).unsqueeze(1).expand(n_rec, RECORD).reshape(-1)
    del rec_row, rec_order
    # pass 2: every entry into its slot
    cols = torch.full((n_rec * RECORD,), -1, dtype=torch.int32, device=dev)
    logx = torch.zeros(n_rec * RECORD, dtype=torch.float32, device=dev)
    fx = torch.zeros_like(logx)
    top = top.to(torch.float64)
    step = 1 << 26  # float64 temporaries of a slice at a time
    while pieces:
        pcols, pcounts, rec_id, lane = pieces.pop(0)
        slot = new_pos[rec_id] * RECORD + lane
        del rec_id, lane
        cols[slot] = pcols
        del pcols
        for lo in range(0, slot.numel(), step):
            x = (pcounts[lo:lo + step].to(torch.float64) / top).to(torch.float32).to(torch.float64)
            logx[slot[lo:lo + step]] = torch.log(x).to(torch.float32)
            fx[slot[lo:lo + step]] = torch.pow(x, float(alpha)).to(torch.float32)
        del slot, pcounts
    return rows, cols, logx, fx
