"""Freezes a few outputs of the CPU oracle into ``oracle_karate.npz`` (self-golden: a regression
pin between rounds, NOT a reference-derived vector -- the reference has none).

    python tests/golden/make_oracle_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import embiggen_amd as E  # noqa: E402
from oracle import oracle as O  # noqa: E402

if __name__ == "__main__":
    g = E.karate_club()
    og = O.OracleGraph(g.row_ptr, g.col_idx)
    wp = O.WalkParams(16, 2, 0.25, 4.0, 100, 0)
    out = {"walks": O.walks(og, wp, 42, 0, 0, 68)}
    for model, key in ((0, "sgns"), (1, "cbow")):
        tp = O.TrainParams(model, 8, 8, 2, 4, 3, 0.01, 0.9, 6.0, 1, 8 ** -0.5)
        c, x, pairs = O.fit(og, wp, tp, 42)
        out[f"{key}_central"], out[f"{key}_contextual"], out[f"{key}_pairs"] = c, x, pairs
    out["ba_dst"] = O.ba_edges(500, 3, 42)[1]
    np.savez_compressed(os.path.join(HERE, "oracle_karate.npz"), **out)
    print("written", {k: np.shape(v) for k, v in out.items()})

    # block-partitioned schedule (round 2): rank 1 of 2, 4 parts x 2 slices, two hot rows per cell
    blk = {}
    plan = O.block_plan(34, 2, 1, 4, 2, 16, 3, 1, 4, hot_rows=2)
    alias, cell_rows, hub_bits, hot_list, hot_slot = O.block_alias(og, 4, 2, 2)
    words, offsets = O.block_extract(og, plan, out["walks"], 42, 0, 0, hub_bits=hub_bits)
    tp = O.TrainParams(0, 8, 8, 1, 4, 3, 0.01, 0.9, 6.0, 1, 8 ** -0.5)
    central = O.init_table_rows(17, 8, 8, 42, 0, 8 ** -0.5, 1, 2)
    parts = []
    for part in range(4):
        rows = (34 - part + 3) // 4
        x = O.init_table_rows(rows, 8, 8, 42, 1, 8 ** -0.5, part, 4)
        O.block_step(og, tp, plan, words, offsets, alias, cell_rows, central, x, 7, part, 42,
                     0, 0.05)
        parts.append(x)
    blk.update(words=words, offsets=offsets, alias=alias, cell_rows=cell_rows,
               hub_bits=hub_bits, hot_list=hot_list, hot_slot=hot_slot, central=central, **{f"part{p}": x for p, x in enumerate(parts)})
    # round 5: the same walks under the placement of round 3 (whole graph: classes = 1), a plan
    # of resident cells (1 part x 17 slices: two rows a cell), the contextual table in node order
    place, inv = O.block_placement(34, 1, 42, 3)
    rplan = O.block_plan(34, 1, 0, 1, 17, 16, 3, 1, 8)
    ralias, rcell_rows = O.block_alias(og, 1, 17, 0, inv=inv)[:2]
    rwords, roffsets = O.block_extract(og, rplan, out["walks"], 42, 0, 0, place=place)
    rc = O.init_table(34, 8, 8, 42, 0, 8 ** -0.5)
    rx = O.init_table(34, 8, 8, 42, 1, 8 ** -0.5)
    O.block_step(og, tp, rplan, rwords, roffsets, ralias, rcell_rows, rc, rx, 3, 0, 42, 0, 0.05,
                 inv=inv, natural=True)
    blk.update(placed_place=place, placed_inv=inv, placed_words=rwords, placed_offsets=roffsets,
               placed_alias=ralias, placed_central=rc, placed_contextual=rx)
    np.savez_compressed(os.path.join(HERE, "oracle_blocks.npz"), **blk)
    print("written", {k: np.shape(v) for k, v in blk.items()})
