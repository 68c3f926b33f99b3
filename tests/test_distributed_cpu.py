"""Walk partitioning across ranks (embiggen_amd/distributed.py).  The N > 1 exchange itself is
covered by tests/test_blocks_cpu.py (2 gloo ranks) and tests/test_gpu_blocks.py."""
def test_walk_slices_are_disjoint_and_contiguous():
    from embiggen_amd.distributed import walk_slice

    seen = []
    for step in range(4):
        for rank in range(3):
            first, n = walk_slice(step, rank, 3, 10)
            seen.extend(range(first, first + n))
    assert seen == list(range(4 * 3 * 10))


