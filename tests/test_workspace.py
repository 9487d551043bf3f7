"""dpvo_amd.workspace: caller-owned scratch handed to the C ABI.  Grown geometrically (round 5): the global BA's scratch grows by a few
KB with every frame that runs it, and a buffer sized exactly was a fresh 15-100 MB device allocation per frame."""
import torch

from dpvo_amd import workspace


def test_scratch_buffers_grow_geometrically_and_are_reused():
    dev = torch.device("cpu")
    a = workspace.get(3 << 20, dev, "t_ws")
    assert a.numel() >= 3 << 20 and a.dtype == torch.uint8
    assert workspace.get(1 << 20, dev, "t_ws") is a and workspace.get(3 << 20, dev, "t_ws") is a       # smaller or equal: the same buffer
    grown, allocs, need = a, 0, 3 << 20
    for _ in range(200):                                    # a request that creeps up by 64 KB per call, as the global BA's does
        need += 64 << 10
        b = workspace.get(need, dev, "t_ws")
        assert b.numel() >= need
        if b is not grown:
            allocs += 1
            assert b.numel() >= grown.numel() + grown.numel() // 2          # at least x 1.5: geometric
            grown = b
    assert allocs <= 5, allocs                              # 3 MB -> 15.5 MB in steps of 64 KB: a handful of allocations, not 200
    assert workspace.get(16, dev, "t_ws_other") is not grown                # tags do not share
    assert workspace.get(16, dev, "t_ws_other").numel() >= 1 << 20          # (minimum size)
