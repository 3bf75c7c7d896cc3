"""CPU: the slot enumeration of the pipelined fp4 Hamming kernel (csrc/hamming_mfma.hip HammingSlots).

A block of hamming_mfma_pipe_kernel walks its train tiles as slots of four-tile stages: its full tiles, then phantoms, the
pair's ragged tile in the last slot of the last stage (where the kernel uses the C operand that excludes the rows >= nt - 1).
The struct is host + device code; rgbdfe_debug_hamming_slots returns what a block would walk.  Checked here for every train
size and split count the launch geometry can produce: the blocks of a pair cover every tile of [0, ceil((nt - 1) / 32)) exactly
once, a tile with excluded rows only ever sits in a "last slot of the last stage", and nothing else does."""
import ctypes

import numpy as np

from rgbdslam_v2_amd import _lib

PHANTOM = 0x7FFFFFFF


def _slots(L, nt, tsplit, split):
    out = (ctypes.c_uint32 * 4096)()
    rag, ntt = ctypes.c_int(0), ctypes.c_uint32(0)
    n = L.rgbdfe_debug_hamming_slots(nt, tsplit, split, out, 4096, ctypes.byref(rag), ctypes.byref(ntt))
    assert n >= 0
    return np.array(out[:n], dtype=np.uint32), bool(rag.value), int(ntt.value)


def test_blocks_of_a_pair_cover_every_train_tile_once():
    L = ctypes.CDLL(_lib.LIB_PATH)
    L.rgbdfe_debug_hamming_slots.restype = ctypes.c_int
    L.rgbdfe_debug_hamming_slots.argtypes = [ctypes.c_uint32] * 3 + [ctypes.POINTER(ctypes.c_uint32), ctypes.c_int,
                                                                    ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_uint32)]
    rng = np.random.default_rng(5)
    sizes = list(range(0, 200)) + [255, 256, 257, 999, 1000, 1001, 1024, 1025, 1056, 1057, 3999, 4000, 4001, 4095, 4096, 4097,
                                   32767, 32768] + [int(v) for v in rng.integers(200, 33000, 300)]
    for nt in sizes:
        nts = max(nt - 1, 0)
        n_ttiles = (nts + 31) // 32
        for tsplit in (1, 2, 3, 5, 8, 17, 32):
            seen = np.zeros(n_ttiles, np.int32)
            for split in range(tsplit):
                tiles, has_ragged, ntt = _slots(L, nt, tsplit, split)
                assert ntt == n_ttiles and len(tiles) % 4 == 0
                real = tiles[tiles != PHANTOM]
                assert np.all(real < n_ttiles)
                np.add.at(seen, real, 1)
                # a tile with excluded rows (the ragged one) may only sit in the last slot of the block's last stage ...
                ragged = real[(real.astype(np.int64) * 32 + 32) > nts]
                if len(ragged):
                    assert has_ragged and len(ragged) == 1 and tiles[-1] == ragged[0]
                # ... and that slot holds nothing but the ragged tile or a phantom
                if has_ragged:
                    assert tiles[-1] == n_ttiles - 1 and (n_ttiles * 32 > nts)
                # full tiles come first, in order
                lead = tiles[:len(real) - (1 if has_ragged else 0)]
                assert np.all(np.diff(lead.astype(np.int64)) == 1) if len(lead) > 1 else True
            assert np.all(seen == 1), (nt, tsplit)
