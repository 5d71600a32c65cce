"""STORM_t serialized form (host side, no GPU needed): the reference defines only its size
(storm.c:372-394, :963-973); STORM_serialize must write exactly that many bytes, round-trip through
STORM_deserialize, and reject damaged streams."""
import numpy as np
import pytest

import stormbitmaps_amd as sb
from stormbitmaps_amd import synth

CASES = [  # (M, N, draws): list-kind only | bitmap-kind only | mixed kinds | ragged with empty rows
    (524288, 60, 524), (131072, 40, 60000), (196608, 80, 12690), (70000, 33, 7),
]


@pytest.mark.parametrize("M,N,d", CASES)
def test_serialize_writes_the_reference_byte_count_and_round_trips(orc, M, N, d):
    rows = synth.positions(M, N, d, seed=M + d)
    rows[N // 2] = np.zeros(0, dtype=np.uint32)          # an empty row still takes a container (storm.c:864)
    s = sb.Storm()
    for r in rows:
        s.add(r)
    o = orc.storm(rows)
    assert s.serialized_size() == o.serialized_size()   # the oracle's restatement of the size formulas
    data = s.serialize()
    assert data.size == s.serialized_size()
    assert int(np.frombuffer(data[:4].tobytes(), dtype=np.uint32)[0]) == N
    back = sb.Storm.deserialize(data)
    assert back.serialized_size() == data.size
    assert np.array_equal(back.serialize(), data)
    # the rebuilt container keeps working as a container
    back.add(rows[0])
    assert back.serialized_size() > data.size
    s.free()
    back.free()


def test_deserialize_rejects_damaged_streams():
    rows = synth.positions(196608, 20, 9000, seed=3)
    s = sb.Storm()
    for r in rows:
        s.add(r)
    data = s.serialize()
    s.free()
    for bad in (data[:-1], data[:7], data[: data.size // 2], np.concatenate([data, np.zeros(2, np.uint8)])):
        with pytest.raises(ValueError):
            sb.Storm.deserialize(bad)
    wrong_magic = data.copy(); wrong_magic[4] ^= 0xFF
    with pytest.raises(ValueError):
        sb.Storm.deserialize(wrong_magic)
    more_rows = data.copy(); more_rows[0] += 1             # claims a row that is not there
    with pytest.raises(ValueError):
        sb.Storm.deserialize(more_rows)
    empty = sb.Storm()
    assert empty.serialize().size == 8 == empty.serialized_size()
    assert sb.Storm.deserialize(empty.serialize()).serialized_size() == 8


def test_deserialize_rejects_headers_that_lie():
    """A stream is bounded by its own length before anything is allocated, and a list must be strictly
    ascending and as long as the block's set-bit count (the device walker of
    storm_hip_sparse_create_serialized applies the same rules: tests/test_gpu_round3.py)."""
    import struct
    # 16 bytes claiming 2^32 - 1 rows: rejected without sizing an allocation by it
    huge = np.frombuffer(struct.pack("<IIII", 0xFFFFFFFF, 0x314D5453, 0, 0), dtype=np.uint8)
    with pytest.raises(ValueError):
        sb.Storm.deserialize(huge)
    rows = [np.array([5, 9, 70000, 70001], dtype=np.uint32), np.array([1, 2, 3], dtype=np.uint32)]
    s = sb.Storm()
    for r in rows:
        s.add(r)
    data = s.serialize()
    s.free()
    assert sb.Storm.deserialize(data).serialized_size() == data.size
    # row 0: header 12 + 2 ids (8) + block header 16 + list [5, 9] -> swap the two list entries
    at = 8 + 12 + 8 + 16
    assert struct.unpack_from("<HH", data.tobytes(), at) == (5, 9)
    unsorted = data.copy(); unsorted[at:at + 4] = np.frombuffer(struct.pack("<HH", 9, 5), dtype=np.uint8)
    duplicate = data.copy(); duplicate[at:at + 4] = np.frombuffer(struct.pack("<HH", 5, 5), dtype=np.uint8)
    wrong_count = data.copy(); wrong_count[8 + 12 + 8 + 4] = 7          # n_bits_set of the first block
    wrong_id = data.copy(); wrong_id[8 + 12 + 8 + 12] = 3               # block id != block_ids[0]
    for bad in (unsorted, duplicate, wrong_count, wrong_id):
        with pytest.raises(ValueError):
            sb.Storm.deserialize(bad)
