"""Regenerates tests/golden/synth_totals.json.

Run from the repo root in the BUILD container:  python tests/golden/make_golden.py
Inputs come from the repo's own splitmix64 generator (stormbitmaps_amd/synth.py, seed 42);
expected totals come from the CPU oracle (oracle/, scalar leaf), and each is cross-checked
against both independent truths (naive bit loop where affordable, column-count identity
always) before it is written. The reference itself cannot be run here, so
these vectors pin the product against the oracle, and the oracle is pinned against the
reference by survey_totals.json.
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

from stormbitmaps_amd import synth  # noqa: E402
from tests._orc import Oracle  # noqa: E402

DENSE = [  # (name, M, N, draws)
    ("c1_dense", 4096, 256, 2048), ("c1_40", 4096, 256, 40), ("c1_5", 4096, 256, 5),
    ("c1_1", 4096, 256, 1),
    ("m65536_n700_dense", 65536, 700, 32768), ("m65536_n700_6553", 65536, 700, 6553),
    ("m65536_n700_262", 65536, 700, 262), ("m65536_n700_13", 65536, 700, 13),
    ("m65536_n700_150", 65536, 700, 150), ("m65536_n700_1", 65536, 700, 1),
    ("c2_first2000", 65536, 2000, 32768), ("c3_first300", 524288, 300, 262144),
    ("ragged_m1000_n130", 1000, 130, 300), ("ragged_m70_n3", 70, 3, 20),
    ("n2", 4096, 2, 2048), ("n129", 4096, 129, 1024), ("n257", 8192, 257, 2048),
    ("n513_w3", 130, 513, 60),
]
SPARSE = [  # STORM_t: (name, M, N, draws)
    ("c4_n300_524", 524288, 300, 524), ("c4_n300_5242", 524288, 300, 5242),
    ("c4_n300_20971", 524288, 300, 20971), ("c4_n300_52428", 524288, 300, 52428),
    ("c4_n300_131072", 524288, 300, 131072), ("c4_n300_262144", 524288, 300, 262144),
    ("mixed_kinds_4200", 65536, 300, 4200), ("mixed_kinds_4230", 65536, 300, 4230),
    ("mixed_kinds_4600", 65536, 300, 4600),
    ("sparse_5", 524288, 300, 5), ("sparse_1", 524288, 300, 1),
]


def main():
    orc = Oracle()
    out = {"_about": __doc__.strip(), "generator": "splitmix64", "seed": 42, "dense": [],
           "sparse": []}
    for name, M, N, d in DENSE:
        mat = synth.dense_matrix(M, N, d, seed=42)
        total = orc.wrapper_diag(mat, kind=0)
        assert total == orc.truth_columns(mat), name
        if N * N * mat.shape[1] < 3e8:
            assert total == orc.truth_naive(mat), name
        rows = synth.positions_from_dense(mat)
        c = orc.contig(M, rows)
        assert {c.pairw(), c.pairw_blocked(7), c.pairw_list() if c.cutoff() else total,
                orc.wrapper_diag_blocked(mat, 31)} == {total}, name
        out["dense"].append({"name": name, "M": M, "N": N, "draws": d, "total": total})
        print(name, total)
    for name, M, N, d in SPARSE:
        mat = synth.dense_matrix(M, N, d, seed=42)
        rows = synth.positions_from_dense(mat)
        s = orc.storm(rows)
        total = s.pairw()
        assert total == s.pairw_blocked(0) == orc.truth_columns(mat), name
        n_list, n_bitmap = s.census()
        out["sparse"].append({"name": name, "M": M, "N": N, "draws": d, "total": total,
                              "list_blocks": n_list, "bitmap_blocks": n_bitmap,
                              "serialized_size": s.serialized_size()})
        print(name, total, n_list, n_bitmap)
    with open(os.path.join(os.path.dirname(__file__), "synth_totals.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
