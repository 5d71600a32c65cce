"""Provenance of tests/golden/survey_totals.json — what in it can be regenerated here, and how.

Run from the repo root in the BUILD container:  python tests/golden/make_survey_totals.py [--write]

The file holds 7 all-pairs totals (plus 4 defect cases) that SURVEY.md §8c / §8 a-note recorded
at survey time from the UNMODIFIED reference storm.c driven with harness-equivalent inputs. Two
halves of that provenance, kept apart on purpose:

  * `total` / `reference_value` — numbers the SURVEY recorded from the reference. The reference
    cannot be rebuilt in this image (storm.h:33 needs the un-vendored libalgebra, benchmark.cpp
    needs CRoaring; stand-in headers are not allowed), so these fields are transcribed from
    SURVEY.md and this script NEVER changes them. Parity therefore remains "unpinned by the
    reference's own tests"  — this script does not alter that status.
  * `oracle_total` / `truth` — what THIS repo can regenerate: the same inputs
    (oracle/mt19937_inputs.cpp: libstdc++ std::mt19937(42) + uniform_int_distribution, the
    harness's draw loop, benchmark.cpp:750-772) pushed through every entry point of the CPU
    oracle (oracle/storm_oracle.c) and through two independent truths (naive bit loop, column
    identity). The script recomputes them, checks that every oracle entry point agrees, and
    compares with the recorded numbers; `--write` stores them next to the recorded ones.

Output: one table row per case; exit status 1 if any regenerated number differs from the file.
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

from tests._orc import Oracle  # noqa: E402

PATH = os.path.join(ROOT, "tests", "golden", "survey_totals.json")


def dense_from_rows(rows, M):
    W = (M + 63) // 64
    mat = np.zeros((len(rows), W), dtype=np.uint64)
    for i, r in enumerate(rows):
        r = np.asarray(r, dtype=np.uint64)
        np.bitwise_or.at(mat[i], (r >> np.uint64(6)).astype(np.int64), np.uint64(1) << (r & np.uint64(63)))
    return mat


def oracle_entry_points(orc, M, N, draws, seed):
    """Every oracle entry point + both truths on the survey's inputs -> {name: total}."""
    rows = orc.mt_positions(M, N, draws, seed)
    mat = dense_from_rows(rows, M)
    W = mat.shape[1]
    bsize = max(5, int(256e3 // (W * 8)))  # benchmark.cpp:823-824
    c = orc.contig(M, rows)
    out = {
        "truth_naive": orc.truth_naive(mat),
        "truth_columns": orc.truth_columns(mat),
        "wrapper_diag": orc.wrapper_diag(mat),
        "wrapper_diag_blocked": orc.wrapper_diag_blocked(mat, bsize),
        "contig_pairw": c.pairw(),
        "contig_pairw_blocked": c.pairw_blocked(bsize),
        "contig_pairw_blocked_7": c.pairw_blocked(7),
        "contig_pairw_list": c.pairw_list(),
        "contig_pairw_blocked_list": c.pairw_blocked_list(bsize),
    }
    if M >= 65536:  # the harness gates the STORM_t rows on M >= 65536 (benchmark.cpp:832)
        s = orc.storm(rows)
        out["storm_pairw"] = s.pairw()
        out["storm_pairw_blocked_0"] = s.pairw_blocked(0)
    return out


def main():
    write = "--write" in sys.argv[1:]
    doc = json.load(open(PATH))
    orc = Oracle()
    seed = doc["seed"]
    bad = 0
    print(f"{'case':<34}{'recorded':>14}{'oracle (all entry points)':>28}  status")
    for case in doc["agree"]:
        got = oracle_entry_points(orc, case["M"], case["N"], case["draws"], seed)
        vals = set(got.values())
        ok = vals == {case["total"]}
        bad += not ok
        name = f"M={case['M']} N={case['N']} draws={case['draws']}"
        print(f"{name:<34}{case['total']:>14}{'/'.join(str(v) for v in sorted(vals)):>28}  "
              f"{'agree' if ok else 'DIFFER: ' + json.dumps(got)}")
        case["oracle_total"] = got["truth_naive"]
        case["oracle_entry_points"] = sorted(got)
    for case in doc["defects"]:
        got = oracle_entry_points(orc, case["M"], case["N"], case["draws"], seed)
        vals = set(got.values())
        ok = vals == {case["truth"]}
        bad += not ok
        name = f"{case['id']} M={case['M']} N={case['N']} draws={case['draws']}"
        print(f"{name:<34}{case['truth']:>14}{'/'.join(str(v) for v in sorted(vals)):>28}  "
              f"{'truth reproduced' if ok else 'DIFFER: ' + json.dumps(got)}"
              f"  (reference recorded {case['reference_value']}: defect, never expected)")
        case["oracle_total"] = got["truth_naive"]
    if write:
        doc["_regenerated_by"] = ("tests/golden/make_survey_totals.py: oracle_total / truth fields are "
                                  "recomputed from oracle/mt19937_inputs.cpp + oracle/storm_oracle.c; "
                                  "total / reference_value are transcribed from SURVEY.md and never rewritten")
        with open(PATH, "w") as f:
            json.dump(doc, f, indent=1)
            f.write("\n")
        print(f"wrote {PATH}")
    print("all regenerated numbers equal the recorded ones" if not bad else f"{bad} case(s) differ")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
