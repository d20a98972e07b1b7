"""Global-memory bytes k_enc_basen<2> moves for ONE Enc, source by source, counted from the kernel's script (csrc/kernels_basen.hpp) — the
attribution the round-4 verdict asked for beside the PMC total (profiles/r05/traffic_split.json).  An n-sized value is 72 limbs = 288 B, a pair
576 B.  Squarings move nothing (everything they touch is in LDS); every other base-n product reads its multiplier pair from the group's table
slot and parks partial results.   python tools/dev/traffic_model.py [--old]   (--old: the round-4 kernel)"""
import importlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

synth = importlib.import_module("zk-paillier_amd.synth")
V = 288                                            # bytes of an n-sized value as 29-bit limbs in 32-bit words


def split(old: bool):
    sq, mul = bench.sliding_ladder_products(synth.BENCH_N)      # squarings, other products of the ladder (X0^2 and the 31 table rounds included)
    table_rounds = 32                                           # X0^2 and T[1..31]: destination = a table entry
    window = mul - table_rounds                                 # destination = the staged pair
    r = {}
    # operands in, raw pair out, the item's limbs parked in the table slot
    r["operands r, m (32-bit words) in"] = (2 * 256, 0)
    r["r, (1, m) as limbs into the table slot, read back by the first and the last product"] = (3 * V, 3 * V)
    r["key constants RR (L2-resident: one record per launch)"] = (2 * V, 0)
    r["raw pair out (to k_basen_finish)"] = (0, 2 * V)
    # to the Montgomery domain: two slots, no cross product; x~ becomes table entry 0
    r["to the Montgomery domain: multiplier, parked a part, entry 0"] = ((2 if old else 1) * V + V, V + 2 * V)
    # a product with a cross product rb * a:  rb in | ra in (twice in round 4) | U parked and read back | the a part parked and read back (round 4; table destinations still store it: it is the entry)
    per_window = (V + (2 if old else 1) * V + V + (V if old else 0), V + (V if old else 0))
    per_table = (V + (2 if old else 1) * V + V + (V if old else 0) + V, V + 2 * V)      # + the copy of X0^2.a restaged each round; a and b parts stored into the entry
    r[f"{window} window multiplications: table entry in (ra{' twice' if old else ''}, rb), cross product U parked + read" + (", a part parked + read" if old else "")] = tuple(window * v for v in per_window)
    r[f"{table_rounds} table rounds: the same + the entry written, X0^2.a restaged"] = tuple(table_rounds * v for v in per_table)
    r["first window: its table entry in"] = (2 * V, 0)
    r["final product by (1, m)"] = ((3 if old else 2) * V + V, V)
    rd = sum(v[0] for v in r.values()); wr = sum(v[1] for v in r.values())
    return {"kernel": "k_enc_basen<2>" + (" (round 4)" if old else " (round 5: a part of a window multiplication kept in registers, ra loaded once)"),
            "squarings": sq, "other_products": mul, "bytes_read": rd, "bytes_written": wr, "bytes_total": rd + wr,
            "by_source": {k: {"read": v[0], "written": v[1]} for k, v in r.items()}}


if __name__ == "__main__":
    print(json.dumps(split("--old" in sys.argv), indent=1))
