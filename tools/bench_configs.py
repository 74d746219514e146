#!/usr/bin/env python3
"""Secondary measurements for the BASELINE.json configs that are parity-test cases rather than the headline bench
line (run on the GPU box; prints one JSON object per config).  The measurements live in legosnark_amd/benchcfg.py,
which bench.py also runs for its `configs` block: every time belongs to a checked result."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log2n", type=int, default=20)
    ap.add_argument("--d", type=int, default=20)
    ap.add_argument("--log2pairs", type=int, default=12)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--only", default="", help="comma-separated: cplink_prover,g2_msm,cppoly,pairing,cphad_verify")
    args = ap.parse_args()
    import numpy as np
    import torch
    import legosnark_amd as lsa
    from legosnark_amd import benchcfg
    lsa.init(0)
    only = set(args.only.split(",")) if args.only else None
    for line in benchcfg.measure(lsa, torch, np, torch.device("cuda:0"), log2n=args.log2n, d=args.d, log2pairs=args.log2pairs, reps=args.reps, only=only):
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
