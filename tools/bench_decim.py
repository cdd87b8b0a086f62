#!/usr/bin/env python3
"""The two decimating front-end lines of bench.py's extra.side alone (C3: remez127 /50, C4: BH151 /34; chunk list in one launch, chunk
loop, one chunk), a few times: for A/B runs of dd_chain.hip builds (tools/each_variant.sh python tools/bench_decim.py)."""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
args = argparse.Namespace(log2n=26, force_direct=False)
eng = bench.HipStep(args, 0, 0)
for rnd in range(int(os.environ.get("ROUNDS", "2"))):
    for r in bench.side_configs(eng, steps=20, only_decim=True):
        print("%-16s chunk list %.4f ms (%.3f)   loop %.4f ms   one chunk %.4f ms (%.3f)" % (r["config"][:16], r["ms_per_pass"], r["frac_of_8TBs"], r["chunk_loop"]["ms"],
                                                                                          r["one_chunk"]["ms"], r["one_chunk"]["frac_of_8TBs"]))
eng.close()
