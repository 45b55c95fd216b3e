#!/usr/bin/env python3
"""Build experimental variants of libfishing_hip.so side by side (gym_fishing_amd/_lib/variants/):

    python scripts/build_variants.py name=-DFLAG[,-DFLAG2] ...      (an empty flag list = the default build)
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gym_fishing_amd import build
from concurrent.futures import ThreadPoolExecutor

def one(spec):
    name, _, flags = spec.partition("=")
    out = os.path.join(build.LIB_DIR, "variants", "libfishing_hip_%s.so" % name)
    build.build(out=out, extra_flags=[f for f in flags.split(",") if f])
    return out

if __name__ == "__main__":
    with ThreadPoolExecutor(max_workers=3) as ex:
        for o in ex.map(one, sys.argv[1:]):
            print(o)
