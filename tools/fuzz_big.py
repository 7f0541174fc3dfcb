#!/usr/bin/env python3
"""The randomised parity sweep (tools/fuzz_parity.py) on the geometries its default draw leaves out: BlockSize 8192 ... 32768 and
1 ... 6 channels (k_xf_big, the general decoder kernel, heaps that do not fit LDS).  python tools/fuzz_big.py [seconds] [seed]"""
import sys
import fuzz_parity
fuzz_parity.run(float(sys.argv[1]) if len(sys.argv) > 1 else 120.0, int(sys.argv[2]) if len(sys.argv) > 2 else 7,
                sizes=[8192, 8192, 16384, 32768], chans=[1, 2, 2, 3, 4, 6])
