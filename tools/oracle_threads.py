"""How many host threads should the CPU oracle use on this box?  One full-size STDiTBlock forward-sample of the oracle
(the sample of bench.py's cpu_baseline) timed at several torch thread counts - the evidence behind the thread cap of
tests/conftest.py and the sweep inside cpu_baseline.  Measures the CHECKER, never the product.
    python tools/oracle_threads.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    n0 = torch.get_num_threads()
    print("default threads", n0, "cpu_count", os.cpu_count(), flush=True)
    once = bench.cpu_block_timer()
    for n in [n0, 96, 64, 48, 32, 24, 16, 8]:
        if n > n0:
            continue
        torch.set_num_threads(n)
        print("threads %3d  best of 2 = %.2f s" % (n, min(once(), once())), flush=True)
    torch.set_num_threads(n0)


if __name__ == "__main__":
    main()
