"""Run GPU tests with the caching allocator's free memory POISONED (NaN bit patterns / huge values): a kernel that reads a buffer
before anything wrote it, or accumulates into one that nobody zeroed, then fails deterministically instead of once in a few
full-suite runs (when an earlier, larger test left other bytes in the reused blocks).  python scripts/poison_run.py <pytest args>"""
import sys
import pytest
import torch

if torch.cuda.is_available():
    junk = [torch.full((1 << 28,), float("nan"), device="cuda") for _ in range(6)]      # 6 GiB of NaN
    junk += [torch.full((1 << 28,), 3.0e38, device="cuda") for _ in range(2)]
    # the allocator serves requests below 1 MiB from a pool of its own (2 MiB segments) and 1-10 MiB ones from 20 MiB segments
    junk += [torch.full((n,), float("nan"), device="cuda") for n in (1 << 7, 1 << 10, 1 << 13, 1 << 16, 1 << 17) for _ in range(400)]
    junk += [torch.full((n,), float("nan"), device="cuda") for n in (1 << 19, 1 << 20, 1 << 21) for _ in range(100)]
    torch.cuda.synchronize()
    del junk                                                                            # back to the cache, contents intact
sys.exit(pytest.main(sys.argv[1:]))
